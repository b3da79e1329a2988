#!/usr/bin/env python3
"""Phase sums of the parallel gzip reader over one file (SCFQ_VERBOSE laps of scfq_pgz.hpp): pgz_phase_sums.py <file.gz>"""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = "import sys; sys.path.insert(0, %r); import scfq; scfq.count_file(%r); import time; t = time.time(); c = scfq.count_file(%r); print('WALL', time.time() - t, c.input_bytes)" % (
    os.path.join(ROOT, "seq-collection_amd", "pyhost"), os.path.join(ROOT, "tests", "golden", "dup.fq.gz"), sys.argv[1])
r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SCFQ_VERBOSE="1"), capture_output=True, text=True)
sums, n = collections.Counter(), collections.Counter()
for line in r.stderr.splitlines():
    m = re.match(r"scfq pgz:\s+(.*?)\s+([0-9.]+) ms$", line)
    if m and not m.group(1).startswith("("):
        sums[m.group(1)] += float(m.group(2)); n[m.group(1)] += 1
for k, v in sums.items():
    print("%-24s %8.1f ms over %d batches" % (k, v, n[k]))
print(r.stdout.strip())
