#!/usr/bin/env python3
"""8 small .gz files through `sc fq-count` with and without --jobs, SCFQ_VERBOSE traces: where does the set-up time of several sessions go"""
import gzip, os, subprocess, sys, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "seq-collection_amd", "pyhost"))
import scfq
plan = scfq.synth_plan(0, 20260101, 256 << 20)
data, info = scfq.synth_host(0, 20260101, plan.records)
member = 64 << 20
with ThreadPoolExecutor(8) as ex:
    blobs = list(ex.map(lambda i: gzip.compress(data[i:i + member].tobytes(), 6), range(0, data.size, member)))
many = []
for i in range(8):
    p = "/tmp/scfq_many_%d.fq.gz" % i
    open(p, "wb").write(b"".join(blobs))
    many.append(p)
sc = os.path.join(ROOT, "seq-collection_amd", "sc")
for arg in (["--jobs=1"], ["--jobs=4"], [], ["--jobs=2"]):
    for rep in range(2):
        t = time.time(); r = subprocess.run([sc, "fq-count"] + arg + many, capture_output=True, text=True, env=dict(os.environ, SCFQ_VERBOSE="1")); dt = time.time() - t
    print("====", arg, round(dt, 3), r.returncode)
    print("\n".join(l for l in r.stderr.splitlines() if "t+" in l or "wall" in l)[:3000])
