#!/usr/bin/env python3
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
for arg in (["--jobs=1"], ["--jobs=2"], ["--jobs=1"], ["--jobs=2"]):
    for rep in range(3):
        t = time.time(); r = subprocess.run([sc, "fq-count"] + arg + many, capture_output=True, text=True, env=dict(os.environ, SCFQ_VERBOSE="1")); dt = time.time() - t
        tl = [l for l in r.stderr.splitlines() if "t+" in l]
        print(arg, "wall %.3f" % dt, "| first:", tl[0].split("ms")[0].strip() if tl else "", "| last:", tl[-1].strip() if tl else "")
    time.sleep(1.0)
