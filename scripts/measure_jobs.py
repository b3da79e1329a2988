#!/usr/bin/env python3
"""`sc fq-count a.fq.gz b.fq.gz ...` over 8 gzip files of 256 MiB (4 members of 64 MiB each), process start included,
by --jobs.  The settings take turns in shuffled order (6 rounds, a second apart) so that the state of the box — a process that follows another one starts more slowly —
hits them alike; minimum and median per setting.  Rows must be identical (argv order, sc.nim:115-116).  One JSON line per setting."""
import gzip, json, os, statistics, subprocess, sys, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "seq-collection_amd", "pyhost"))
import scfq
tmp = sys.argv[1] if len(sys.argv) > 1 else "/tmp"
plan = scfq.synth_plan(0, 20260101, 256 << 20)
data, info = scfq.synth_host(0, 20260101, plan.records)
member = 64 << 20
with ThreadPoolExecutor(8) as ex:
    blobs = list(ex.map(lambda i: gzip.compress(data[i:i + member].tobytes(), 6), range(0, data.size, member)))
many = []
for i in range(8):
    p = os.path.join(tmp, "scfq_many_%d.fq.gz" % i)
    open(p, "wb").write(b"".join(blobs))
    many.append(p)
sc = os.path.join(ROOT, "seq-collection_amd", "sc")
settings = [("--jobs=1", ["--jobs=1"]), ("default (two files in flight)", []), ("--jobs=2", ["--jobs=2"]), ("--jobs=4", ["--jobs=4"]), ("--jobs=8", ["--jobs=8"])]
walls = {name: [] for name, _ in settings}
want = None
subprocess.run([sc, "fq-count"] + many, capture_output=True)          # page cache, first process on the box
import random
rng = random.Random(20260104)
for rnd in range(6):
    order = list(settings)
    rng.shuffle(order)          # (a process that follows another one closely starts 0.2 s more slowly about every other time, whatever the two are)
    for name, arg in order:
        time.sleep(1.0)
        t = time.time(); r = subprocess.run([sc, "fq-count"] + arg + many, capture_output=True, text=True); dt = time.time() - t
        assert r.returncode == 0 and len(r.stdout.splitlines()) == 8, r.stderr
        want = want or r.stdout
        assert r.stdout == want
        walls[name].append(round(dt, 4))
total = 8 * data.size
for name, _ in settings:
    w = walls[name]
    print(json.dumps({"setting": name, "files": 8, "inflated_bytes_each": int(data.size), "wall_s_min": min(w), "wall_s_median": statistics.median(w), "walls_s": w,
                      "inflated_GBps_at_min": round(total / min(w) / 1e9, 2), "what": "sc fq-count over 8 gzip files, process start included"}), flush=True)
for p in many: os.remove(p)
