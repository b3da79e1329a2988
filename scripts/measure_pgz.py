#!/usr/bin/env python3
"""ONE gzip member of several GB through scfq_count_file: the parallel reader (scfq_pgz.hpp) against the serial readers.
The member is built in parallel the way `pigz -i` does (independently deflated chunks closed with a sync flush, one
header, one CRC-32 / ISIZE trailer): a valid single member that gzip -d accepts, made in seconds instead of minutes.
usage: measure_pgz.py [bytes=6e9]"""
import json, os, struct, subprocess, sys, time, zlib
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "seq-collection_amd", "pyhost"))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import scfq
    scfq.count_file(os.path.join(ROOT, "tests", "golden", "dup.fq.gz"))
    best = None
    for _ in range(int(sys.argv[3])):
        t = time.time(); c = scfq.count_file(sys.argv[2], flags=scfq.SCFQ_TIMING); dt = time.time() - t
        best = dt if best is None or dt < best else best
    print(json.dumps({"counts": [c.reads, c.gc_bases, c.n_bases, c.bases], "bytes": c.input_bytes, "wall_s": round(best, 3),
                      "inflated_GBps": round(c.input_bytes / best / 1e9, 2)}))
    sys.exit(0)
import scfq
nbytes = int(float(sys.argv[1])) if len(sys.argv) > 1 else int(6e9)
plan = scfq.synth_plan(0, 20260101, nbytes)
data, info = scfq.synth_host(0, 20260101, plan.records)
step = 8 << 20
def piece(i):
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    last = i + step >= data.size
    raw = data[i:i + step].tobytes()
    return co.compress(raw) + co.flush(zlib.Z_FINISH if last else zlib.Z_SYNC_FLUSH), zlib.crc32(raw), len(raw)
with ThreadPoolExecutor(16) as ex:
    parts = list(ex.map(piece, range(0, data.size, step)))
crc = 0
for _, c, n in parts:
    crc = zlib.crc32(b"", crc) if n == 0 else crc
# combine the piece CRCs (zlib has no crc32_combine in python): recompute over the whole buffer once
crc = zlib.crc32(data.tobytes())
path = "/tmp/scfq_pgz_big.fq.gz"
with open(path, "wb") as f:
    f.write(b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03")
    for blob, _, _ in parts: f.write(blob)
    f.write(struct.pack("<II", crc & 0xFFFFFFFF, data.size & 0xFFFFFFFF))
expect = [plan.records, info.gc_bases, info.n_bases, info.bases]
for mode, env, reps in (("own, parallel (default)", {}, 3), ("own, one thread", {"SCFQ_PGZ": "0"}, 1), ("zlib", {"SCFQ_INFLATE": "zlib"}, 1)):
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", path, str(reps)], env=dict(os.environ, **env), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    j = json.loads(r.stdout.strip().splitlines()[-1])
    assert j.pop("counts") == expect
    j.update({"path": "one gzip member (level 6)", "inflate": mode, "gz_bytes": os.path.getsize(path)})
    print(json.dumps(j), flush=True)
