#!/usr/bin/env python3
"""gzip ingest (BASELINE configs[3]) with the library's own inflate vs zlib (SCFQ_INFLATE=zlib), same files, same box:
one gzip -6 member and a BGZF file.  usage: measure_gz.py [bytes] [tmpdir]; prints one JSON object per line."""
import json, os, struct, subprocess, sys, time, zlib
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "seq-collection_amd", "pyhost"))

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import scfq
    path = sys.argv[2]
    scfq.count_file(os.path.join(ROOT, "tests", "golden", "dup.fq.gz"))      # context + module warm-up
    best = None
    for _ in range(3):
        t = time.time(); c = scfq.count_file(path, flags=scfq.SCFQ_TIMING); dt = time.time() - t
        tm = scfq.last_timing()
        if best is None or dt < best[0]:
            best = (dt, tm.host_fill_ms, tm.h2d_ms, tm.scan_kernel_ms, tm.ingest_wall_ms)
    print(json.dumps({"counts": [c.reads, c.gc_bases, c.n_bases, c.bases], "bytes": c.input_bytes, "wall_s": round(best[0], 4),
                      "inflated_GBps": round(c.input_bytes / best[0] / 1e9, 3), "host_fill_ms": round(best[1], 1),
                      "h2d_copy_ms": round(best[2], 2), "scan_kernel_ms": round(best[3], 3), "ingest_wall_ms": round(best[4], 1)}))
    sys.exit(0)

import numpy as np
import scfq
nbytes = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1 << 30
tmp = sys.argv[2] if len(sys.argv) > 2 else "/tmp"
plan = scfq.synth_plan(0, 20260101, nbytes)
data, info = scfq.synth_host(0, 20260101, plan.records)
plain = os.path.join(tmp, "scfq_gzm.fq")
data.tofile(plain)
subprocess.check_call(["gzip", "-6", "-k", "-f", plain])
def bgzf_block(b):
    co = zlib.compressobj(6, zlib.DEFLATED, -15); payload = co.compress(b) + co.flush(); bs = 18 + len(payload) + 8
    return b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bs - 1) + payload + struct.pack("<II", zlib.crc32(b) & 0xFFFFFFFF, len(b))
def bgzf_span(a):
    raw = a.tobytes(); return b"".join(bgzf_block(raw[i:i + 0xff00]) for i in range(0, len(raw), 0xff00))
parts = [data[i:i + (0xff00 * 512)] for i in range(0, data.size, 0xff00 * 512)]
with ThreadPoolExecutor(16) as ex:
    blobs = list(ex.map(bgzf_span, parts))
bgz = os.path.join(tmp, "scfq_gzm_bgzf.fq.gz")
with open(bgz, "wb") as f:
    for b in blobs: f.write(b)
expect = [plan.records, info.gc_bases, info.n_bases, info.bases]
for label, path in (("gzip -6, one member", plain + ".gz"), ("BGZF", bgz)):
    for mode in ("own", "zlib") + (("device",) if label == "BGZF" else ("own, one thread",)):
        env = dict(os.environ, SCFQ_INFLATE="zlib" if mode == "zlib" else "own", SCFQ_BGZF_DEVICE="1" if mode == "device" else "0",
                   SCFQ_PGZ="0" if mode == "own, one thread" else "1")
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", path], env=env, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        j = json.loads(r.stdout.strip().splitlines()[-1])
        assert j.pop("counts") == expect
        j.update({"path": label, "inflate": mode, "gz_bytes": os.path.getsize(path), "host_cores": os.cpu_count()})
        print(json.dumps(j), flush=True)
