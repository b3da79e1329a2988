#!/usr/bin/env python3
"""BGZF ingest: device-side inflate vs the host paths, same file, one box (quick: level-1 blocks built in threads).
usage: measure_bgzf_device.py [bytes]"""
import json, os, struct, subprocess, sys, time, zlib
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "seq-collection_amd", "pyhost"))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import scfq
    path = sys.argv[2]
    scfq.count_file(os.path.join(ROOT, "tests", "golden", "dup.fq.gz"))
    best = None
    for _ in range(4):
        t = time.time(); c = scfq.count_file(path, flags=scfq.SCFQ_TIMING); dt = time.time() - t
        tm = scfq.last_timing()
        if best is None or dt < best[0]:
            best = (dt, tm.host_fill_ms, tm.h2d_ms, tm.scan_kernel_ms, tm.ingest_wall_ms)
    print(json.dumps({"counts": [c.reads, c.gc_bases, c.n_bases, c.bases], "bytes": c.input_bytes, "wall_s": round(best[0], 4),
                      "inflated_GBps": round(c.input_bytes / best[0] / 1e9, 2), "host_fill_ms": round(best[1], 1),
                      "h2d_copy_ms": round(best[2], 2), "scan_kernel_ms": round(best[3], 3), "ingest_wall_ms": round(best[4], 1)}))
    sys.exit(0)
import scfq
nbytes = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1 << 30
level = int(sys.argv[2]) if len(sys.argv) > 2 else 6
plan = scfq.synth_plan(0, 20260101, nbytes)
data, info = scfq.synth_host(0, 20260101, plan.records)
def bgzf_block(b):
    co = zlib.compressobj(level, zlib.DEFLATED, -15); payload = co.compress(b) + co.flush(); bs = 18 + len(payload) + 8
    return b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bs - 1) + payload + struct.pack("<II", zlib.crc32(b) & 0xFFFFFFFF, len(b))
def span(a):
    raw = a.tobytes(); return b"".join(bgzf_block(raw[i:i + 0xff00]) for i in range(0, len(raw), 0xff00))
parts = [data[i:i + (0xff00 * 256)] for i in range(0, data.size, 0xff00 * 256)]
with ThreadPoolExecutor(16) as ex:
    blobs = list(ex.map(span, parts))
path = "/tmp/scfq_bgzf_dev.fq.gz"
with open(path, "wb") as f:
    for b in blobs: f.write(b)
expect = [plan.records, info.gc_bases, info.n_bases, info.bases]
for mode, env in (("device", {"SCFQ_BGZF_DEVICE": "1"}), ("host, own inflate", {"SCFQ_BGZF_DEVICE": "0"}),
                  ("host, zlib", {"SCFQ_BGZF_DEVICE": "0", "SCFQ_INFLATE": "zlib"})):
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", path], env=dict(os.environ, **env), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    j = json.loads(r.stdout.strip().splitlines()[-1])
    assert j.pop("counts") == expect
    j.update({"path": "BGZF level %d" % level, "inflate": mode, "gz_bytes": os.path.getsize(path)})
    print(json.dumps(j), flush=True)
