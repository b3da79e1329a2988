#!/usr/bin/env python3
"""How the compressed bytes reach the pinned ring (device inflate paths): SCFQ_COPY_THREADS x SCFQ_COPY_PREAD, a 2 GB BGZF file and a 2 GB
gzip member, six calls each in one process: every wall time, not the best.   usage: measure_copy_variants.py [tmpdir]"""
import json, os, struct, subprocess, sys, time, zlib
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "seq-collection_amd", "pyhost"))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import scfq
    path = sys.argv[2]
    scfq.count_file(os.path.join(ROOT, "tests", "golden", "dup.fq.gz"))
    walls = []
    for _ in range(6):
        t = time.time(); c = scfq.count_file(path, flags=scfq.SCFQ_TIMING); walls.append(round((time.time() - t) * 1e3, 1))
    print(json.dumps({"walls_ms": walls, "reads": c.reads}))
    sys.exit(0)
import scfq
tmp = sys.argv[1] if len(sys.argv) > 1 else "/tmp"
plan = scfq.synth_plan(0, 20260101, int(2e9))
data, info = scfq.synth_host(0, 20260101, plan.records)

def block(b):
    co = zlib.compressobj(6, zlib.DEFLATED, -15); payload = co.compress(b) + co.flush()
    return b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 18 + len(payload) + 8 - 1) + payload + struct.pack("<II", zlib.crc32(b) & 0xFFFFFFFF, len(b))

def span(i):
    a = data[i:i + (32 << 20)]
    return b"".join(block(a[o:o + 65280].tobytes()) for o in range(0, a.size, 65280))

def piece(i):
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    chunk = data[i:i + (64 << 20)]
    return co.compress(chunk.tobytes()) + co.flush(zlib.Z_FINISH if i + (64 << 20) >= data.size else zlib.Z_SYNC_FLUSH)
with ThreadPoolExecutor(16) as ex:
    spans = list(ex.map(span, range(0, data.size, 32 << 20)))
    parts = list(ex.map(piece, range(0, data.size, 64 << 20)))
bg = os.path.join(tmp, "scfq_cv_bgzf.fq.gz")
with open(bg, "wb") as f:
    for s_ in spans: f.write(s_)
    f.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
gz = os.path.join(tmp, "scfq_cv_member.fq.gz")
crc = 0
for o in range(0, data.size, 64 << 20): crc = zlib.crc32(data[o:o + (64 << 20)], crc)
with open(gz, "wb") as f:
    f.write(b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03")
    for b in parts: f.write(b)
    f.write(struct.pack("<II", crc & 0xFFFFFFFF, data.size & 0xFFFFFFFF))
os.sync()
for name, path in (("bgzf", bg), ("gzip member", gz)):
    for env in ({"SCFQ_COPY_PREAD": "0", "SCFQ_COPY_THREADS": "8"}, {"SCFQ_COPY_PREAD": "1", "SCFQ_COPY_THREADS": "4"}, {"SCFQ_COPY_PREAD": "1", "SCFQ_COPY_THREADS": "8"},
                {"SCFQ_COPY_PREAD": "1", "SCFQ_COPY_THREADS": "12"}, {"SCFQ_COPY_PREAD": "0", "SCFQ_COPY_THREADS": "12"}):
        r = subprocess.run([sys.executable, __file__, "--child", path], capture_output=True, text=True, env=dict(os.environ, **env))
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        print(json.dumps({"file": name, "env": env, "result": json.loads(line[-1]) if line else r.stderr[-300:]}), flush=True)
os.remove(bg); os.remove(gz)
