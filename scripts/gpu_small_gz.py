#!/usr/bin/env python3
"""warm wall of small gzip members (device path), by size: the latency floor of `sc fq-count small.fq.gz` inside one process"""
import json, os, sys, time, zlib
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "seq-collection_amd", "pyhost"))
import scfq
for nbytes in (64 << 20, 256 << 20, 512 << 20, 1 << 30):
    plan = scfq.synth_plan(0, 20260101, nbytes)
    data, info = scfq.synth_host(0, 20260101, plan.records)
    step = 64 << 20
    cuts = list(range(0, data.size, step))
    def piece(i):
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        chunk = data[cuts[i]:cuts[i] + step].tobytes()
        return co.compress(chunk) + co.flush(zlib.Z_FINISH if i == len(cuts) - 1 else zlib.Z_SYNC_FLUSH)
    with ThreadPoolExecutor(16) as ex:
        parts = list(ex.map(piece, range(len(cuts))))
    crc = 0
    for c0 in cuts: crc = zlib.crc32(data[c0:c0 + step], crc)
    path = "/tmp/scfq_small.fq.gz"
    with open(path, "wb") as f:
        f.write(b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03")
        for b in parts: f.write(b)
        f.write(int(crc & 0xFFFFFFFF).to_bytes(4, "little") + int(data.size & 0xFFFFFFFF).to_bytes(4, "little"))
    walls = []
    for rep in range(5):
        t = time.time(); c = scfq.count_file(path); walls.append(time.time() - t)
        assert (c.reads, c.gc_bases, c.n_bases, c.bases) == (plan.records, info.gc_bases, info.n_bases, info.bases)
    print(json.dumps({"inflated_bytes": int(data.size), "gz_bytes": os.path.getsize(path), "first_call_ms": round(walls[0] * 1e3, 2), "warm_ms": round(min(walls[1:]) * 1e3, 2),
                      "warm_GBps": round(data.size / min(walls[1:]) / 1e9, 2)}), flush=True)
