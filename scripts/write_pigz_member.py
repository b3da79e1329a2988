#!/usr/bin/env python3
"""Write the synthetic Illumina stream (BASELINE configs[1], seed 20260101) as ONE gzip member the way pigz does it: zlib level-6 raw
deflate of 64 MiB pieces on 16 threads, every piece but the last ended with a sync flush, CRC-32 / ISIZE of the whole input.
usage: write_pigz_member.py <inflated bytes> <out.gz> [--plain out.fq] [--bgzf]     prints one JSON line: records, tallies, sizes.
--bgzf: the same bytes as a BGZF file instead (members of 65280 bytes, level 6, with the empty end-of-file member)."""
import json, os, sys, time, zlib
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "seq-collection_amd", "pyhost"))
import torch      # (before the library: both then share one HIP runtime)
import scfq

nbytes = int(float(sys.argv[1])); out = sys.argv[2]
plan = scfq.synth_plan(0, 20260101, nbytes)
if torch.cuda.is_available():
    buf = torch.empty(plan.bytes + 4096, dtype=torch.uint8, device="cuda")
    info = scfq.synth_device(0, 20260101, plan.records, buf.data_ptr(), plan.bytes)
    data = buf[:plan.bytes].cpu().numpy()
    del buf
    torch.cuda.empty_cache()
else:
    data, info = scfq.synth_host(0, 20260101, plan.records)
if "--plain" in sys.argv:
    data.tofile(sys.argv[sys.argv.index("--plain") + 1])
t0 = time.time()
if "--bgzf" in sys.argv:
    import struct
    def block(b):
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        payload = co.compress(b) + co.flush()
        return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 18 + len(payload) + 8 - 1) + payload +
                struct.pack("<II", zlib.crc32(b) & 0xFFFFFFFF, len(b)))
    def span(i):
        a = data[i:i + (32 << 20)]
        return b"".join(block(a[o:o + 65280].tobytes()) for o in range(0, a.size, 65280))
    with ThreadPoolExecutor(16) as ex, open(out, "wb") as f:
        for s_ in ex.map(span, range(0, data.size, 32 << 20)):
            f.write(s_)
        f.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
    print(json.dumps({"records": plan.records, "gc_bases": info.gc_bases, "n_bases": info.n_bases, "bases": info.bases, "inflated_bytes": int(data.size),
                      "gz_bytes": os.path.getsize(out), "compress_s": round(time.time() - t0, 1), "format": "bgzf"}))
    sys.exit(0)
step = 64 << 20
cuts = list(range(0, data.size, step))
def piece(i):
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    return co.compress(data[cuts[i]:cuts[i] + step].tobytes()) + co.flush(zlib.Z_FINISH if i == len(cuts) - 1 else zlib.Z_SYNC_FLUSH)
with ThreadPoolExecutor(16) as ex:
    parts = list(ex.map(piece, range(len(cuts))))
# (crc32 of the whole = the pieces' combined; zlib's Python binding has no crc32_combine: fold sequentially only when small)
crc = 0
for c0 in cuts:
    crc = zlib.crc32(data[c0:c0 + step], crc)
with open(out, "wb") as f:
    f.write(b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03")
    for b in parts:
        f.write(b)
    f.write(int(crc & 0xFFFFFFFF).to_bytes(4, "little") + int(data.size & 0xFFFFFFFF).to_bytes(4, "little"))
print(json.dumps({"records": plan.records, "gc_bases": info.gc_bases, "n_bases": info.n_bases, "bases": info.bases, "inflated_bytes": int(data.size),
                  "gz_bytes": os.path.getsize(out), "compress_s": round(time.time() - t0, 1)}))
