#!/usr/bin/env python3
"""BGZF device inflate on FASTQ that looks like a NovaSeq run rather than like the synthetic benchmark input: Illumina headers,
binned qualities in long runs of 'F' (distance-1 matches, which the lane-parallel symbol loop takes one at a time), both symbol
loops on the same file.  usage: measure_real_like_bgzf.py     (GPU box)"""
import sys, zlib, struct, time, numpy as np
sys.path.insert(0, "seq-collection_amd/pyhost")
import scfq
rng = np.random.default_rng(11)
n = 1_300_000
L = 150
seq = rng.choice(np.frombuffer(b"ACGT", np.uint8), (n, L))
# NovaSeq-like binned qualities: long runs of F, occasional : , #
q = np.full((n, L), ord("F"), np.uint8)
m = rng.random((n, L))
q[m < 0.06] = ord(":"); q[m < 0.025] = ord(","); q[m < 0.004] = ord("#")
recs = []
hdr = [("@A00123:45:HXXXXXXXX:%d:%d:%d:%d 1:N:0:ACGTACGT+TGCATGCA\n" % (1 + i % 4, 1101 + (i // 4000) % 78, 1000 + (i * 37) % 30000, 1000 + (i * 91) % 30000)).encode() for i in range(n)]
out = bytearray()
for i in range(n):
    out += hdr[i]; out += seq[i].tobytes(); out += b"\n+\n"; out += q[i].tobytes(); out += b"\n"
raw = bytes(out)
print("bytes", len(raw), flush=True)
def blk(b):
    co = zlib.compressobj(6, zlib.DEFLATED, -15); p = co.compress(b) + co.flush(); bs = 18 + len(p) + 8
    return b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bs - 1) + p + struct.pack("<II", zlib.crc32(b) & 0xFFFFFFFF, len(b))
from concurrent.futures import ThreadPoolExecutor
chunks = [raw[i:i + 0xff00 * 64] for i in range(0, len(raw), 0xff00 * 64)]
with ThreadPoolExecutor(16) as ex:
    blobs = list(ex.map(lambda c: b"".join(blk(c[i:i + 0xff00]) for i in range(0, len(c), 0xff00)), chunks))
img = b"".join(blobs)
open("/tmp/real_like.fq.gz", "wb").write(img)
print("ratio", round(len(raw) / len(img), 2), flush=True)
import os
for loop in ("lanes", "serial"):
    os.environ["X"] = loop
import subprocess
for loop in ("lanes", "serial"):
    r = subprocess.run([sys.executable, "-c", """
import sys, time
sys.path.insert(0, "seq-collection_amd/pyhost")
import scfq
scfq.count_file("tests/golden/dup.fq.gz")
best = 1e9
for _ in range(4):
    t = time.time(); c = scfq.count_file("/tmp/real_like.fq.gz"); best = min(best, time.time() - t)
print("%s: %.1f ms  %.1f GB/s inflated  reads %d" % ("LOOP", best * 1e3, c.input_bytes / best / 1e9, c.reads))
""".replace("LOOP", loop)], env=dict(os.environ, SCFQ_INFLATE_LOOP=loop), capture_output=True, text=True)
    print(r.stdout.strip(), r.stderr.strip()[-300:])
