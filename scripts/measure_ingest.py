#!/usr/bin/env python3
"""GPU-box measurement of the ingest paths around the hot kernel (not the roofline number):
   host buffer -> HBM (PCIe-inclusive), plain file via pread + pinned double buffering, and gzip input
   (host zlib inflate overlapped with copy + scan: BASELINE.json configs[3]).  Every row is checked against the
   generator tallies.  Writes one JSON object per line."""
import gzip, json, os, sys, time, zlib
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "seq-collection_amd", "pyhost"))
import numpy as np
import scfq

nbytes = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_000_000_000
tmp = sys.argv[2] if len(sys.argv) > 2 else "/tmp"
seed = 20260101
plan = scfq.synth_plan(0, seed, nbytes)
t = time.time(); data, info = scfq.synth_host(0, seed, plan.records); gen_s = time.time() - t
expect = (plan.records, info.gc_bases, info.n_bases, info.bases)
def check(c): assert (c.reads, c.gc_bases, c.n_bases, c.bases) == expect, (c.reads, c.gc_bases, c.n_bases, c.bases, expect)
rows = []
def row(name, **kw):
    kw["path"] = name; rows.append(kw); print(json.dumps(kw), flush=True)

# host buffer (pageable numpy) -> staged H2D + scan
scfq.count_host(data[:1 << 20])
for it in range(2):
    t = time.time(); c = scfq.count_host(data, flags=scfq.SCFQ_TIMING); dt = time.time() - t
check(c); tm = scfq.last_timing()
row("host buffer (pageable memcpy into pinned ring -> H2D -> scan)", bytes=data.size, wall_s=round(dt, 4), GBps=round(data.size / dt / 1e9, 2),
    scan_kernel_ms=round(tm.scan_kernel_ms, 3), host_fill_ms=round(tm.host_fill_ms, 1), h2d_copy_ms=round(tm.h2d_ms, 2), ingest_wall_ms=round(tm.ingest_wall_ms, 1))

plain = os.path.join(tmp, "scfq_synth.fq")
data.tofile(plain)
for it in range(2):
    t = time.time(); c = scfq.count_file(plain, flags=scfq.SCFQ_TIMING); dt = time.time() - t
check(c); tm = scfq.last_timing()
row("plain file, page cache (pread -> pinned -> H2D -> scan)", bytes=data.size, wall_s=round(dt, 4), GBps=round(data.size / dt / 1e9, 2),
    scan_kernel_ms=round(tm.scan_kernel_ms, 3), host_fill_ms=round(tm.host_fill_ms, 1), h2d_copy_ms=round(tm.h2d_ms, 2), ingest_wall_ms=round(tm.ingest_wall_ms, 1))

# gzip: (i) one member (first 512 MB), (ii) 64 MiB-per-member concatenation of the whole image
one = data[: min(data.size, 512 << 20)]
cut = int(np.flatnonzero(one[-4096:] == 10)[-1]) + one.size - 4096 + 1
one = one[:cut]
t = time.time(); blob = zlib.compress(one.tobytes(), 6); comp_s = time.time() - t
# zlib.compress gives a zlib stream; use gzip container for gzopen
gz1 = os.path.join(tmp, "scfq_one_member.fq.gz")
with gzip.open(gz1, "wb", compresslevel=6) as f: f.write(one.tobytes())
oc = scfq.count_host(one); 
# (two calls, the second reported, as for the BGZF row below: the first of a new size also grows the device buffers the context keeps)
firsts = {}
for it in range(2):
    t = time.time(); c = scfq.count_file(gz1, flags=scfq.SCFQ_TIMING); dt = time.time() - t
    firsts.setdefault("one", round(dt, 3))
tm = scfq.last_timing()
assert (c.reads, c.gc_bases, c.n_bases, c.bases) == (oc.reads, oc.gc_bases, oc.n_bases, oc.bases)
row("gzip -6, one member (device-side inflate; compressed bytes H2D)", inflated_bytes=one.size, gz_bytes=os.path.getsize(gz1), wall_s=round(dt, 3), first_call_wall_s=firsts["one"],
    inflated_GBps=round(one.size / dt / 1e9, 3), host_inflate_ms=round(tm.host_fill_ms, 1), h2d_copy_ms=round(tm.h2d_ms, 2), scan_kernel_ms=round(tm.scan_kernel_ms, 3),
    ingest_wall_ms=round(tm.ingest_wall_ms, 1), overlap_efficiency=round(tm.host_fill_ms / tm.ingest_wall_ms, 4))

member = 64 << 20
parts = [data[i:i + member] for i in range(0, data.size, member)]
with ThreadPoolExecutor(32) as ex:
    blobs = list(ex.map(lambda a: gzip.compress(a.tobytes(), 6), parts))
blobs_gz = blobs
gzm = os.path.join(tmp, "scfq_multi_member.fq.gz")
with open(gzm, "wb") as f:
    for b in blobs: f.write(b)
for it in range(2):
    t = time.time(); c = scfq.count_file(gzm, flags=scfq.SCFQ_TIMING); dt = time.time() - t
    firsts.setdefault("multi", round(dt, 3))
check(c); tm = scfq.last_timing()
row("gzip -6, %d concatenated 64 MiB members (device-side inflate)" % len(parts), inflated_bytes=data.size, gz_bytes=os.path.getsize(gzm), wall_s=round(dt, 3), first_call_wall_s=firsts["multi"],
    inflated_GBps=round(data.size / dt / 1e9, 3), host_inflate_ms=round(tm.host_fill_ms, 1), h2d_copy_ms=round(tm.h2d_ms, 2), scan_kernel_ms=round(tm.scan_kernel_ms, 3),
    ingest_wall_ms=round(tm.ingest_wall_ms, 1), overlap_efficiency=round(tm.host_fill_ms / tm.ingest_wall_ms, 4))
# BGZF (what `bgzip` writes): 64 KiB blocks with their compressed size in the header -> block-parallel inflate
import struct
def bgzf_block(b):
    co = zlib.compressobj(6, zlib.DEFLATED, -15); payload = co.compress(b) + co.flush(); bs = 18 + len(payload) + 8
    return b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bs - 1) + payload + struct.pack("<II", zlib.crc32(b) & 0xFFFFFFFF, len(b))
def bgzf_span(a):
    raw = a.tobytes(); return b"".join(bgzf_block(raw[i:i + 0xff00]) for i in range(0, len(raw), 0xff00))
with ThreadPoolExecutor(32) as ex:
    blobs = list(ex.map(bgzf_span, parts))
bgz = os.path.join(tmp, "scfq_bgzf.fq.gz")
with open(bgz, "wb") as f:
    for b in blobs: f.write(b)
    f.write(bgzf_block(b""))
for it in range(2):
    t = time.time(); c = scfq.count_file(bgz, flags=scfq.SCFQ_TIMING); dt = time.time() - t
check(c); tm = scfq.last_timing()
row("BGZF (bgzip layout), default path: compressed bytes H2D || device-side inflate || scan (host_inflate_ms = host copy of compressed bytes)", inflated_bytes=data.size, gz_bytes=os.path.getsize(bgz), wall_s=round(dt, 3),
    inflated_GBps=round(data.size / dt / 1e9, 3), host_inflate_ms=round(tm.host_fill_ms, 1), h2d_copy_ms=round(tm.h2d_ms, 2), scan_kernel_ms=round(tm.scan_kernel_ms, 3),
    ingest_wall_ms=round(tm.ingest_wall_ms, 1), overlap_efficiency=round(tm.host_fill_ms / tm.ingest_wall_ms, 4))
os.environ["SCFQ_NO_BGZF"] = "1"
t = time.time(); c = scfq.count_file(bgz, flags=scfq.SCFQ_TIMING); dt = time.time() - t
check(c)
row("same BGZF file through serial gzread (SCFQ_NO_BGZF=1)", inflated_bytes=data.size, wall_s=round(dt, 3), inflated_GBps=round(data.size / dt / 1e9, 3))
# (many files by --jobs: scripts/measure_jobs.py — the settings in shuffled order, a second apart: processes that follow one another
# closely wait for the driver's wipe of what the one before freed, which hit whichever setting came later in a fixed order)
many = []
for p in [plain, gz1, gzm, bgz] + many:
    os.remove(p)
