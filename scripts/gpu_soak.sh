#!/bin/bash
# Soak: repeated parity suites + fold stress (many ranges, concurrent sessions) to catch intermittent failures
# (inter-workgroup visibility in the fused fold, ring/ticket reuse, context pool).
set -e
for i in 1 2 3 4 5; do python -m pytest tests/test_gpu_parity.py tests/test_gpu_variants.py -q -m gpu -x 2>&1 | tail -1; done
python - <<'PY'
import os, sys, threading
sys.path.insert(0, "seq-collection_amd/pyhost")
import torch, scfq
plan = scfq.synth_plan(0, 7, 1_000_000_000)
buf = torch.empty(plan.bytes + 4096, dtype=torch.uint8, device="cuda")
info = scfq.synth_device(0, 7, plan.records, buf.data_ptr(), plan.bytes)
want = (plan.records, info.gc_bases, info.n_bases, info.bases)
bad = 0
for rep in range(300):
    c = scfq.count_device(buf.data_ptr(), plan.bytes)
    if (c.reads, c.gc_bases, c.n_bases, c.bases) != want: bad += 1
print("300 repeated 1 GB scans, mismatches:", bad)
# concurrent sessions from 8 host threads on one device (context pool)
errs = []
def worker(k):
    for rep in range(40):
        off = 4096 * k
        c = scfq.count_device(buf.data_ptr(), plan.bytes)
        if (c.reads, c.gc_bases, c.n_bases, c.bases) != want: errs.append((k, rep))
ts = [threading.Thread(target=worker, args=(k,)) for k in range(8)]
[t.start() for t in ts]; [t.join() for t in ts]
print("8 threads x 40 concurrent scans, mismatches:", len(errs))
assert bad == 0 and not errs
PY
for tpr in 1 2 5; do SCFQ_TILES_PER_RANGE=$tpr python - <<'PY'
import os, sys
sys.path.insert(0, "seq-collection_amd/pyhost")
import torch, scfq
plan = scfq.synth_plan(1, 9, 600_000_000)
buf = torch.empty(plan.bytes + 4096, dtype=torch.uint8, device="cuda")
info = scfq.synth_device(1, 9, plan.records, buf.data_ptr(), plan.bytes)
for rep in range(20):
    c = scfq.count_device(buf.data_ptr() , plan.bytes, flags=scfq.SCFQ_STRUCT_CHECK)
    assert (c.reads, c.gc_bases, c.n_bases, c.bases, c.bad_at, c.bad_plus) == (plan.records, info.gc_bases, info.n_bases, info.bases, 0, 0), rep
print("tiles_per_range", os.environ["SCFQ_TILES_PER_RANGE"], "ok (", (plan.bytes + 4095) // 4096 // int(os.environ["SCFQ_TILES_PER_RANGE"]), "ranges )")
PY
done
echo SOAK OK
# speculative K3 under repetition and concurrency: the verify / redo hand-off and the shared workgroup histogram
python - <<'PY'
import sys, threading
sys.path.insert(0, "seq-collection_amd/pyhost")
import torch, scfq
plan = scfq.synth_plan(0, 11, 1_000_000_000)
buf = torch.empty(plan.bytes + 4096, dtype=torch.uint8, device="cuda")
info = scfq.synth_device(0, 11, plan.records, buf.data_ptr(), plan.bytes)
ref = scfq.count_device(buf.data_ptr(), plan.bytes, flags=scfq.SCFQ_QUAL_HIST | scfq.SCFQ_HIST_EXACT)
want = list(ref.qual_hist)
assert sum(want) == info.bases
bad = 0
for rep in range(100):
    c = scfq.count_device(buf.data_ptr(), plan.bytes, flags=scfq.SCFQ_QUAL_HIST)
    bad += list(c.qual_hist) != want
print("100 repeated 1 GB speculative histograms, mismatches:", bad)
errs = []
def worker(k):
    for rep in range(15):
        c = scfq.count_device(buf.data_ptr(), plan.bytes, flags=scfq.SCFQ_QUAL_HIST | (scfq.SCFQ_STRUCT_CHECK if k & 1 else 0))
        if list(c.qual_hist) != want: errs.append((k, rep))
ts = [threading.Thread(target=worker, args=(k,)) for k in range(6)]
[t.start() for t in ts]; [t.join() for t in ts]
print("6 threads x 15 concurrent speculative histograms, mismatches:", len(errs))
# fq-dedup repeated (pool reuse) and the device inflate on a BGZF image, repeated
a, _ = scfq.synth_host(0, 13, scfq.synth_plan(0, 13, 64 << 20).records)
import numpy as np
two = np.concatenate([a, a])
first = scfq.dedup_host(two)[0]
for rep in range(10):
    assert scfq.dedup_host(two)[0] == first
assert first == a.tobytes()
print("10 repeated de-duplications identical")
# device-side BGZF inflate: one 512 MB file counted 20 times, then by 4 threads at once (own sessions, shared device)
import zlib, struct, os
from concurrent.futures import ThreadPoolExecutor
bplan = scfq.synth_plan(0, 17, 512 << 20)
bdata, binfo = scfq.synth_host(0, 17, bplan.records)
def bgzf_block(b):
    co = zlib.compressobj(6, zlib.DEFLATED, -15); p = co.compress(b) + co.flush(); bs = 18 + len(p) + 8
    return b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bs - 1) + p + struct.pack("<II", zlib.crc32(b) & 0xFFFFFFFF, len(b))
def span(x):
    raw = x.tobytes(); return b"".join(bgzf_block(raw[i:i + 0xff00]) for i in range(0, len(raw), 0xff00))
with ThreadPoolExecutor(16) as ex:
    blobs = list(ex.map(span, [bdata[i:i + 0xff00 * 256] for i in range(0, bdata.size, 0xff00 * 256)]))
bpath = "/tmp/scfq_soak.fq.gz"
with open(bpath, "wb") as f:
    for b in blobs: f.write(b)
bwant = (bplan.records, binfo.gc_bases, binfo.n_bases, binfo.bases)
def bcount():
    c = scfq.count_file(bpath); return (c.reads, c.gc_bases, c.n_bases, c.bases)
bbad = sum(bcount() != bwant for _ in range(20))
print("20 repeated device-inflated BGZF counts, mismatches:", bbad)
berrs = []
def bworker(k):
    for rep in range(5):
        if bcount() != bwant: berrs.append((k, rep))
ts = [threading.Thread(target=bworker, args=(k,)) for k in range(4)]
[t.start() for t in ts]; [t.join() for t in ts]
print("4 threads x 5 concurrent BGZF counts, mismatches:", len(berrs))
os.remove(bpath)
assert bbad == 0 and not berrs
assert bad == 0 and not errs
PY
