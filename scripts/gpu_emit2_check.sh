#!/bin/bash
# The paired-chunk output of symbol_loop_dense (-DSCFQ_DENSE_EMIT2=1 builds in seq-collection_amd/ablate/, made on the build host:
# `make -C seq-collection_amd emit2`: libsc_fqcount_hip_emit2.so, ..._emit2_lprof.so, libsc_fqcount_hip_lprof.so for the default form): the BGZF device tests on it, then
# cycle stamps of both forms over 256 MB of level-6 BGZF.
R=$(cd "$(dirname "$0")/.." && pwd); A=$R/seq-collection_amd/ablate
mkdir -p $R/gpurun_out/r03
SCFQ_LIB_OVERRIDE=$A/libsc_fqcount_hip_emit2.so timeout -k 5 120 python -m pytest $R/tests/test_gpu_bgzf_device.py -x -q -m gpu -k "not other_symbol_loops and not dedup" 2>&1 | tail -3
for L in lprof emit2_lprof; do
echo "== $L"
SCFQ_LIB_OVERRIDE=$A/libsc_fqcount_hip_$L.so timeout -k 5 120 python - <<PY 2>&1 | grep -E "dprof|ok" | head -3
import sys, zlib, struct
sys.path.insert(0, "$R/seq-collection_amd/pyhost")
import scfq
plan = scfq.synth_plan(0, 20260101, 256 << 20)
data, info = scfq.synth_host(0, 20260101, plan.records)
raw = data.tobytes()
def blk(b):
    co = zlib.compressobj(6, zlib.DEFLATED, -15); p = co.compress(b) + co.flush(); bs = 18 + len(p) + 8
    return b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bs - 1) + p + struct.pack("<II", zlib.crc32(b) & 0xFFFFFFFF, len(b))
img = b"".join(blk(raw[i:i + 0xff00]) for i in range(0, len(raw), 0xff00))
for rep in range(2):
    out = scfq.debug_bgzf_inflate(img, len(raw))
print("ok", bytes(out) == raw, flush=True)
PY
done
