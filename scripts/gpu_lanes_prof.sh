#!/bin/bash
# Where a round of the lane-parallel symbol loop spends its cycles (a -DSCFQ_LPROF build of the library; run on the GPU box):
# shader-clock stamps around window wait, decode, walk, the pending store (which waits for the previous round's load) and emit
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
S=$R/seq-collection_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DSCFQ_LPROF -o /tmp/libscfq_lprof.so \
  $S/scfq_api.hip $S/scfq_host.cpp $S/scfq_synth.hip $S/scfq_dedup.hip $S/scfq_meta.cpp $S/scfq_comm.cpp -lz -lpthread -ldl 2>/dev/null
for LOOP in ${SCFQ_PROF_LOOPS:-lanes dense}; do
echo "== SCFQ_INFLATE_LOOP=$LOOP"
SCFQ_INFLATE_LOOP=$LOOP SCFQ_LIB_OVERRIDE=/tmp/libscfq_lprof.so python - <<PY
import sys, zlib, struct
sys.path.insert(0, "$R/seq-collection_amd/pyhost")
import scfq
plan = scfq.synth_plan(0, 20260101, ${1:-256} << 20)
data, info = scfq.synth_host(0, 20260101, plan.records)
raw = data.tobytes()
def blk(b, level):
    co = zlib.compressobj(level, zlib.DEFLATED, -15); p = co.compress(b) + co.flush(); bs = 18 + len(p) + 8
    return b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bs - 1) + p + struct.pack("<II", zlib.crc32(b) & 0xFFFFFFFF, len(b))
for level in (6,):
    img = b"".join(blk(raw[i:i + 0xff00], level) for i in range(0, len(raw), 0xff00))
    for rep in range(2):
        out = scfq.debug_bgzf_inflate(img, len(raw))
    print("level", level, "ratio", round(len(raw) / len(img), 2), "ok", bytes(out) == raw, flush=True)
PY
done
