#!/bin/bash
# Where the device-side BGZF inflate spends its time: builds of the library with parts of the kernel removed
# (SCFQ_DABLATE: 1 no match copies, 2 no CRC, 4 no literal stores, 8 / 16 ten extra scalar / vector instructions per symbol; the counts are then wrong or the call fails, only the
# wall time is read), same file, same box.   usage: gpu_dinflate_ablate.sh [bytes]   (run on the GPU box)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
N=${1:-2e9}
OUT=$R/gpurun_out/dinflate_ablate; mkdir -p $OUT
S=$R/seq-collection_amd/csrc
for A in ${ABLATIONS:-0 2 3 7 8 16}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DSCFQ_DABLATE=$A -o /tmp/libscfq_dab$A.so \
    $S/scfq_api.hip $S/scfq_host.cpp $S/scfq_synth.hip $S/scfq_dedup.hip $S/scfq_meta.cpp $S/scfq_comm.cpp -lz -lpthread -ldl 2>/dev/null &
done
wait
python $R/scripts/measure_bgzf_device.py $N > $OUT/base.jsonl
for A in ${ABLATIONS:-0 2 3 7 8 16}; do
  echo "ablate $A" >> $OUT/ablate.txt
  SCFQ_LIB_OVERRIDE=/tmp/libscfq_dab$A.so SCFQ_BGZF_DEVICE=1 python - >> $OUT/ablate.txt 2>&1 <<PY
import sys, time, os
sys.path.insert(0, "$R/seq-collection_amd/pyhost")
import scfq
best = 1e9
for _ in range(4):
    t = time.time()
    try:
        scfq.count_file("/tmp/scfq_bgzf_dev.fq.gz")
    except scfq.ScfqError as e:
        pass
    best = min(best, time.time() - t)
print("  wall_ms", round(best * 1e3, 1))
PY
done
cat $OUT/ablate.txt
