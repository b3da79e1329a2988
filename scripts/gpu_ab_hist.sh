#!/bin/bash
# A/B of the speculative histogram kernel builds (make qwaves): default library vs ablate/libsc_fqcount_hip_q*.so
OUT=gpurun_out/${1:-ab}; mkdir -p $OUT
run() { # name, lib, extra args
  SCFQ_LIB_OVERRIDE=$2 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --flags 1 $3 2>/dev/null | tail -1 > $OUT/$1.json
  python -c "import json;d=json.load(open('$OUT/$1.json'));print('$1',d['ms_per_step'],d['roofline']['achieved'],d['roofline'].get('avg_kernel_ms'),d['roofline'].get('avg_fold_ms'))"
}
run q_default "" ""
for f in seq-collection_amd/ablate/libsc_fqcount_hip_q*.so; do n=$(basename $f .so); run $n $PWD/$f "--no-verify"; done
run q_default_nano "" "--workload nanopore"
for f in seq-collection_amd/ablate/libsc_fqcount_hip_q*.so; do n=$(basename $f .so); run ${n}_nano $PWD/$f "--workload nanopore --no-verify"; done
