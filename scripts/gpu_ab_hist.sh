#!/bin/bash
# A/B of the speculative histogram kernel builds: the default library vs seq-collection_amd/ablate/libsc_fqcount_hip_q*.so
# (workgroup shapes: -DSCFQ_QWAVES / -DSCFQ_QREP; qold = the library of the commit before).  Every build first passes the
# partial / histogram parity check against the oracle (tests/_variant_check.py), then runs bench.py for K3 (Illumina, long reads),
# K3 + K4, K4 and K1.  usage: scripts/gpu_ab_hist.sh <out-subdir>
OUT=gpurun_out/${1:-ab}; mkdir -p $OUT
run() { # name, lib, extra args
  SCFQ_LIB_OVERRIDE=$2 timeout -k 10 120 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --ingest-bytes 0 $3 2>$OUT/$1.err | tail -1 > $OUT/$1.json || { echo "$1 FAILED"; tail -3 $OUT/$1.err; return; }
  python -c "import json;d=json.load(open('$OUT/$1.json'));print('%-34s ms/step %.4f  kernel %.4f ms  frac %.4f  fold %.4f' % ('$1',d['ms_per_step'],d['roofline'].get('avg_kernel_ms'),d['roofline']['frac'],d['roofline'].get('avg_fold_ms')))" | tee -a $OUT/summary.txt
}
for f in "" seq-collection_amd/ablate/libsc_fqcount_hip_q*.so; do
  n=default; lib=""
  if [ -n "$f" ]; then n=$(basename $f .so | sed 's/libsc_fqcount_hip_//'); lib=$PWD/$f; fi
  if SCFQ_LIB_OVERRIDE=$lib timeout -k 10 200 python tests/_variant_check.py 11 60 > $OUT/check_$n.log 2>&1; then echo "$n: parity ok" | tee -a $OUT/summary.txt; else echo "$n: PARITY FAILED" | tee -a $OUT/summary.txt; tail -5 $OUT/check_$n.log; continue; fi
  run ${n}_hist "$lib" "--flags 1"
  run ${n}_hist_nano "$lib" "--flags 1 --workload nanopore"
  run ${n}_hist_struct "$lib" "--flags 3"
  run ${n}_struct "$lib" "--flags 2"
  run ${n}_k1 "$lib" ""
done
