#!/bin/bash
# timing-only ablation of the scan kernel (diagnostic builds; counters are wrong on purpose)
one() { python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-verify | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['roofline']['achieved'], d['roofline']['avg_kernel_ms'])"; }
one full
for n in 1 2 3; do SCFQ_LIB_OVERRIDE=$PWD/seq-collection_amd/ablate/libsc_fqcount_hip_$n.so one ablate$n; done
