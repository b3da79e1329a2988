#!/bin/bash
# A/B of tuning knobs on the GPU box; each line: knobs -> achieved GB/s of the scan kernel
run() { env "$@" python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['roofline']['achieved'], d['roofline']['avg_kernel_ms'], d['value'])"; }
for r in 2 3; do for nt in 0 1; do run SCFQ_RING=$r SCFQ_NT=$nt; done; done
