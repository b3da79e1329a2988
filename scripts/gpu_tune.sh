#!/bin/bash
# A/B of tuning knobs on the GPU box; each line: knobs -> achieved GB/s of the scan kernel
run() { env "$@" python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['roofline']['achieved'], d['roofline']['avg_kernel_ms'], d['value'])"; }
run SCFQ_RING=2
for r in 3 4; do run SCFQ_RING=$r; done
for tpr in 50 200 400; do run SCFQ_TILES_PER_RANGE=$tpr; done
run SCFQ_RING=2
