#!/bin/bash
# correctness + 3 repeated bench runs (variance) on one box
python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -2
for i in 1 2 3; do python bench.py --steps 30 --warmup 3 --no-cpu-baseline $BENCH_ARGS | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['avg_kernel_ms'])"; done
