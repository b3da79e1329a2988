#!/bin/bash
# Run on the GPU box (via gpurun): smoke + a short bench; results under gpurun_out/
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
python bench.py --steps ${STEPS:-10} --warmup 2 ${BENCH_ARGS} 2>&1 | tail -5 | tee gpurun_out/bench_latest.json
nproc; free -g | head -2
