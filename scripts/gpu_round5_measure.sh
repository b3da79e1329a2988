#!/bin/bash
# Round-5 measurement pass on the GPU box (run through gpurun in pieces).
# usage: scripts/gpu_round5_measure.sh <part>
#   part = bench | hist | prof | profhist | gz | gzkernels | sharded | cold | dedup | tests
TAG=r05; mkdir -p gpurun_out/$TAG
case "$1" in
bench)
  python3 bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/$TAG/bench.err | tail -1 | tee gpurun_out/$TAG/bench_n1.json | cut -c1-400 ;;
hist)
  for w in "" "--workload nanopore"; do for f in 1 3 2; do
    n=hist_f${f}$(echo $w | sed 's/--workload /_/')
    python bench.py --steps 10 --warmup 2 --no-cpu-baseline --ingest-bytes 0 --flags $f $w 2>/dev/null | tail -1 > gpurun_out/$TAG/bench_$n.json
    python -c "import json;d=json.load(open('gpurun_out/$TAG/bench_$n.json'));print('$n',d['ms_per_step'],d['roofline']['frac'],d['roofline'].get('avg_kernel_ms'),d['roofline'].get('avg_fold_ms'))"
  done; done
  # K3's exact form (SCFQ_HIST_EXACT; what a range whose guess does not verify is counted again with), the whole workload through it
  for w in "" "--workload nanopore"; do
    n=hist_exact$(echo $w | sed 's/--workload /_/')
    SCFQ_HIST_MODE=exact python bench.py --steps 6 --warmup 2 --no-cpu-baseline --ingest-bytes 0 --flags 1 $w 2>/dev/null | tail -1 > gpurun_out/$TAG/bench_$n.json
    python -c "import json;d=json.load(open('gpurun_out/$TAG/bench_$n.json'));print('$n',d['ms_per_step'],d['roofline']['frac'],d['roofline'].get('avg_kernel_ms'))"
  done ;;
prof)
  bash scripts/gpu_profile.sh $TAG > gpurun_out/$TAG/profile_summary.txt 2>&1
  grep -E "fq_scan_tiles|FETCH|WRITE" gpurun_out/$TAG/profile_summary.txt | head -20 ;;
profhist)
  PROF_STEPS=8 bash scripts/gpu_profile.sh ${TAG}_hist --flags 1 > gpurun_out/$TAG/profile_hist_summary.txt 2>&1
  PROF_STEPS=8 bash scripts/gpu_profile.sh ${TAG}_hist_struct --flags 3 > gpurun_out/$TAG/profile_hist_struct_summary.txt 2>&1
  PROF_STEPS=8 bash scripts/gpu_profile.sh ${TAG}_hist_nano --flags 1 --workload nanopore > gpurun_out/$TAG/profile_hist_nano_summary.txt 2>&1
  grep -E "fq_scan_tiles<(false|true), 2" gpurun_out/$TAG/profile_hist_summary.txt gpurun_out/$TAG/profile_hist_struct_summary.txt gpurun_out/$TAG/profile_hist_nano_summary.txt | head -60 ;;
gz)      # the warm device-gzip call of the 10 GB member: seven calls, twice; the kernel + memory-copy timeline of one call
  REPS=7 VARIANTS="default|SCFQ_NOTHING=1;default_again|SCFQ_NOTHING=1" bash scripts/gpu_r5_gz_probe.sh $TAG/gz 10e9 trace,variants ;;
gzkernels)   # decode / search / BGZF kernels with bytes per dispatch (the 10 GB member as ONE dispatch; a 6 GB BGZF file)
  bash scripts/gpu_r5_gz_probe.sh $TAG/gz_one 10e9 one
  bash scripts/gpu_profile_inflate.sh $TAG/inflate_2g 2e9
  bash scripts/gpu_profile_inflate.sh $TAG/inflate_6g 6e9 ;;
sharded)
  python3 scripts/measure_gz_sharded.py 10e9 gpurun_out/$TAG/gz_sharded.jsonl 2> gpurun_out/$TAG/gz_sharded.err; cut -c1-300 gpurun_out/$TAG/gz_sharded.jsonl ;;
cold)
  python3 scripts/measure_cold_ab.py 10e9 5 "default|SCFQ_NOTHING=1" > gpurun_out/$TAG/cold_10g.jsonl 2> gpurun_out/$TAG/cold.err
  python3 scripts/measure_cold_ab.py 2e9 5 "default|SCFQ_NOTHING=1" x bgzf > gpurun_out/$TAG/cold_bgzf_2g.jsonl 2>> gpurun_out/$TAG/cold.err
  cut -c1-260 gpurun_out/$TAG/cold_10g.jsonl gpurun_out/$TAG/cold_bgzf_2g.jsonl ;;
dedup)
  python scripts/bench_dedup.py 2>gpurun_out/$TAG/dedup.err | tee gpurun_out/$TAG/bench_dedup.json | cut -c1-600 ;;
tests)
  python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/$TAG/pytest_gpu.txt 2>&1; tail -25 gpurun_out/$TAG/pytest_gpu.txt ;;
esac
