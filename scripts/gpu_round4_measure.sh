#!/bin/bash
# Round-4 measurement pass on the GPU box (run through gpurun in pieces).
# usage: scripts/gpu_round4_measure.sh <part>     part = bench | gz | cold | prof | aux | hist | dedup
TAG=r04; mkdir -p gpurun_out/$TAG
case "$1" in
bench)
  python3 bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/$TAG/bench.err | tail -1 | tee gpurun_out/$TAG/bench_n1.json | cut -c1-400 ;;
anchor)
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --bytes-per-gpu 25e9 --no-cpu-baseline --ingest-bytes 0 2>gpurun_out/$TAG/bench_anchor.err | tail -1 | tee gpurun_out/$TAG/bench_n1_anchor_25GB.json | cut -c1-600 ;;
gz)
  python scripts/measure_gz_device.py 2e9 /tmp > gpurun_out/$TAG/gz_device.jsonl 2>gpurun_out/$TAG/gzd.err
  SCFQ_MEASURE_LOG=gpurun_out/$TAG/gz_device_10g.log python scripts/measure_gz_device.py 10e9 /tmp >> gpurun_out/$TAG/gz_device.jsonl 2>>gpurun_out/$TAG/gzd.err
  cut -c1-900 gpurun_out/$TAG/gz_device.jsonl ;;
cold)
  python scripts/measure_cold_stages.py /tmp 5 0.5e9,2e9 > gpurun_out/$TAG/cold_stages.jsonl 2> gpurun_out/$TAG/cold_stages.err
  python scripts/measure_cold_stages.py /tmp 3 10e9 > gpurun_out/$TAG/cold_stages_10g.jsonl 2>> gpurun_out/$TAG/cold_stages.err
  python -c "
import json
for f in ('gpurun_out/$TAG/cold_stages.jsonl','gpurun_out/$TAG/cold_stages_10g.jsonl'):
    for l in open(f):
        j=json.loads(l); print(j['file'], j['runs_ms_in_order'], 'median', j['median_ms'])" ;;
prof)
  bash scripts/gpu_profile.sh $TAG > gpurun_out/$TAG/profile_summary.txt 2>&1
  grep -E "fq_scan_tiles|FETCH|WRITE" gpurun_out/$TAG/profile_summary.txt | head -20 ;;
profhist)
  PROF_STEPS=8 bash scripts/gpu_profile.sh ${TAG}_hist --flags 1 > gpurun_out/$TAG/profile_hist_summary.txt 2>&1
  PROF_STEPS=8 bash scripts/gpu_profile.sh ${TAG}_hist_nano --flags 1 --workload nanopore > gpurun_out/$TAG/profile_hist_nano_summary.txt 2>&1
  grep -E "fq_scan_tiles<false, 2" gpurun_out/$TAG/profile_hist_summary.txt gpurun_out/$TAG/profile_hist_nano_summary.txt | head -60 ;;
hist)
  for w in "" "--workload nanopore"; do for f in 1 3 2; do
    n=hist_f${f}$(echo $w | sed 's/--workload /_/')
    python bench.py --steps 10 --warmup 2 --no-cpu-baseline --ingest-bytes 0 --flags $f $w 2>/dev/null | tail -1 > gpurun_out/$TAG/bench_$n.json
    python -c "import json;d=json.load(open('gpurun_out/$TAG/bench_$n.json'));print('$n',d['ms_per_step'],d['roofline']['frac'],d['roofline'].get('avg_kernel_ms'))"
  done; done ;;
jobs)
  python scripts/measure_jobs.py /tmp > gpurun_out/$TAG/jobs_8_files.jsonl 2> gpurun_out/$TAG/jobs.err; cut -c1-330 gpurun_out/$TAG/jobs_8_files.jsonl ;;
aux)
  python scripts/measure_ingest.py 2e9 /tmp > gpurun_out/$TAG/ingest.jsonl 2>gpurun_out/$TAG/ingest.err
  python scripts/measure_bgzf_device.py 4e9 > gpurun_out/$TAG/bgzf_device.jsonl 2>gpurun_out/$TAG/bgzf.err
  cut -c1-400 gpurun_out/$TAG/ingest.jsonl; cat gpurun_out/$TAG/bgzf_device.jsonl ;;
dedup)
  python scripts/bench_dedup.py 2>gpurun_out/$TAG/dedup.err | tee gpurun_out/$TAG/bench_dedup.json | cut -c1-600 ;;
esac
