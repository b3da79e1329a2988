#!/bin/bash
# Round-3 measurement pass on the GPU box (run through gpurun in pieces).
# usage: scripts/gpu_round3_measure.sh <part>     part = bench | gz | cold | prof | aux
TAG=r03; mkdir -p gpurun_out/$TAG
case "$1" in
bench)
  python bench.py --steps 30 --warmup 3 2>gpurun_out/$TAG/bench.err | tail -1 | tee gpurun_out/$TAG/bench_n1.json | cut -c1-400
  python -c "import json;d=json.load(open('gpurun_out/$TAG/bench_n1.json'));print(json.dumps(d.get('ingest'),indent=1))" ;;
gz)
  python scripts/measure_gz_device.py 1e9 /tmp > gpurun_out/$TAG/gz_device.jsonl 2>gpurun_out/$TAG/gzd.err
  python scripts/measure_gz_device.py 2e9 /tmp >> gpurun_out/$TAG/gz_device.jsonl 2>>gpurun_out/$TAG/gzd.err
  SCFQ_MEASURE_LOG=gpurun_out/$TAG/gz_device_10g.log python scripts/measure_gz_device.py 10e9 /tmp >> gpurun_out/$TAG/gz_device.jsonl 2>>gpurun_out/$TAG/gzd.err
  cut -c1-700 gpurun_out/$TAG/gz_device.jsonl ;;
cold)
  rm -f gpurun_out/$TAG/gz_cold.log
  SCFQ_MEASURE_LOG=gpurun_out/$TAG/gz_cold.log python scripts/measure_gz_cold.py 10e9 /tmp pigz 4 > gpurun_out/$TAG/gz_cold.jsonl 2> gpurun_out/$TAG/gz_cold.err
  SCFQ_MEASURE_LOG=gpurun_out/$TAG/gz_cold.log python scripts/measure_gz_cold.py 2e9 /tmp bgzf 3 >> gpurun_out/$TAG/gz_cold.jsonl 2>> gpurun_out/$TAG/gz_cold.err
  SCFQ_MEASURE_LOG=gpurun_out/$TAG/gz_cold.log python scripts/measure_gz_cold.py 0.5e9 /tmp pigz 3 >> gpurun_out/$TAG/gz_cold.jsonl 2>> gpurun_out/$TAG/gz_cold.err
  cat gpurun_out/$TAG/gz_cold.err ;;
gzip6)
  python scripts/measure_gz_cold.py 2.2e9 /tmp gzip 3 > gpurun_out/$TAG/gz_plain_gzip6.jsonl 2> gpurun_out/$TAG/gz_plain_gzip6.err; cat gpurun_out/$TAG/gz_plain_gzip6.err ;;
prof)
  bash scripts/gpu_profile.sh $TAG > gpurun_out/$TAG/profile_summary.txt 2>&1
  grep -E "fq_scan_tiles|FETCH|WRITE" gpurun_out/$TAG/profile_summary.txt | head -20 ;;
jobs)
  python scripts/measure_jobs.py /tmp > gpurun_out/$TAG/jobs_8_files.jsonl 2> gpurun_out/$TAG/jobs.err; cut -c1-330 gpurun_out/$TAG/jobs_8_files.jsonl ;;
aux)
  python scripts/measure_ingest.py 2e9 /tmp > gpurun_out/$TAG/ingest.jsonl 2>gpurun_out/$TAG/ingest.err
  python scripts/measure_bgzf_device.py 4e9 > gpurun_out/$TAG/bgzf_device.jsonl 2>gpurun_out/$TAG/bgzf.err
  cut -c1-400 gpurun_out/$TAG/ingest.jsonl; cat gpurun_out/$TAG/bgzf_device.jsonl ;;
esac
