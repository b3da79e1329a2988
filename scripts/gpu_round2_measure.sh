#!/bin/bash
# Round-2 measurement pass on the GPU box (run through gpurun in pieces: see the calls in the round's log).
# usage: scripts/gpu_round2_measure.sh <part>     part = tests | bench | aux | prof | inflate
TAG=r02; mkdir -p gpurun_out/$TAG
case "$1" in
tests)
  python -m pytest tests -q -m gpu 2>&1 | tail -6 | tee gpurun_out/$TAG/pytest_gpu.txt ;;
bench)
  python bench.py --steps 30 --warmup 3 2>gpurun_out/$TAG/bench.err | tail -1 | tee gpurun_out/$TAG/bench_n1.json | cut -c1-300
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --flags 2 2>/dev/null | tail -1 > gpurun_out/$TAG/bench_struct.json
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --flags 1 2>/dev/null | tail -1 > gpurun_out/$TAG/bench_hist.json
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --flags 3 2>/dev/null | tail -1 > gpurun_out/$TAG/bench_hist_struct.json
  SCFQ_HIST_MODE=exact python bench.py --steps 5 --warmup 1 --no-cpu-baseline --flags 1 2>/dev/null | tail -1 > gpurun_out/$TAG/bench_hist_exact.json
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --flags 1 --workload nanopore 2>/dev/null | tail -1 > gpurun_out/$TAG/bench_hist_nanopore.json
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload nanopore 2>/dev/null | tail -1 > gpurun_out/$TAG/bench_nanopore.json
  python bench.py --gpus 1 --exchange-at-1 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/$TAG/bench_exchange_at_1.json
  python bench.py --gpus 1 --exchange-at-1 --bytes-per-gpu 25e9 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/$TAG/bench_exchange_at_1_25GB.json
  python scripts/bench_dedup.py 10e9 0.2 5 2>/dev/null | tail -1 > gpurun_out/$TAG/bench_dedup.json
  for f in n1 struct hist hist_struct hist_exact hist_nanopore nanopore exchange_at_1 exchange_at_1_25GB; do python -c "import json;d=json.load(open('gpurun_out/$TAG/bench_$f.json'));print('$f',d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline']['avg_kernel_ms'],d['config'].get('exchange','')[:40])"; done
  cut -c1-250 gpurun_out/$TAG/bench_dedup.json ;;
aux)
  python scripts/measure_ingest.py 2e9 /tmp > gpurun_out/$TAG/ingest.jsonl 2>gpurun_out/$TAG/ingest.err
  python scripts/measure_bgzf_device.py 4e9 > gpurun_out/$TAG/bgzf_device.jsonl 2>gpurun_out/$TAG/bgzf.err
  python scripts/measure_gz_device.py 1e9 /tmp > gpurun_out/$TAG/gz_device.jsonl 2>gpurun_out/$TAG/gzd.err
  python scripts/measure_gz_device.py 2e9 /tmp >> gpurun_out/$TAG/gz_device.jsonl 2>>gpurun_out/$TAG/gzd.err
  python scripts/measure_gz_device.py 10e9 /tmp >> gpurun_out/$TAG/gz_device.jsonl 2>>gpurun_out/$TAG/gzd.err
  cat gpurun_out/$TAG/bgzf_device.jsonl; cut -c1-600 gpurun_out/$TAG/gz_device.jsonl ;;
prof)
  bash scripts/gpu_profile.sh $TAG > gpurun_out/$TAG/profile_summary.txt 2>&1
  PROF_STEPS=8 bash scripts/gpu_profile.sh ${TAG}_hist --flags 1 > gpurun_out/$TAG/profile_summary_hist.txt 2>&1
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof/${TAG}_dedup -o dd -- python3 $GRAFT_REPO_ROOT/scripts/bench_dedup.py 10e9 0.2 3 > /dev/null 2>&1)
  python3 - <<PY > gpurun_out/$TAG/dedup_kernel_stats.txt
import csv, glob
for f in glob.glob("gpurun_out/prof/${TAG}_dedup/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:24]:
        print(r["Name"][:90].ljust(90), r["Calls"].rjust(4), "%10.3f ms avg" % (float(r["AverageNs"]) / 1e6))
PY
  grep -E "fq_scan_tiles|FETCH|WRITE" gpurun_out/$TAG/profile_summary.txt | head -20; grep -E "fq_scan_tiles" gpurun_out/$TAG/profile_summary_hist.txt | head; head -12 gpurun_out/$TAG/dedup_kernel_stats.txt ;;
inflate)
  bash scripts/gpu_profile_inflate.sh $TAG 2e9 ;;
esac
