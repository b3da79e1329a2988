#!/bin/bash
# Full measurement pass on the GPU box: all -m gpu tests, bench (with cpu baseline), variants, ingest paths, rocprofv3.
TAG=${1:-r01}
mkdir -p gpurun_out/$TAG
python -m pytest tests -q -m gpu 2>&1 | tail -5 | tee gpurun_out/$TAG/pytest_gpu.txt
python bench.py --steps 30 --warmup 3 2>gpurun_out/$TAG/bench.err | tail -1 | tee gpurun_out/$TAG/bench_n1.json | cut -c1-400
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --flags 2 2>/dev/null | tail -1 > gpurun_out/$TAG/bench_struct.json
python bench.py --steps 5 --warmup 1 --no-cpu-baseline --flags 1 2>/dev/null | tail -1 > gpurun_out/$TAG/bench_hist.json
SCFQ_HIST_MODE=exact python bench.py --steps 5 --warmup 1 --no-cpu-baseline --flags 1 2>/dev/null | tail -1 > gpurun_out/$TAG/bench_hist_exact.json
python bench.py --steps 5 --warmup 1 --no-cpu-baseline --flags 1 --workload nanopore 2>/dev/null | tail -1 > gpurun_out/$TAG/bench_hist_nanopore.json
python scripts/bench_dedup.py 10e9 0.2 5 2>/dev/null | tail -1 > gpurun_out/$TAG/bench_dedup.json
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload nanopore 2>/dev/null | tail -1 > gpurun_out/$TAG/bench_nanopore.json
python scripts/measure_ingest.py 2e9 /tmp > gpurun_out/$TAG/ingest.jsonl 2>gpurun_out/$TAG/ingest.err
python scripts/measure_gz.py 2e9 /tmp > gpurun_out/$TAG/gz_inflate.jsonl 2>gpurun_out/$TAG/gz.err
python scripts/measure_bgzf_device.py 4e9 > gpurun_out/$TAG/bgzf_device.jsonl 2>gpurun_out/$TAG/bgzf.err
python scripts/measure_pgz.py 6e9 > gpurun_out/$TAG/gz_single_member_parallel.jsonl 2>gpurun_out/$TAG/pgz.err
cat gpurun_out/$TAG/bench_dedup.json
for f in struct hist hist_exact hist_nanopore nanopore; do python -c "import json;d=json.load(open('gpurun_out/$TAG/bench_$f.json'));print('$f',d['value'],d['roofline'])"; done
cat gpurun_out/$TAG/ingest.jsonl gpurun_out/$TAG/gz_inflate.jsonl gpurun_out/$TAG/bgzf_device.jsonl
bash scripts/gpu_profile.sh $TAG > gpurun_out/$TAG/profile_summary.txt 2>&1
PROF_STEPS=8 bash scripts/gpu_profile.sh ${TAG}_hist --flags 1 > gpurun_out/$TAG/profile_summary_hist.txt 2>&1
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof/${TAG}_dedup -o dd -- python3 $GRAFT_REPO_ROOT/scripts/bench_dedup.py 10e9 0.2 3 > /dev/null 2>&1)
python3 - <<PY > gpurun_out/$TAG/dedup_kernel_stats.txt
import csv
for r in list(csv.DictReader(open("gpurun_out/prof/${TAG}_dedup/dd_kernel_stats.csv")))[:24]:
    print(r["Name"][:90].ljust(90), r["Calls"].rjust(4), "%10.3f ms avg" % (float(r["AverageNs"]) / 1e6))
PY
grep -E "fq_scan_tiles" gpurun_out/$TAG/profile_summary.txt | head -40
