#!/bin/bash
# Kernel durations of the device-side BGZF inflate (rocprofv3 kernel trace of three counts of one file) -> gpurun_out/dinflate_prof/
# usage: gpu_dinflate_profile.sh [bytes]     (run on the GPU box)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
N=${1:-2e9}
OUT=$R/gpurun_out/dinflate_prof; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/scripts/measure_bgzf_device.py $N > $OUT/measure.jsonl 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o t -- python3 $R/scripts/count_file_loop.py /tmp/scfq_bgzf_dev.fq.gz 3 > $OUT/run.txt 2>&1
F=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
cp $F $OUT/kernel_stats.csv
head -4 $OUT/kernel_stats.csv | cut -c1-200
