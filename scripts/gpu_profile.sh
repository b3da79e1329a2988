#!/bin/bash
# rocprofv3 passes over a short bench run (GPU box, via gpurun). Summaries land in gpurun_out/prof/<tag>/
# usage: scripts/gpu_profile.sh <tag> [bench args...]
TAG=${1:-r01}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BARGS="--steps ${PROF_STEPS:-20} --warmup 3 --no-cpu-baseline --ingest-bytes 0 $@"
# 1) kernel trace + stats
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $R/bench.py $BARGS > $OUT/trace_bench.json 2> $OUT/trace.err
# 2) PMC passes (counters only; each group in its own run)
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$name -o pmc -- python3 $R/bench.py $BARGS > /dev/null 2> $OUT/pmc_$name.err
done
cd $R
python3 scripts/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt | head -60
