#!/bin/bash
# Round 5, what feeds and fills the device during a warm device-gzip call of the 10 GB member (BASELINE configs[3]):
#   1. rocprofv3 --kernel-trace --memory-copy-trace of three calls: every H2D piece (engine / agent, bytes, duration, GB/s) next to the
#      decode kernels of the last call -> copies.txt, timeline.txt
#   2. the same call with the whole file as ONE batch (SCFQ_GZ_DEVICE_BATCH_SEGMENTS=40000): one decode dispatch of ~37000 segments,
#      longest first, on 5120 wave slots — what the decode reaches when the device never runs out of waves -> one_batch.txt
#   3. walls of schedule variants, unprofiled (SCFQ_VERBOSE phases): variants.jsonl
# usage: [VARIANTS="name|ENV=..;name2|ENV=.. ENV2=.."] scripts/gpu_r5_gz_probe.sh <tag> [inflated bytes] [steps: trace,one,variants]
TAG=${1:-r05_gz_probe}; N=${2:-10e9}; STEPS=${3:-trace,one,variants}
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
GZ=/tmp/r5_gz.fq.gz
python3 scripts/write_pigz_member.py $N $GZ ${WRITE_FLAGS:-} > $OUT/file.txt 2>&1 || { cat $OUT/file.txt; exit 1; }
cat $OUT/file.txt
cat > /tmp/r5_count.py <<'PY'
import sys, time
sys.path.insert(0, sys.argv[1] + "/seq-collection_amd/pyhost")
import scfq
for _ in range(int(sys.argv[3]) if len(sys.argv) > 3 else 3):
    t = time.time(); c = scfq.count_file(sys.argv[2]); print(c.reads, c.input_bytes, round((time.time() - t) * 1e3, 1), "ms", flush=True)
    time.sleep(0.1)      # (the trace summary finds a call by the pause before it)
PY
trace_one() {      # $1 = sub-directory, further arguments: environment assignments
  local d=$1; shift
  (cd /tmp && export TMPDIR=/tmp && export "$@" && timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/$d -o t -- python3 /tmp/r5_count.py $R $GZ ${TRACE_REPS:-4} > $OUT/$d.out 2> $OUT/$d.err)
  cat $OUT/$d.out
  python3 $R/scripts/gz_trace_summary.py $OUT/$d > $OUT/$d.txt
  # (the raw traces are tens of MB: the summaries are what is kept)
  find $OUT/$d -name '*.csv' -size +2M -delete
}
case ",$STEPS," in *,trace,*)
  trace_one trace_default ${TRACE_ENV:-SCFQ_NOTHING=1}
  head -c 5000 $OUT/trace_default.txt;;
esac
case ",$STEPS," in *,one,*)
  trace_one trace_one_batch SCFQ_GZ_DEVICE_BATCH_SEGMENTS=40000
  head -c 3000 $OUT/trace_one_batch.txt;;
esac
case ",$STEPS," in *,variants,*)
  IFS=';' read -ra VLIST <<< "${VARIANTS:-default|SCFQ_NOTHING=1;slots3|SCFQ_GZ_DEVICE_SLOTS=3;one_batch|SCFQ_GZ_DEVICE_BATCH_SEGMENTS=40000;default_again|SCFQ_NOTHING=1}"
  for v in "${VLIST[@]}"; do
    name=${v%%|*}; envs=${v#*|}
    ( export $envs SCFQ_VERBOSE=1; timeout -k 10 200 python3 /tmp/r5_count.py $R $GZ ${REPS:-4} > $OUT/var_$name.out 2> $OUT/var_$name.err )
    echo "== $name ($envs): $(tr '\n' ' ' < $OUT/var_$name.out)" | tee -a $OUT/variants.txt
    grep -E "copy to HBM|segment decode|window chain|wall|copier thread|batch\(es\)|high water" $OUT/var_$name.err | tail -8 >> $OUT/variants.txt
  done;;
esac
rm -f $GZ
