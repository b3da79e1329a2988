#!/bin/bash
# fq-dedup A/B on the GPU box: the pipeline's knobs against each other, then its host-clock stage times and a kernel timeline.
# usage: scripts/gpu_dedup_ab.sh <out-subdir>
TAG=${1:-dedup}
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd $ROOT
OUT=gpurun_out/$TAG; mkdir -p $OUT
one() { # name, env...
  n=$1; shift
  env SCFQ_NOOP=1 "$@" timeout -k 10 200 python scripts/bench_dedup.py 10e9 0.2 6 2>$OUT/$n.err | tail -1 > $OUT/$n.json || { echo "$n FAILED"; tail -3 $OUT/$n.err; return 1; }
  python -c "import json;d=json.load(open('$OUT/$n.json'));print('%-28s %.2f ms  %.1f GB/s  collisions %d' % ('$n',d['ms'],d['value'],d['hash_collisions']))" | tee -a $OUT/summary.txt
  grep "rep times" $OUT/$n.err | tee -a $OUT/summary.txt
}
OLD=$PWD/seq-collection_amd/ablate/libsc_fqcount_hip_ddold.so     # the library of the commit before (built by hand from git show HEAD:...)
for round in 1 2; do
  if [ -f $OLD ]; then one old_$round SCFQ_LIB_OVERRIDE=$OLD || exit 1; fi
  one new_$round || exit 1
done
SCFQ_DEDUP_TRACE=2 timeout -k 10 200 python scripts/bench_dedup.py 10e9 0.2 2 2>$OUT/trace2.err >/dev/null; echo "host enqueue times (no synchronisation between the marks):" | tee -a $OUT/summary.txt; grep "scfq dedup" $OUT/trace2.err | sed -n 22,33p | tee -a $OUT/summary.txt
SCFQ_DEDUP_TRACE=1 timeout -k 10 200 python scripts/bench_dedup.py 10e9 0.2 2 2>$OUT/trace.err >/dev/null; grep "scfq dedup" $OUT/trace.err | tail -14 | tee -a $OUT/summary.txt
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/prof/dedup_$TAG -o dd -- python3 $ROOT/scripts/bench_dedup.py 10e9 0.2 2 > /dev/null 2>&1)
python3 - <<PY | tee $OUT/timeline.txt
import csv, glob
rows=[]
for f in glob.glob("gpurun_out/prof/dedup_$TAG/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# the last 10 GB call: find the last fq_index_masks with a long duration
big=[i for i,r in enumerate(rows) if "fq_index" in r["Kernel_Name"] and int(r["End_Timestamp"])-int(r["Start_Timestamp"])>1000000]
if big:
    i0=big[-1]
    while i0>0 and int(rows[i0]["Start_Timestamp"])-int(rows[i0-1]["End_Timestamp"])<50000: i0-=1
    t0=int(rows[i0]["Start_Timestamp"]); prev=t0; ksum=0
    for r in rows[i0:]:
        s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
        if s-prev>5000000: break
        print("%-70s start %9.1f us  dur %9.1f us  gap %8.1f us" % (r["Kernel_Name"][:70],(s-t0)/1e3,(e-s)/1e3,(s-prev)/1e3))
        prev=e; ksum+=e-s
    print("# span %.1f us, kernel time %.1f us" % ((prev-t0)/1e3, ksum/1e3))
PY
