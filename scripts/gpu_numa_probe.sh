#!/bin/bash
R=$(pwd); OUT=$R/gpurun_out/r05_numa; mkdir -p $OUT
python3 scripts/write_pigz_member.py 10e9 /tmp/n.fq.gz > $OUT/file.txt
cat > /tmp/r5_count.py <<'PY'
import sys, time
sys.path.insert(0, sys.argv[1] + "/seq-collection_amd/pyhost")
import scfq
for _ in range(int(sys.argv[3])):
    t = time.time(); c = scfq.count_file(sys.argv[2]); print(round((time.time() - t) * 1e3, 1), end=" ", flush=True)
print()
PY
for rep in 1 2; do
for cpus in "all|0-255" "node0|0-63,128-191" "node1|64-127,192-255" "node0_32|0-31"; do
  name=${cpus%%|*}; set=${cpus#*|}
  echo -n "$name ($set): " | tee -a $OUT/numa.txt
  SCFQ_VERBOSE=1 taskset -c $set python3 /tmp/r5_count.py $R /tmp/n.fq.gz 7 2> $OUT/$name.err | tee -a $OUT/numa.txt
  grep "copier thread" $OUT/$name.err | tail -1 | cut -c1-120 | tee -a $OUT/numa.txt
done; done
rm -f /tmp/n.fq.gz
