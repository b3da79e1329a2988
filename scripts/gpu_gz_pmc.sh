#!/bin/bash
# PMC counters of the device gzip kernels over a warm multi-batch call (rocprofv3 --pmc, counters only): what the decode waves wait for.
# usage: scripts/gpu_gz_pmc.sh <tag> [inflated bytes]        (expects /tmp/tl_gz.fq.gz from scripts/gpu_gz_timeline.sh, or writes it)
TAG=${1:-r03}; N=${2:-6e9}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
if [ ! -f /tmp/tl_gz.fq.gz ]; then
python3 - <<PY
import os, sys, zlib
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, "seq-collection_amd/pyhost")
import scfq
plan = scfq.synth_plan(0, 20260101, int(float("$N")))
data, info = scfq.synth_host(0, 20260101, plan.records)
step = 64 << 20
cuts = list(range(0, data.size, step))
def piece(i):
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    return co.compress(data[cuts[i]:cuts[i] + step].tobytes()) + co.flush(zlib.Z_FINISH if i == len(cuts) - 1 else zlib.Z_SYNC_FLUSH)
with ThreadPoolExecutor(16) as ex:
    members = list(ex.map(piece, range(len(cuts))))
crc = 0
for c0 in cuts: crc = zlib.crc32(data[c0:c0 + step], crc)
with open("/tmp/tl_gz.fq.gz", "wb") as f:
    f.write(b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03"); [f.write(b) for b in members]
    f.write(int(crc & 0xFFFFFFFF).to_bytes(4, "little") + int(data.size & 0xFFFFFFFF).to_bytes(4, "little"))
PY
fi
cat > /tmp/tl_count.py <<'PY'
import sys, time
sys.path.insert(0, sys.argv[1] + "/seq-collection_amd/pyhost")
import scfq
for _ in range(2):
    t = time.time(); c = scfq.count_file(sys.argv[2]); print(c.reads, c.input_bytes, round((time.time() - t) * 1e3, 1), "ms", flush=True)
PY
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-30)
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc $grp --output-format csv -d $OUT/gzpmc_$name -o pmc -- python3 /tmp/tl_count.py $R /tmp/tl_gz.fq.gz > $OUT/gzpmc_$name.out 2> $OUT/gzpmc_$name.err)
done
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); nd = collections.defaultdict(set)
for f in glob.glob("$OUT/gzpmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-36:]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); nd[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
with open("$OUT/gz_pmc.txt", "w") as out:
    out.write("# rocprofv3 --pmc over two counts of a 6 GB pigz-style member (device gzip path): per kernel, counter sums over all dispatches\n")
    for k in sorted(tot):
        if not any(s in k for s in ("gz_", "fq_scan")): continue
        out.write(k + "\n")
        for cname in sorted(tot[k]):
            out.write("   %-24s %18.0f   (%d dispatches)\n" % (cname, tot[k][cname], len(nd[(k, cname)])))
print(open("$OUT/gz_pmc.txt").read()[:5000])
PY
