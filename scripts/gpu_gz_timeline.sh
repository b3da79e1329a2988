#!/bin/bash
# Kernel timeline of ONE warm device-gzip call (rocprofv3 --kernel-trace): every dispatch of the last of three counts of a pigz-style member,
# in start order, with start / duration / stream — where the device is idle or crowded.   usage: scripts/gpu_gz_timeline.sh <tag> [inflated bytes]
TAG=${1:-r03}; N=${2:-6e9}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
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
cat > /tmp/tl_count.py <<'PY'
import sys, time
sys.path.insert(0, sys.argv[1] + "/seq-collection_amd/pyhost")
import scfq
for _ in range(3):
    t = time.time(); c = scfq.count_file(sys.argv[2]); print(c.reads, c.input_bytes, round((time.time() - t) * 1e3, 1), "ms", flush=True)
PY
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT/tl_gz -o t -- python3 /tmp/tl_count.py $R /tmp/tl_gz.fq.gz > $OUT/tl_gz.out 2> $OUT/tl_gz.err)
cat $OUT/tl_gz.out
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$OUT/tl_gz/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:], r.get("Queue_Id", ""), r.get("Stream_Id", "")))
rows.sort()
# the last call: from the last gz_sync_search that follows a pause of > 20 ms
starts = [i for i in range(1, len(rows)) if rows[i][0] - max(r[1] for r in rows[:i]) > 20_000_000]
lo = starts[-1] if starts else 0
t0 = rows[lo][0]
with open("$OUT/gz_timeline.txt", "w") as out:
    out.write("# one warm device-gzip call (the last of three), every dispatch: start ms, duration ms, kernel, queue\n")
    for s, e, n, q, st in rows[lo:]:
        if (e - s) > 200_000 or "decode" in n or "search" in n:
            out.write("%9.3f %9.3f  %-40s q%s\n" % ((s - t0) / 1e6, (e - s) / 1e6, n, q))
print(open("$OUT/gz_timeline.txt").read()[:6000])
PY
