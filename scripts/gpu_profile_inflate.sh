#!/bin/bash
# Kernel trace of the device-side inflate kernels with BYTES PER DISPATCH next to every duration, so that the GB/s quoted in
# DESIGN.md can be recomputed from the committed file:
#   bgzf_inflate      : the library appends one line per dispatch (members, compressed bytes, inflated bytes) to SCFQ_BGZF_LAUNCH_LOG
#   gz_segment_decode : one dispatch per file here; its bytes are the file's inflated size
# usage: scripts/gpu_profile_inflate.sh <tag> [inflated bytes, default 2e9]     (GPU box; results in gpurun_out/<tag>/)
TAG=${1:-r02}; N=${2:-2e9}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
python3 - <<PY
import os, struct, sys, zlib
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, "seq-collection_amd/pyhost")
import scfq
plan = scfq.synth_plan(0, 20260101, int(float("$N")))
data, info = scfq.synth_host(0, 20260101, plan.records)
def bgzf_block(b):
    co = zlib.compressobj(6, zlib.DEFLATED, -15); payload = co.compress(b) + co.flush(); bs = 18 + len(payload) + 8
    return b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bs - 1) + payload + struct.pack("<II", zlib.crc32(b) & 0xFFFFFFFF, len(b))
def span(a):
    raw = a.tobytes(); return b"".join(bgzf_block(raw[i:i + 0xff00]) for i in range(0, len(raw), 0xff00))
parts = [data[i:i + (0xff00 * 256)] for i in range(0, data.size, 0xff00 * 256)]
with ThreadPoolExecutor(16) as ex:
    blobs = list(ex.map(span, parts))
with open("/tmp/prof_bgzf.fq.gz", "wb") as f:
    for b in blobs: f.write(b)
step = 64 << 20
cuts = list(range(0, data.size, step))
def piece(i):
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    return co.compress(data[cuts[i]:cuts[i] + step].tobytes()) + co.flush(zlib.Z_FINISH if i == len(cuts) - 1 else zlib.Z_SYNC_FLUSH)
with ThreadPoolExecutor(16) as ex:
    members = list(ex.map(piece, range(len(cuts))))
crc = 0
for c0 in cuts: crc = zlib.crc32(data[c0:c0 + step], crc)
with open("/tmp/prof_gz.fq.gz", "wb") as f:
    f.write(b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03"); [f.write(b) for b in members]
    f.write(int(crc & 0xFFFFFFFF).to_bytes(4, "little") + int(data.size & 0xFFFFFFFF).to_bytes(4, "little"))
open("/tmp/prof_inflated_bytes", "w").write(str(data.size))
PY
cat > /tmp/prof_count.py <<'PY'
import sys
sys.path.insert(0, sys.argv[1] + "/seq-collection_amd/pyhost")
import scfq
for _ in range(3):
    c = scfq.count_file(sys.argv[2])
print(c.reads, c.input_bytes)
PY
rm -f /tmp/bgzf_launch.log
export SCFQ_BGZF_LAUNCH_LOG=/tmp/bgzf_launch.log
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_bgzf -o t -- python3 /tmp/prof_count.py $R /tmp/prof_bgzf.fq.gz > $OUT/prof_bgzf.out 2> $OUT/prof_bgzf.err)
unset SCFQ_BGZF_LAUNCH_LOG
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_gz -o t -- python3 /tmp/prof_count.py $R /tmp/prof_gz.fq.gz > $OUT/prof_gz.out 2> $OUT/prof_gz.err)
python3 - <<PY
import csv, glob
out = "$OUT"
inflated = int(open("/tmp/prof_inflated_bytes").read())
launch = [l.split() for l in open("/tmp/bgzf_launch.log")]
rows = []
for f in glob.glob(out + "/prof_bgzf/**/*kernel_trace.csv", recursive=True):
    rows += [r for r in csv.DictReader(open(f)) if "bgzf_inflate" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# (the library's first call launches an empty one-workgroup bgzf_inflate on a helper thread to have the kernel's code and scratch set up:
# it is not in the launch log)
while len(rows) > len(launch) and int(rows[0]["Grid_Size_X"]) == int(rows[0]["Workgroup_Size_X"]): rows.pop(0)
assert len(rows) == len(launch), (len(rows), len(launch))
with open(out + "/bgzf_dispatches.tsv", "w") as w:
    w.write("# bgzf_inflate dispatches of three counts of a %d-byte (inflated) BGZF level-6 file: rocprofv3 --kernel-trace durations joined, in dispatch order, with the library's launch log\n" % inflated)
    w.write("dispatch\tmembers\tcompressed_bytes\tinflated_bytes\tduration_us\tinflated_GBps\n")
    tot_b = tot_t = 0
    for k, (r, l) in enumerate(zip(rows, launch)):
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        w.write("%d\t%s\t%s\t%s\t%.1f\t%.1f\n" % (k, l[0], l[1], l[2], dur, int(l[2]) / dur / 1e3))
        tot_b += int(l[2]); tot_t += dur
    w.write("# %d dispatches (%d launch-log lines), %.0f bytes in %.1f us of kernel time: %.1f GB/s\n" % (len(rows), len(launch), tot_b, tot_t, tot_b / max(tot_t, 1e-9) / 1e3))
agg = {}
for f in glob.glob(out + "/prof_gz/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gz_segment_decode" in r["Kernel_Name"] and int(r["Grid_Size_X"]) == int(r["Workgroup_Size_X"]) and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) < 200000: continue   # the empty warm-up launch of a process's first call
        n = r["Kernel_Name"].split("(")[0][:60]
        a = agg.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
with open(out + "/gz_device_kernels.tsv", "w") as w:
    w.write("# kernels of three counts of a %d-byte (inflated) single-member gzip (zlib level 6, pigz-style), rocprofv3 --kernel-trace\n" % inflated)
    w.write("kernel\tdispatches\ttotal_us\tavg_us\tinflated_GBps_of_the_file_per_dispatch\n")
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        w.write("%s\t%d\t%.1f\t%.1f\t%s\n" % (n, c, t, t / c, ("%.1f" % (inflated / (t / c) / 1e3)) if n.startswith("scfq_dinflate::gz_") or "gz_" in n else ""))
print(open(out + "/bgzf_dispatches.tsv").read()[-400:])
print(open(out + "/gz_device_kernels.tsv").read()[:1500])
PY
