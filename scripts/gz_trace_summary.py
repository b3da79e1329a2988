#!/usr/bin/env python3
"""Summary of a `rocprofv3 --kernel-trace --memory-copy-trace` directory of device-gzip calls (scripts/gpu_r5_gz_probe.sh):
the LAST call's dispatches and memory copies on one time axis, the copies' rates alone and beside decode kernels, and which
runtime blit kernels (if any) ran — i.e. whether host-to-device pieces go through an SDMA engine or through CUs.
usage: gz_trace_summary.py <dir>"""
import csv, glob, sys, collections
d = sys.argv[1]
kern, cops = [], []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kern.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-44:], r.get("Queue_Id", "")))
copy_cols = None
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    rd = csv.DictReader(open(f))
    copy_cols = rd.fieldnames
    for r in rd:
        b = r.get("Bytes") or r.get("Size") or r.get("bytes") or "0"
        cops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", r.get("Kind", "?")), int(b), r.get("Source_Agent_Id", ""), r.get("Destination_Agent_Id", ""), r.get("Stream_Id", "")))
kern.sort(); cops.sort()
print("# memory-copy trace columns:", copy_cols)
if not kern:
    print("no kernel trace"); sys.exit(0)
# the last call: from the last gz_sync_search that follows a pause of > 20 ms in the kernel stream
starts = [i for i in range(1, len(kern)) if kern[i][0] - max(k[1] for k in kern[:i]) > 20_000_000]
lo = starts[-1] if starts else 0
t0 = min(kern[lo][0], min([c[0] for c in cops if c[0] > kern[lo][0] - 15_000_000] or [kern[lo][0]]))
t1 = max(k[1] for k in kern[lo:])
call_k = kern[lo:]
call_c = [c for c in cops if t0 <= c[0] <= t1]
print("# the last call: %.1f ms from its first copy / kernel to its last kernel; %d dispatches, %d memory copies" % ((t1 - t0) / 1e6, len(call_k), len(call_c)))
blit = collections.Counter()
for s, e, n, q in call_k:
    if "rocclr" in n or "blit" in n.lower() or "copyBuffer" in n:
        blit[n] += 1
print("# runtime blit kernels in the call (device-to-device one-byte parks etc.):", dict(blit) or "none")
dec = [(s, e) for s, e, n, q in call_k if "segment_decode" in n]
def overlap(a, b):
    return sum(max(0, min(b, e) - max(a, s)) for s, e in dec)
print("# H2D pieces >= 1 MiB: start ms, ms, MB, GB/s, share of the copy's time with a decode kernel running, agents src->dst")
big = [c for c in call_c if c[3] >= (1 << 20) and "HOST_TO_DEVICE" in c[2].upper().replace(" ", "_")]
if not big:
    big = [c for c in call_c if c[3] >= (1 << 20)]
tot_b = tot_t = 0
alone_b = alone_t = beside_b = beside_t = 0
for s, e, dr, b, sa, da, st in big:
    ov = overlap(s, e) / max(1, e - s)
    print("%9.3f %8.3f %8.1f %7.1f  %4.0f %%  %s->%s  %s" % ((s - t0) / 1e6, (e - s) / 1e6, b / 1e6, b / max(1, e - s), 100 * ov, sa, da, dr))
    tot_b += b; tot_t += e - s
    if ov > 0.8: beside_b += b; beside_t += e - s
    elif ov < 0.2: alone_b += b; alone_t += e - s
if big:
    span = max(c[1] for c in big) - min(c[0] for c in big)
    print("# H2D total %.1f MB in %.1f ms of copy time = %.1f GB/s while a copy runs; first start to last end %.1f ms = %.1f GB/s" % (tot_b / 1e6, tot_t / 1e6, tot_b / max(1, tot_t), span / 1e6, tot_b / max(1, span)))
    if alone_t: print("#   pieces with no decode kernel beside them (<20 %% overlap): %.1f GB/s over %.1f MB" % (alone_b / alone_t, alone_b / 1e6))
    if beside_t: print("#   pieces beside decode kernels (>80 %% overlap):            %.1f GB/s over %.1f MB" % (beside_b / beside_t, beside_b / 1e6))
small = [c for c in call_c if c[3] < (1 << 20)]
print("# %d small copies (< 1 MiB), %.2f ms of copy time in all" % (len(small), sum(e - s for s, e, *_ in small) / 1e6))
print("# every dispatch >= 0.2 ms, and the decode / search kernels: start ms, ms, kernel, queue")
for s, e, n, q in call_k:
    if (e - s) > 200_000 or "decode" in n or "search" in n:
        print("%9.3f %9.3f  %-44s q%s" % ((s - t0) / 1e6, (e - s) / 1e6, n, q))
bysum = collections.defaultdict(lambda: [0, 0])
for s, e, n, q in call_k:
    bysum[n][0] += 1; bysum[n][1] += e - s
print("# per kernel: dispatches, summed ms")
for n, (c, t) in sorted(bysum.items(), key=lambda kv: -kv[1][1]):
    print("%-46s %5d %9.3f" % (n, c, t / 1e6))
