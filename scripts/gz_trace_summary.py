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
# (this rocprofv3's memory-copy trace carries no byte count: the ring piece's size is given on the command line — SCFQ_VERBOSE prints
# it — and applies to the pieces whose duration is within 35 % of the median; a batch's last piece is shorter)
piece_mb = float(sys.argv[2]) if len(sys.argv) > 2 else 128.0
h2d = [c for c in call_c if "HOST_TO_DEVICE" in c[2].upper() and (c[1] - c[0]) > 300_000]
print("# H2D copies longer than 0.3 ms (the ring's pieces; agents %s): start ms, ms, GB/s if a full piece of %.0f MiB, share of its time with a decode kernel running" % (sorted(set(c[4] + "->" + c[5] for c in h2d)), piece_mb))
if h2d:
    durs = sorted(c[1] - c[0] for c in h2d)
    med = durs[len(durs) // 2]
    alone, beside = [], []
    for s_, e_, dr, b, sa, da, st in h2d:
        ov = overlap(s_, e_) / max(1, e_ - s_)
        full = abs((e_ - s_) - med) < 0.35 * med
        rate = piece_mb * 1.048576e6 / (e_ - s_) if full else 0.0
        print("%9.3f %8.3f %7s  %4.0f %%" % ((s_ - t0) / 1e6, (e_ - s_) / 1e6, ("%.1f" % rate) if full else "-", 100 * ov))
        if full: (beside if ov > 0.8 else alone if ov < 0.2 else []).append(rate)
    span = max(c[1] for c in h2d) - min(c[0] for c in h2d)
    busy = sum(c[1] - c[0] for c in h2d)
    print("# %d pieces, %.1f ms of copy time inside %.1f ms from the first piece's start to the last one's end (the engine idle %.1f ms of it)" % (len(h2d), busy / 1e6, span / 1e6, (span - busy) / 1e6))
    if alone: print("#   full pieces with no decode kernel beside them: %d, %.1f GB/s on average" % (len(alone), sum(alone) / len(alone)))
    if beside: print("#   full pieces beside decode kernels (> 80 %% of their time): %d, %.1f GB/s on average" % (len(beside), sum(beside) / len(beside)))
other = [c for c in call_c if (c[1] - c[0]) > 300_000 and "HOST_TO_DEVICE" not in c[2].upper()]
if other:
    print("# other copies longer than 0.3 ms (the host-writes feed's staging buffer -> batch buffer, device to device): start ms, ms, direction")
    for s_, e_, dr, b, sa, da, st in other:
        print("%9.3f %8.3f  %s" % ((s_ - t0) / 1e6, (e_ - s_) / 1e6, dr))
small = [c for c in call_c if (c[1] - c[0]) <= 300_000]
print("# %d short copies (tables, one-byte parks; <= 0.3 ms each), %.2f ms of copy time in all" % (len(small), sum(c[1] - c[0] for c in small) / 1e6))
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
