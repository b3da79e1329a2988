#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + PMC counters) into a small text summary."""
import csv, glob, os, sys, collections
out = sys.argv[1]
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats:", os.path.relpath(f, out))
    for row in csv.DictReader(open(f)):
        print("  %-60s calls=%s avg_ns=%s total_ns=%s pct=%s" % (row.get("Name", "")[:60], row.get("Calls"), row.get("AverageNs"), row.get("TotalDurationNs"), row.get("Percentage")))
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            k = (row.get("Kernel_Name", "")[:40], row.get("Counter_Name"))
            agg[k][0] += float(row.get("Counter_Value", 0) or 0)
            agg[k][1] += 1
        print("== pmc:", os.path.relpath(f, out))
        for (kn, cn), (v, n) in sorted(agg.items()):
            if "fq_" in kn:
                print("  %-40s %-28s per_dispatch=%.6g dispatches=%d" % (kn, cn, v / max(n, 1), n))
