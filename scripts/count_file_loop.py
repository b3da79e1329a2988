#!/usr/bin/env python3
"""count one file N times (for profiling runs): count_file_loop.py <path> [n]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "seq-collection_amd", "pyhost"))
import scfq
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
for _ in range(n):
    t = time.time()
    c = scfq.count_file(sys.argv[1])
    print(c.reads, c.bases, round((time.time() - t) * 1e3, 1), "ms", flush=True)
