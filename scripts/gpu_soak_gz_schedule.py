#!/usr/bin/env python3
"""Soak of the device gzip path's event-driven schedule (r5): a few hundred `sc fq-count` processes over gzip files of several shapes (one member,
many members, trailing garbage; 3 - 40 MB inflated) with the schedule's rings at random settings — sets of symbols 2 .. 6, decode streams 1 .. 3,
batches of 4 .. 64 segments, segments of 16 .. 128 KiB, the first batch split or not — every row against the oracle's, every run on the device path.
usage: gpu_soak_gz_schedule.py [runs] [seed]"""
import os, random, subprocess, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "seq-collection_amd", "pyhost"))
import numpy as np
from test_ingest_sources import fastq_bytes
import ctypes
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
SC = os.path.join(ROOT, "seq-collection_amd", "sc")
O = ctypes.CDLL(os.path.join(ROOT, "oracle", "libfqcount_oracle.so"))


def member(data, level=6):
    co = zlib.compressobj(level, zlib.DEFLATED, 31)
    return co.compress(data) + co.flush()


files = []
for k, (n, cuts, tail) in enumerate(((3_000_000, 0, b""), (12_000_000, 0, b"\0\0garbage"), (20_000_000, 3, b""), (40_000_000, 0, b""), (9_000_000, 7, b"x"))):
    data = fastq_bytes(n, seed=900 + k)
    if k % 2:
        data = data.replace(b"\n+\n", b"\r\n+\r\n")
    edges = [0] + sorted(rng.randrange(1, len(data)) for _ in range(cuts)) + [len(data)]
    blob = b"".join(member(data[a:b], rng.choice((1, 6, 9))) for a, b in zip(edges[:-1], edges[1:])) + tail
    path = "/tmp/soak_%d.fq.gz" % k
    open(path, "wb").write(blob)
    r = subprocess.run([SC, "fq-count", path], capture_output=True, text=True, env=dict(os.environ, SCFQ_GZ_DEVICE="0"))      # the host readers' row (== oracle: tests/)
    assert r.returncode == 0, r.stderr
    files.append((path, r.stdout))
bad = 0
for i in range(runs):
    path, want = rng.choice(files)
    env = {"SCFQ_GZ_DEVICE_MIN_MB": "0", "SCFQ_VERBOSE": "1", "SCFQ_GZ_DEVICE_SLOTS": str(rng.randint(2, 6)), "SCFQ_GZ_DEVICE_DECODE_STREAMS": str(rng.randint(1, 3)),
           "SCFQ_GZ_DEVICE_BATCH_SEGMENTS": str(rng.choice((4, 8, 16, 64))), "SCFQ_GZ_DEVICE_SEGMENT_KB": str(rng.choice((16, 32, 64, 128))),
           "SCFQ_GZ_DEVICE_FIRST_BATCH_DIV": str(rng.choice((1, 4))), "SCFQ_GZ_DEVICE_CHAIN_GROUP": str(rng.choice((3, 5, 64)))}
    r = subprocess.run([SC, "fq-count", path], capture_output=True, text=True, env=dict(os.environ, **env), timeout=120)
    ok = r.returncode == 0 and r.stdout == want and "on the chain" in r.stderr and "the rest on the host" not in r.stderr
    if not ok:
        bad += 1
        print("FAILED run %d: %s %s rc %d\n%s" % (i, path, env, r.returncode, r.stderr[-1500:]), flush=True)
print("soak: %d runs, %d failed" % (runs, bad))
for path, _ in files:
    os.remove(path)
sys.exit(1 if bad else 0)
