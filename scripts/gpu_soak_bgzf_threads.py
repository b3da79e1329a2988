#!/usr/bin/env python3
"""Soak of the BGZF device path's member-walk thread and staging buffers (r5): several host threads of ONE process count BGZF files of different
shapes at the same time (each session on a context of its own: own walk thread, own fine-grained staging buffers), over and over, every row
against the host readers' (SCFQ_BGZF_DEVICE=0 in a child process).  Shapes: full members, short members, empty members in between, a file that
stops being BGZF half way (an ordinary gzip member follows: the device path hands the file over), chunk sizes from one launch to many.
usage: gpu_soak_bgzf_threads.py [rounds] [threads]"""
import os, random, struct, subprocess, sys, threading, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "seq-collection_amd", "pyhost"))
from test_ingest_sources import fastq_bytes
import scfq
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n_threads = int(sys.argv[2]) if len(sys.argv) > 2 else 4
SC = os.path.join(ROOT, "seq-collection_amd", "sc")
rng = random.Random(11)


def member(raw, level=6):
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    d = co.compress(raw) + co.flush()
    return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(d) + 25) + d + struct.pack("<II", zlib.crc32(raw), len(raw)))


EOFM = member(b"")
files = []
for k, (n, lo, hi, empties, tail) in enumerate(((200_000_000, 65280, 65280, 0, b""), (60_000_000, 100, 65280, 40, b""), (20_000_000, 1, 3000, 400, b""),
                                                (80_000_000, 30000, 65280, 3, b"gz"), (300_000_000, 65280, 65280, 0, b""))):
    data = fastq_bytes(n, seed=1300 + k)
    out, o = [], 0
    while o < len(data):
        if empties and rng.random() < empties / (len(data) / ((lo + hi) / 2)):
            out.append(EOFM)
        m = rng.randint(lo, hi)
        out.append(member(data[o:o + m], rng.choice((1, 6))))
        o += m
    blob = b"".join(out) + EOFM
    if tail == b"gz":      # an ordinary gzip member behind the BGZF ones: gzread reads on through it, and so must we
        extra = fastq_bytes(2_000_000, seed=77)
        co = zlib.compressobj(6, zlib.DEFLATED, 31)
        blob += co.compress(extra) + co.flush()
    path = "/tmp/soak_bgzf_%d.fq.gz" % k
    open(path, "wb").write(blob)
    r = subprocess.run([SC, "fq-count", path], capture_output=True, text=True, env=dict(os.environ, SCFQ_BGZF_DEVICE="0", SCFQ_GZ_DEVICE="0"))
    assert r.returncode == 0, r.stderr
    c = r.stdout.strip().split("\t")
    files.append((path, (int(c[0]), int(c[2]), int(c[3]), int(c[4]))))
    print("file", k, len(blob), "bytes", files[-1][1], flush=True)
bad = []


def worker(t):
    r = random.Random(100 + t)
    for i in range(rounds):
        path, want = r.choice(files)
        c = scfq.count_file(path)
        got = (c.reads, c.gc_bases, c.n_bases, c.bases)
        if got != want:
            bad.append((t, i, path, got, want))


th = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
for x in th: x.start()
for x in th: x.join()
for p, _ in files: os.remove(p)
print("soak: %d threads x %d counts, %d wrong" % (n_threads, rounds, len(bad)), bad[:3])
sys.exit(1 if bad else 0)
