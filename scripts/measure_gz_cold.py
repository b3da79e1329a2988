#!/usr/bin/env python3
"""BASELINE configs[3] the way a user runs it: `sc fq-count x.fq.gz`, ONE PROCESS PER INVOCATION (sc.nim:114-116), several
invocations back to back.  Every row is the wall time of a whole process (start, context, buffers, ingest, exit) and is checked
against the generator's tallies.  The device memory a process frees is wiped by the driver after it exits, and the next process's
allocations may have to wait for that: the second and third invocation are the ones a shell loop over files sees.
usage: measure_gz_cold.py [inflated bytes] [tmpdir] [layout: pigz | gzip | bgzf] [invocations]"""
import json, os, re, struct, subprocess, sys, time, zlib
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "seq-collection_amd", "pyhost"))
import scfq

nbytes = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2 << 30
tmp = sys.argv[2] if len(sys.argv) > 2 else "/tmp"
layout = sys.argv[3] if len(sys.argv) > 3 else "pigz"
n_inv = int(sys.argv[4]) if len(sys.argv) > 4 else 3
plan = scfq.synth_plan(0, 20260101, nbytes)
data, info = scfq.synth_host(0, 20260101, plan.records)
path = os.path.join(tmp, "scfq_cold_%s.fq.gz" % layout)
t0 = time.time()
if layout == "gzip":
    # one zlib stream written by one thread, no sync flushes (what `gzip -6` writes): ~40 s per GB
    co = zlib.compressobj(6, zlib.DEFLATED, 31)
    with open(path, "wb") as f:
        step = 256 << 20
        for o in range(0, data.size, step):
            f.write(co.compress(data[o:o + step].tobytes()))
        f.write(co.flush())
    how = "gzip -6 layout (one deflate stream, no sync flushes)"
elif layout == "bgzf":
    def block(b):
        co = zlib.compressobj(6, zlib.DEFLATED, -15); payload = co.compress(b) + co.flush(); bs = 18 + len(payload) + 8
        return b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bs - 1) + payload + struct.pack("<II", zlib.crc32(b) & 0xFFFFFFFF, len(b))
    def span(i):
        a = data[i:i + (32 << 20)]
        return b"".join(block(a[o:o + 65280].tobytes()) for o in range(0, a.size, 65280))
    with ThreadPoolExecutor(16) as ex:
        spans = list(ex.map(span, range(0, data.size, 32 << 20)))
    with open(path, "wb") as f:
        for s in spans: f.write(s)
        f.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
    how = "BGZF (bgzip layout, 65280-byte blocks, level 6)"
else:
    step = 64 << 20
    cuts = list(range(0, data.size, step))
    def piece(i):
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        chunk = data[cuts[i]:cuts[i] + step].tobytes()
        return co.compress(chunk) + co.flush(zlib.Z_FINISH if i == len(cuts) - 1 else zlib.Z_SYNC_FLUSH)
    with ThreadPoolExecutor(16) as ex:
        parts = list(ex.map(piece, range(len(cuts))))
    crc = 0
    for c0 in cuts: crc = zlib.crc32(data[c0:c0 + step], crc)
    with open(path, "wb") as f:
        f.write(b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03")
        for b in parts: f.write(b)
        f.write(int(crc & 0xFFFFFFFF).to_bytes(4, "little") + int(data.size & 0xFFFFFFFF).to_bytes(4, "little"))
    how = "zlib level 6, one member, 64 MiB pieces joined by sync flushes (as pigz writes it)"
sys.stderr.write("%s: written in %.0f s\n" % (how, time.time() - t0))
expect = "%d\t" % plan.records
tail = "\t%d\t%d\t%d" % (info.gc_bases, info.n_bases, info.bases)
sc = os.path.join(ROOT, "seq-collection_amd", "sc")
rows = []
for inv in range(n_inv):
    t = time.time()
    r = subprocess.run([sc, "fq-count", "--stats", path], env=dict(os.environ, SCFQ_VERBOSE="1"), capture_output=True, text=True)
    dt = time.time() - t
    assert r.returncode == 0, r.stderr[-2000:]
    line = r.stdout.strip().splitlines()[-1]
    assert line.startswith(expect) and line.endswith(tail), (line, expect, tail)
    m = re.search(r"buffers grown in ([0-9.]+) ms", r.stderr)
    w = re.search(r"scfq gzdev: wall\s+([0-9.]+) ms", r.stderr)
    hw = re.search(r"device memory high water ([0-9.]+) GB", r.stderr)
    rows.append({"invocation": inv, "process_wall_s": round(dt, 4), "buffers_ms": float(m.group(1)) if m else None,
                 "ingest_wall_ms": float(w.group(1)) if w else None, "hbm_high_water_GB": float(hw.group(1)) if hw else None})
    sys.stderr.write(json.dumps(rows[-1]) + "\n")
    if os.environ.get("SCFQ_MEASURE_LOG"):
        with open(os.environ["SCFQ_MEASURE_LOG"], "a") as lf:
            lf.write("==== %s invocation %d\n%s\n" % (layout, inv, r.stderr))
out = {"path": how, "inflated_bytes": int(data.size), "gz_bytes": os.path.getsize(path), "what": "`sc fq-count --stats FILE`, one process per invocation, back to back",
       "process_wall_s": [r["process_wall_s"] for r in rows], "first_invocation_GBps": round(data.size / rows[0]["process_wall_s"] / 1e9, 2),
       "worst_invocation_GBps": round(data.size / max(r["process_wall_s"] for r in rows) / 1e9, 2), "rows": rows}
print(json.dumps(out), flush=True)
os.remove(path)
