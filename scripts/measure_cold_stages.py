#!/usr/bin/env python3
"""What a PROCESS pays: `sc fq-count --stats FILE` as a fresh process, several times per file, with the library's stage marks
(ms since the library was loaded: runtime initialised, context up, buffers, first copy, first kernel, folded) of the median run.
Files: one gzip member written the way pigz does it (zlib level 6, 64 MiB pieces joined by sync flushes) at each size given, and
a BGZF file of the last size.  Counters are checked against the generator's tallies.
usage: measure_cold_stages.py [tmpdir] [runs] [inflated sizes, e.g. 0.5e9,2e9]      env A/B: SCFQ_PREFAULT=0 ..."""
import json, os, struct, subprocess, sys, time, zlib
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "seq-collection_amd", "pyhost"))
import scfq

tmp = sys.argv[1] if len(sys.argv) > 1 else "/tmp"
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 5
sizes = [int(float(x)) for x in (sys.argv[3] if len(sys.argv) > 3 else "0.5e9,2e9").split(",")]
sc = os.path.join(ROOT, "seq-collection_amd", "sc")


def write_member(data, path):
    step = 64 << 20
    cuts = list(range(0, data.size, step))

    def piece(i):
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        return co.compress(data[cuts[i]:cuts[i] + step].tobytes()) + co.flush(zlib.Z_FINISH if i == len(cuts) - 1 else zlib.Z_SYNC_FLUSH)
    with ThreadPoolExecutor(16) as ex:
        parts = list(ex.map(piece, range(len(cuts))))
    crc = 0
    for c0 in cuts:
        crc = zlib.crc32(data[c0:c0 + step], crc)
    with open(path, "wb") as f:
        f.write(b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03")
        for b in parts:
            f.write(b)
        f.write(struct.pack("<II", crc & 0xFFFFFFFF, data.size & 0xFFFFFFFF))


def write_bgzf(data, path):
    def block(b):
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        payload = co.compress(b) + co.flush()
        return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 18 + len(payload) + 8 - 1) + payload +
                struct.pack("<II", zlib.crc32(b) & 0xFFFFFFFF, len(b)))

    def span(i):
        a = data[i:i + (32 << 20)]
        return b"".join(block(a[o:o + 65280].tobytes()) for o in range(0, a.size, 65280))
    with ThreadPoolExecutor(16) as ex:
        spans = list(ex.map(span, range(0, data.size, 32 << 20)))
    with open(path, "wb") as f:
        for s_ in spans:
            f.write(s_)
        f.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))


def cold(path, want, label):
    os.sync()          # (a process that starts while the kernel writes back what this script has just written pays for it in its runtime initialisation)
    time.sleep(0.3)
    rows = []
    for _ in range(runs):
        t = time.perf_counter()
        r = subprocess.run([sc, "fq-count", "--stats", path], capture_output=True, text=True)
        wall = time.perf_counter() - t
        f_ = r.stdout.strip().split("\t")
        assert r.returncode == 0 and (int(f_[0]), int(f_[2]), int(f_[3]), int(f_[4])) == want, (r.stdout, r.stderr[-800:])
        st = None
        for line in r.stderr.splitlines():
            if line.startswith("{") and "stages_ms" in line:
                st = json.loads(line)
        rows.append((wall, st))
    order = [round(w * 1e3, 1) for w, _ in rows]
    rows.sort(key=lambda x: x[0])
    med = rows[len(rows) // 2]
    out = {"file": label, "bytes": os.path.getsize(path), "runs_ms_in_order": order, "min_ms": round(rows[0][0] * 1e3, 1), "median_ms": round(med[0] * 1e3, 1),
           "max_ms": round(rows[-1][0] * 1e3, 1), "median_run_marks": med[1]["stages_ms"] if med[1] else None,
           "device_bytes_high_water": med[1]["device_bytes_high_water"] if med[1] else None}
    print(json.dumps(out), flush=True)


for n in sizes:
    plan = scfq.synth_plan(0, 20260101, n)
    data, info = scfq.synth_host(0, 20260101, plan.records)
    want = (plan.records, info.gc_bases, info.n_bases, info.bases)
    p = os.path.join(tmp, "scfq_stage_member_%d.fq.gz" % n)
    write_member(data, p)
    cold(p, want, "gzip member, %.1f GB inflated" % (data.size / 1e9))
    os.remove(p)
    if n == sizes[-1]:
        p = os.path.join(tmp, "scfq_stage_bgzf_%d.fq.gz" % n)
        write_bgzf(data, p)
        cold(p, want, "BGZF, %.1f GB inflated" % (data.size / 1e9))
        os.remove(p)
        p = os.path.join(tmp, "scfq_stage_plain_%d.fq" % n)
        data.tofile(p)
        cold(p, want, "plain FASTQ, %.1f GB" % (data.size / 1e9))
        os.remove(p)
