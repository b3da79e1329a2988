#!/usr/bin/env python3
"""BASELINE configs[3]: one gzip -6 member of the synthetic Illumina stream, counted through the device-side inflate
(default) and through the host path behind it (SCFQ_GZ_DEVICE=0).  usage: measure_gz_device.py [inflated bytes] [tmpdir]
Prints one JSON object per line; the device path's phase times (SCFQ_VERBOSE laps of its best run) go into "phases_ms"."""
import json, os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "seq-collection_amd", "pyhost"))

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import scfq
    path = sys.argv[2]
    scfq.count_file(os.path.join(ROOT, "tests", "golden", "dup.fq.gz"))      # context + module warm-up
    for rep in range(3):
        sys.stderr.write("scfq rep %d\n" % rep); sys.stderr.flush()
        t = time.time(); c = scfq.count_file(path, flags=scfq.SCFQ_TIMING); dt = time.time() - t
        tm = scfq.last_timing()
        print(json.dumps({"counts": [c.reads, c.gc_bases, c.n_bases, c.bases], "bytes": c.input_bytes, "wall_s": round(dt, 4),
                          "inflated_GBps": round(c.input_bytes / dt / 1e9, 3), "host_fill_ms": round(tm.host_fill_ms, 1),
                          "scan_kernel_ms": round(tm.scan_kernel_ms, 3), "h2d_bytes": tm.h2d_bytes}), flush=True)
    sys.exit(0)

import torch      # (before the library: both then share ONE HIP runtime — torch ships its own libamdhip64, and a device pointer of one runtime means nothing to the other)
import scfq
nbytes = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2 << 30
tmp = sys.argv[2] if len(sys.argv) > 2 else "/tmp"
plan = scfq.synth_plan(0, 20260101, nbytes)
# (generated on the device and copied back: the host generator — the same pure function — needs 9 s per GB)
_buf = torch.empty(plan.bytes + 4096, dtype=torch.uint8, device="cuda")
info = scfq.synth_device(0, 20260101, plan.records, _buf.data_ptr(), plan.bytes)
data = _buf[:plan.bytes].cpu().numpy()
del _buf
torch.cuda.empty_cache()
plain = os.path.join(tmp, "scfq_gzd.fq")
data.tofile(plain)
t0 = time.time()
if os.environ.get("SCFQ_MEASURE_PLAIN_GZIP"):
    subprocess.check_call(["gzip", "-6", "-k", "-f", plain])          # one zlib stream, one thread: ~75 s per GB
    how = "gzip -6"
else:
    # ONE member written the way pigz does it: level-6 raw deflate of 64 MiB pieces on 16 threads, every piece but the last
    # ended with a sync flush (an empty stored block), the last with the final block; CRC-32 and ISIZE of the whole input
    import zlib
    from concurrent.futures import ThreadPoolExecutor
    raw = data
    step = 64 << 20
    cuts = list(range(0, raw.size, step))
    def piece(i):
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        chunk = raw[cuts[i]:cuts[i] + step].tobytes()
        return co.compress(chunk) + co.flush(zlib.Z_FINISH if i == len(cuts) - 1 else zlib.Z_SYNC_FLUSH), 0, len(chunk)
    with ThreadPoolExecutor(16) as ex:
        parts = list(ex.map(piece, range(len(cuts))))
    crc = 0
    for c0 in cuts:
        crc = zlib.crc32(raw[c0:c0 + step], crc)
    with open(plain + ".gz", "wb") as f:
        f.write(b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03")
        for b, _, _ in parts:
            f.write(b)
        f.write(int(crc & 0xFFFFFFFF).to_bytes(4, "little") + int(raw.size & 0xFFFFFFFF).to_bytes(4, "little"))
    how = "zlib level 6, one member, 64 MiB pieces joined by sync flushes (as pigz writes it)"
sys.stderr.write("%s took %.0f s\n" % (how, time.time() - t0))
expect = [plan.records, info.gc_bases, info.n_bases, info.bases]
modes = [("device", {"SCFQ_VERBOSE": "1"}), ("host (parallel single-member reader)", {"SCFQ_GZ_DEVICE": "0"})]
if os.environ.get("SCFQ_MEASURE_VARIANTS"):      # A/B runs of the device path: JSON list of {"name": ..., "env": {...}}
    modes = [("device " + v["name"], dict(v["env"], SCFQ_VERBOSE="1")) for v in json.loads(os.environ["SCFQ_MEASURE_VARIANTS"])]
for mode, env in modes:
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", plain + ".gz"], env=dict(os.environ, **env), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    if os.environ.get("SCFQ_MEASURE_LOG"):
        with open(os.environ["SCFQ_MEASURE_LOG"], "a") as lf:
            lf.write("==== %s\n%s\n" % (mode, "\n".join(l for l in r.stderr.splitlines() if not l.startswith("scfq pgz"))))
    rows = [json.loads(l) for l in r.stdout.strip().splitlines() if l.startswith("{")]
    best = min(rows, key=lambda j: j["wall_s"])
    assert best.pop("counts") == expect
    best.update({"path": how, "inflate": mode, "gz_bytes": os.path.getsize(plain + ".gz"), "first_call_wall_s": rows[0]["wall_s"]})
    if mode.startswith("device"):
        reps = re.split(r"scfq rep \d+\n", r.stderr)
        k = rows.index(min(rows, key=lambda j: j["wall_s"])) + 1
        best["phases_ms"] = {m.group(1).strip(): float(m.group(2)) for m in re.finditer(r"scfq gzdev: ([a-zA-Z|\- ]+?)\s+([0-9.]+) ms", reps[k])}
        best["summary"] = [l for l in reps[k].splitlines() if "on the chain" in l or "round" in l or "host:" in l or "copier thread" in l or "high water" in l]
    print(json.dumps(best), flush=True)
os.remove(plain); os.remove(plain + ".gz")
