#!/usr/bin/env python3
"""What a fresh `sc fq-count --stats FILE` PROCESS pays for the 10 GB gzip member (BASELINE configs[3]) under different settings of the
device gzip path: wall of the whole process (start to reaped), the library's stage marks, device memory held.  Every variant runs
`runs` processes one right after the other (as bench.py's cold legs and a shell loop do), variants in the given order, twice over.
usage: measure_cold_ab.py [inflated bytes] [runs] ['name|ENV=1 ENV2=2;name2|...'] [bgzf]"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SC = os.path.join(ROOT, "seq-collection_amd", "sc")
n = sys.argv[1] if len(sys.argv) > 1 else "10e9"
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
variants = [(v.split("|")[0], dict(kv.split("=", 1) for kv in v.split("|")[1].split())) for v in (sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] != "x" else "default|SCFQ_NOTHING=1").split(";")]
gz = "/tmp/cold_ab.fq.gz"
info = json.loads(subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "write_pigz_member.py"), n, gz] + (["--bgzf"] if len(sys.argv) > 4 and sys.argv[4] == "bgzf" else []),
                                 capture_output=True, text=True, check=True).stdout.strip().splitlines()[-1])
want = (info["records"], info["gc_bases"], info["n_bases"], info["bases"])
os.sync()
time.sleep(1.0)
for rep in range(2):
    for name, env in variants:
        walls, marks, mem = [], [], 0
        for _ in range(runs):
            t = time.perf_counter()
            r = subprocess.run([SC, "fq-count", "--stats", gz], capture_output=True, text=True, env=dict(os.environ, **env))
            walls.append(round((time.perf_counter() - t) * 1e3, 1))
            c = r.stdout.strip().split("\t")
            assert r.returncode == 0 and (int(c[0]), int(c[2]), int(c[3]), int(c[4])) == want, (r.stdout, r.stderr[-600:])
            st = [json.loads(l) for l in r.stderr.splitlines() if l.startswith("{") and "stages_ms" in l][-1]
            mem = st["device_bytes_high_water"]
            m = dict((k, v) for k, v in st["stages_ms"])
            marks.append({"runtime_up": m.get("runtime initialised (hipGetDevice returned)"), "context_up": m.get("context up"),
                          "first_copy_queued": m.get("gzip engine: first batch's compressed bytes written to the device", m.get("gzip engine: first batch's compressed bytes queued for the device", m.get("BGZF: first chunk's compressed bytes queued for the device"))),
                          "first_decode_queued": m.get("gzip engine: first decode kernel queued", m.get("BGZF: first inflate kernel queued")),
                          "folded": m.get("session folded"), "row_computed": m.get("sc: row computed"), "ingest_wall_ms": round(st["ingest_wall_ms"], 1)})
        print(json.dumps({"variant": name, "env": env, "walls_ms": walls, "median_ms": sorted(walls)[len(walls) // 2], "device_GB": round(mem / 1e9, 2), "marks": marks}), flush=True)
os.remove(gz)
