#!/usr/bin/env python3
"""ONE gzip member of BASELINE configs[3]'s size counted by 1, 2 and 4 `sc fq-count --shard-rank` processes on one device (TCP transport):
the group's wall, every rank's ingest wall, compressed bytes moved, batches and device-path phases (SCFQ_VERBOSE), one JSON line per
configuration.  usage: measure_gz_sharded.py [inflated bytes] [out.jsonl]"""
import json, os, re, socket, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SC = os.path.join(ROOT, "seq-collection_amd", "sc")
n = sys.argv[1] if len(sys.argv) > 1 else "10e9"
out = sys.argv[2] if len(sys.argv) > 2 else "/dev/stdout"
gz = "/tmp/gzs_member.fq.gz"
info = json.loads(subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "write_pigz_member.py"), n, gz], capture_output=True, text=True, check=True).stdout.strip().splitlines()[-1])
want = (info["records"], info["gc_bases"], info["n_bases"], info["bases"])


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run(world, **env):
    port = free_port()
    t0 = time.time()
    if world == 1:
        procs = [subprocess.Popen([SC, "fq-count", "--stats", gz], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=dict(os.environ, SCFQ_VERBOSE="1", **env))]
    else:
        procs = [subprocess.Popen([SC, "fq-count", "--shard-rank=%d" % r, "--shard-world=%d" % world, "--rendezvous=127.0.0.1:%d" % port, "--transport=tcp", "--devices=0",
                                   "--stats", gz], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=dict(os.environ, SCFQ_VERBOSE="1", **env)) for r in reversed(range(world))]
    outs = [p.communicate(timeout=600) for p in procs][::-1]
    wall = time.time() - t0
    assert all(p.returncode == 0 for p in procs), [o[1][-800:] for o in outs]
    c = outs[0][0].strip().split("\t")
    assert (int(c[0]), int(c[2]), int(c[3]), int(c[4])) == want, outs[0][0]
    ranks = []
    for so, se in outs:
        st = [json.loads(l) for l in se.splitlines() if l.startswith("{")][-1]
        ph = {}
        for m in re.finditer(r"scfq gzdev: ([a-zA-Z|\- ]+?)\s+([0-9.]+) ms", se):
            ph.setdefault(m.group(1).strip(), []).append(float(m.group(2)))
        ranks.append({"ingest_wall_ms": round(st["ingest_wall_ms"], 1), "h2d_bytes": st["h2d_bytes"], "device_bytes_high_water": st["device_bytes_high_water"],
                      "batches": [int(l.split(" batch(es)")[0].split()[-1]) for l in se.splitlines() if " batch(es), " in l and "segments planned" in l],
                      "phases_ms": ph, "host": [l.split("scfq gzdev: ", 1)[1] for l in se.splitlines() if "inside device allocations" in l or "copier thread" in l],
                      "stages_ms": st.get("stages_ms")})
    return {"world": world, "env": env, "group_wall_s": round(wall, 3), "ranks": ranks}


with open(out, "a") as f:
    for world, env in ((1, {}), (2, {}), (4, {}), (4, {"SCFQ_SHARD_GZ_KEEP": "0"}), (2, {"SCFQ_SHARD_GZ_KEEP": "0"}), (1, {})):
        time.sleep(4.0)      # (the driver wipes what the processes before freed, 16 - 70 GB here, for a second or more: a group that starts into that waits for it)
        row = run(world, **env)
        row.update({"inflated_bytes": info["inflated_bytes"], "gz_bytes": info["gz_bytes"]})
        f.write(json.dumps(row) + "\n")
        f.flush()
os.remove(gz)
