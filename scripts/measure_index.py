"""K5 line index over a device-resident 10 GB Illumina FASTQ: wall of one scfq_index_lines call (count + offsets), the compact form
(default) and the mask form (SCFQ_INDEX_COMPACT=0), each in a process of its own.  usage: python scripts/measure_index.py [bytes=10e9]"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "seq-collection_amd", "pyhost"))


def child(nbytes):
    import torch
    import scfq
    plan = scfq.synth_plan(0, 20260101, nbytes)
    buf = torch.empty(plan.bytes + 4096, dtype=torch.uint8, device="cuda")
    scfq.synth_device(0, 20260101, plan.records, buf.data_ptr(), plan.bytes)
    lines = 4 * plan.records
    off = torch.empty(lines + 1, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    walls = []
    for _ in range(7):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        got = scfq.index_lines_device(buf.data_ptr(), plan.bytes, off.data_ptr(), lines + 1)
        walls.append(time.perf_counter() - t0)
        assert got == lines
    # spot check: every 4th line starts with '@', the sentinel is one past the input's last byte + 1
    o = off[::4][:1000].cpu()
    assert bool((buf[o] == ord("@")).all()) and int(off[lines]) == plan.bytes
    print(json.dumps({"form": "mask" if os.environ.get("SCFQ_INDEX_COMPACT") == "0" else "compact", "bytes": plan.bytes, "lines": lines,
                      "wall_ms": round(min(walls[1:]) * 1e3, 3), "first_call_ms": round(walls[0] * 1e3, 3),
                      "input_GBps": round(plan.bytes / min(walls[1:]) / 1e9, 1)}), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(float(sys.argv[2])))
    else:
        n = sys.argv[1] if len(sys.argv) > 1 else "10e9"
        for env in ({}, {"SCFQ_INDEX_COMPACT": "0"}, {}, {"SCFQ_INDEX_COMPACT": "0"}):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", n], env=dict(os.environ, **env), capture_output=True, text=True)
            assert r.returncode == 0, r.stderr[-2000:]
            print(r.stdout.strip().splitlines()[-1], flush=True)
