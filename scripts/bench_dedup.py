"""fq-dedup measurement: the device pipeline over an HBM-resident synthetic Illumina FASTQ in which a fraction of the
records re-appears later (same bytes), next to the CPU restatement (oracle) on a bounded sample.
usage: python scripts/bench_dedup.py [bytes=10e9] [dup_fraction=0.2] [reps=5]"""
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "seq-collection_amd", "pyhost"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import scfq


def main():
    nbytes = int(float(sys.argv[1])) if len(sys.argv) > 1 else int(10e9)
    frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.2
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    seed = 20260101
    uniq_bytes = int(nbytes / (1 + frac))
    plan = scfq.synth_plan(0, seed, uniq_bytes)
    dup_plan = scfq.synth_plan(0, seed, int(uniq_bytes * frac))          # the first records again: all duplicates
    n = plan.bytes + dup_plan.bytes
    buf = torch.empty(n + 4096, dtype=torch.uint8, device="cuda")
    scfq.synth_device(0, seed, plan.records, buf.data_ptr(), plan.bytes)
    buf[plan.bytes:n] = buf[:dup_plan.bytes]
    torch.cuda.synchronize()
    out = torch.empty(n + 4096, dtype=torch.uint8, device="cuda")
    times = []
    for r in range(reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nb, st = scfq.dedup_device(buf.data_ptr(), n, out.data_ptr(), n)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    best = min(times[1:])
    print("rep times ms:", [round(t * 1e3, 1) for t in times], file=sys.stderr)
    # generator IDs (lane:tile:x:y) can repeat by chance: the exact duplicate count comes from the pipeline itself and is
    # checked against the oracle on the sample below; the copies appended above are a lower bound
    assert st.duplicates >= dup_plan.records and st.total_reads == plan.records + dup_plan.records
    # CPU restatement on a bounded sample (first 256 MiB + its own duplicates region is not needed: any prefix works)
    sample = min(n, 256 << 20)
    host = buf[:sample].cpu().numpy()
    import conftest
    L = ctypes.CDLL(conftest._build_oracle())
    L.oracle_dedup.restype = ctypes.c_int64
    L.oracle_dedup.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    o = (ctypes.c_uint8 * (sample + 16))()
    ost = conftest.OracleDedupStats()
    t0 = time.perf_counter()
    m = L.oracle_dedup(host.ctypes.data, sample, o, sample + 16, ctypes.byref(ost))
    cpu_s = time.perf_counter() - t0
    nb_s, st_s = scfq.dedup_device(buf.data_ptr(), sample, out.data_ptr(), n)
    assert nb_s == m and st_s.duplicates == ost.duplicates and out[:nb_s].cpu().numpy().tobytes() == bytes(o[:m])
    print(json.dumps({
        "metric": "fq-dedup input GB/s (HBM-resident in, HBM-resident out)", "value": round(n / best / 1e9, 2), "unit": "GB/s",
        "records_per_s": round((plan.records + dup_plan.records) / best), "ms": round(best * 1e3, 2), "bytes": n,
        "records": plan.records + dup_plan.records, "duplicates": st.duplicates, "bytes_out": nb,
        "hash_collisions": st.hash_collisions,
        "cpu_baseline": {"value": round(sample / cpu_s / 1e9, 3), "unit": "GB/s", "cores": 1, "kind": "port",
                         "sample": "first %d MiB of the same input through oracle_dedup (exact hash set), output identical to the device's" % (sample >> 20)},
    }))


if __name__ == "__main__":
    main()
