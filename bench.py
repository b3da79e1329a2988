#!/usr/bin/env python3
"""bench.py — Gbases/s parsed by the fq-count hot path on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path (scan kernel K1 + ordered fold K2 + the partial exchange across
ranks) over one HBM-resident batch of synthetic FASTQ.  N=1 workload = BASELINE.json configs[1]:
synthetic 10 GB uncompressed 150 bp Illumina FASTQ (seed 20260101).  N>1 workload =
BASELINE.json configs[2]: every rank holds its own 25 GB byte range (200 GB / 8) of one N x 25 GB record
stream (seed 20260102), cut at ARBITRARY byte offsets (not record aligned), scans it, and the ranks
exchange their 32-word partials with one RCCL all-gather issued by the C library itself (scfq_comm_*: its own rendezvous,
librccl; the combine is ordered / non-commutative, so a sum-allreduce of counters would be wrong)
-> "scaling": "weak".  torch.distributed only provides the contract's barrier and the max over ranks (control plane: gloo by
default, so that the library's RCCL communicator is the only one a rank creates).  At N=1 the line also carries a non-headline
`ingest` object: a gzip member and a BGZF file of the same stream, written here, counted by a fresh `sc fq-count` process (cold)
and by the second call in this process (warm) — BASELINE configs[3] end to end, compressed bytes over PCIe, inflate on the device —
and a non-headline `dedup` object: the fq-dedup pipeline over a resident 10 GB input with one record in six repeated.

Launch:  python bench.py --gpus 1 --steps K --warmup W
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes
import json
import os
import sys
import subprocess
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "seq-collection_amd", "pyhost"))

HBM_PEAK_GBPS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
SEED = 20260101


def load_oracle():
    """cpu_baseline leg only: the CPU restatement (oracle/) as the timed "port" of the reference algorithm."""
    import subprocess
    so = os.path.join(ROOT, "oracle", "libfqcount_oracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    L = ctypes.CDLL(so)

    class OC(ctypes.Structure):
        _fields_ = [(n, ctypes.c_uint64) for n in
                    "reads gc_bases n_bases bases lines newlines input_bytes bad_at bad_plus".split()] + \
                   [("qual_hist", ctypes.c_uint64 * 256)]
    L.oracle_count_lines.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(OC)]
    L.oracle_partial.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64), ctypes.c_void_p]
    return L, OC


def cpu_all_cores(L, host, nbytes, threads):
    """BASELINE.md §3 "CPU-best" line: the same byte-range shards + ordered (+) fold as the GPU path, one oracle_partial per
    host thread (ctypes releases the GIL). Informational: not the cpu_baseline object."""
    from concurrent.futures import ThreadPoolExecutor
    bounds = [nbytes * t // threads for t in range(threads + 1)]
    outs = [(ctypes.c_uint64 * 27)() for _ in range(threads)]

    def work(t):
        lo, hi = bounds[t], bounds[t + 1]
        L.oracle_partial(host.ctypes.data + lo, hi - lo, int(host[lo - 1]) if lo else -1, outs[t], None)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(work, range(threads)))
    acc = [0] * 27
    for o in outs:
        k = acc[0] & 3
        new = list(acc)
        for arr in (1, 5, 9):
            for r in range(4):
                new[arr + r] = (acc[arr + r] + o[arr + ((r - k) & 3)]) & (2**64 - 1)
        new[0] = acc[0] + o[0]
        acc = new
    return time.perf_counter() - t0, acc


def ingest_rows(scfq, member_bytes, bgzf_bytes):
    """Non-headline: BASELINE configs[3] end to end — ONE gzip member of `member_bytes` of the synthetic stream (10 GB by default:
    configs[3]'s own size; written the way pigz does it) and a BGZF file of `bgzf_bytes`, written here and counted
    (a) by fresh `sc fq-count` processes, which is how the reference is used (one call per process, sc.nim:114-116): wall time of the
        whole process, context and buffers included, three times over with the library's stage marks;
    (b) by the first and second call in this process (buffers in place): compressed bytes cross PCIe, inflate on the device;
    (c) the member once more by a process with SCFQ_GZ_DEVICE=0: the HOST inflates (the library's parallel reader) into pinned
        buffers while a copy stream moves the chunk before to HBM and the compute stream scans the one before that — north_star's
        "gzip_stream.nim's host-side inflate overlapped with device compute on a second HIP stream".
    Counters must equal the generator's tallies everywhere."""
    import shutil
    import struct
    import subprocess
    import tempfile
    import zlib
    from concurrent.futures import ThreadPoolExecutor
    tmp = tempfile.mkdtemp(prefix="scfq_bench_", dir=os.environ.get("TMPDIR", "/tmp"))
    rows = {"what": "cold = wall of a fresh `sc fq-count FILE` process (median of three); warm = second call in one process; GB/s of inflated "
                    "bytes; counters == generator tallies"}
    sc = os.path.join(ROOT, "seq-collection_amd", "sc")

    def write_member(data, path):
        step = 64 << 20
        cuts = list(range(0, data.size, step))

        def piece(i):
            co = zlib.compressobj(6, zlib.DEFLATED, -15)
            chunk = data[cuts[i]:cuts[i] + step]
            return co.compress(chunk.tobytes()) + co.flush(zlib.Z_FINISH if i == len(cuts) - 1 else zlib.Z_SYNC_FLUSH), zlib.crc32(chunk), chunk.size
        with ThreadPoolExecutor(16) as ex:
            parts = list(ex.map(piece, range(len(cuts))))
        crc = 0
        for _, c_, n_ in parts:          # the CRC-32 of the whole from the pieces' (each computed beside its compression)
            crc = crc32_combine(crc, c_, n_)
        with open(path, "wb") as f:
            f.write(b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03")
            for b_, _, _ in parts:
                f.write(b_)
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

    def run_sc(path, want, env=None):
        t = time.perf_counter()
        r = subprocess.run([sc, "fq-count", "--stats", path], capture_output=True, text=True, env=dict(os.environ, **(env or {})))
        wall = time.perf_counter() - t
        f_ = r.stdout.strip().split("\t")
        assert r.returncode == 0 and (int(f_[0]), int(f_[2]), int(f_[3]), int(f_[4])) == want, ("cold row", r.stdout, r.stderr[-500:], want)
        st = None
        for line in r.stderr.splitlines():
            if line.startswith("{") and "stages_ms" in line:
                try:
                    st = json.loads(line)
                except ValueError:
                    pass
        return wall, st

    def stage_row(wall, st):
        if not st or not st.get("stages_ms"):
            return {"wall_ms": round(wall * 1e3, 1)}
        marks = st["stages_ms"]
        return {"wall_ms": round(wall * 1e3, 1), "marks": [[n, ms] for n, ms in marks],
                "process_start_and_exit_ms": round(wall * 1e3 - marks[-1][1], 1)}      # exec -> library loaded, plus row computed -> reaped

    def measure(name, path, how, data_size, want):
        # (the in-process calls come FIRST: a process that has just exited has its device memory wiped by the driver for a while —
        # 3 to 11 GB per cold process here — and calls that run beside that wipe measured twice their usual time)
        walls = []
        for _ in range(3):
            t = time.perf_counter()
            c = scfq.count_file(path)
            walls.append(time.perf_counter() - t)
            assert (c.reads, c.gc_bases, c.n_bases, c.bases) == want, ("in-process row", name)
        walls = [walls[0], min(walls[1:])]
        # a fresh process three times over, each with --stats: the library's stage marks (ms since it was loaded) say where a slow
        # one spent its time — runtime initialisation, context, allocations, first copy, first kernel, fold
        in_order = [run_sc(path, want) for _ in range(3)]
        colds = sorted(in_order, key=lambda x: x[0])
        cold = colds[1][0]
        # ... and three more with two seconds in front of each: a process that starts while the driver still reclaims what the one before held
        # (its runtime initialisation and its allocations wait for that: 50 - 700 ms, box by box) measures that reclaim, not itself
        spaced = []
        for _ in range(3):
            time.sleep(2.0)
            spaced.append(run_sc(path, want))

        def key_marks(wall, st):
            # the stamps that tell a slow RUNTIME from a slow library: when hipGetDevice returned, when the context was up, when the first
            # copy and the first inflate kernel were queued, when the session was folded (ms since the library was loaded)
            m = dict((n, ms) for n, ms in (st or {}).get("stages_ms") or [])
            pick = lambda *names: next((round(m[n], 1) for n in names if n in m), None)
            return {"wall_ms": round(wall * 1e3, 1), "runtime_initialised_ms": pick("runtime initialised (hipGetDevice returned)"), "context_up_ms": pick("context up"),
                    "first_copy_queued_ms": pick("gzip engine: first batch's compressed bytes written to the device", "gzip engine: first batch's compressed bytes queued for the device",
                                                 "BGZF: first chunk's compressed bytes queued for the device"),
                    "first_inflate_kernel_queued_ms": pick("gzip engine: first decode kernel queued", "BGZF: first inflate kernel queued"),
                    "session_folded_ms": pick("session folded"), "row_computed_ms": pick("sc: row computed"),
                    # which side a slow process was slow on: the scan kernels' device time (HIP events; 0.3 ms per GB when the device is
                    # itself) and the host's time filling the feed
                    "scan_kernel_ms": (st or {}).get("scan_kernel_ms"), "host_fill_ms": (st or {}).get("host_fill_ms"), "ingest_wall_ms": (st or {}).get("ingest_wall_ms")}
        rows[name] = {"layout": how, "inflated_bytes": int(data_size), "compressed_bytes": os.path.getsize(path),
                      "cold_process_wall_s": round(cold, 4), "cold_GBps": round(data_size / cold / 1e9, 2),
                      "cold_process_walls_s": {"min": round(colds[0][0], 4), "median": round(colds[1][0], 4), "max": round(colds[2][0], 4),
                                               "what": "three fresh `sc fq-count --stats FILE` processes one after the other; cold_process_wall_s is their median"},
                      "cold_runs_in_order": [key_marks(w, st) for w, st in in_order],
                      "cold_process_walls_2s_apart_s": {"median": round(sorted(w for w, _ in spaced)[1], 4), "in_order": [round(w, 4) for w, _ in spaced],
                                                        "runs": [key_marks(w, st) for w, st in spaced],
                                                        "what": "the same with a pause of 2 s before each process (not cold_process_wall_s: that stays the back-to-back median)"},
                      "cold_stages_ms": {"median_run": stage_row(*colds[1]), "slowest_run": stage_row(*colds[2]),
                                         "what": "[stage, ms since the library was loaded] (include/sc_fqcount_debug.h: scfq_debug_stages)"},
                      "first_call_wall_s": round(walls[0], 4), "warm_wall_s": round(walls[1], 4), "warm_GBps": round(data_size / walls[1] / 1e9, 2),
                      "warm_what": "the faster of the second and third call in this process", "counters_match_generator": True}

    def synth(nbytes):
        # (generated on the device and copied to the host: the host generator is the same pure function — tests/test_gpu_parity.py holds the
        # two to the same bytes — and needs 9 s per GB on one core)
        import torch
        plan = scfq.synth_plan(0, SEED, nbytes)
        dbuf = torch.empty(plan.bytes + 4096, dtype=torch.uint8, device="cuda")
        info = scfq.synth_device(0, SEED, plan.records, dbuf.data_ptr(), plan.bytes)
        assert info.bytes == plan.bytes
        host = dbuf[:plan.bytes].cpu().numpy()
        del dbuf
        torch.cuda.empty_cache()
        return plan, host, info

    try:
        # ---- BGZF, bgzf_bytes ----
        plan, data, info = synth(bgzf_bytes)
        want = (plan.records, info.gc_bases, info.n_bases, info.bases)
        bg = os.path.join(tmp, "bgzf.fq.gz")
        write_bgzf(data, bg)
        # (what has just been written is dirty page cache; a process that starts while the kernel is still writing it back pays for
        # that in its runtime initialisation — 190 ms instead of 52, scripts/measure_cold_stages.py — which says nothing about the
        # product: the cold runs start once the files are on disk)
        os.sync()
        measure("bgzf", bg, "BGZF (bgzip layout, 65280-byte blocks, level 6)", data.size, want)
        os.remove(bg)
        # ---- one gzip member, member_bytes (configs[3]) ----
        if member_bytes != bgzf_bytes:
            del data
            plan, data, info = synth(member_bytes)
            want = (plan.records, info.gc_bases, info.n_bases, info.bases)
        gz = os.path.join(tmp, "member.fq.gz")
        write_member(data, gz)
        n_inflated = int(data.size)
        del data
        os.sync()
        measure("gzip_member", gz, "one gzip member, zlib level 6, 64 MiB pieces joined by sync flushes (as pigz writes it)", n_inflated, want)
        # ---- the same member, inflated on the HOST and overlapped with copy + scan (the path behind the device inflate) ----
        wall, st = run_sc(gz, want, {"SCFQ_GZ_DEVICE": "0"})
        rows["gzip_member_host_inflate_overlap"] = {
            "what": "SCFQ_GZ_DEVICE=0: the library's parallel host reader fills pinned buffers while the copy stream moves the chunk before to "
                    "HBM and the compute stream scans the one before that; wall of a fresh process",
            "process_wall_s": round(wall, 4), "GBps": round(n_inflated / wall / 1e9, 2),
            "host_fill_ms": st and st.get("host_fill_ms"), "ingest_wall_ms": st and st.get("ingest_wall_ms"),
            "scan_kernel_ms": st and st.get("scan_kernel_ms"),
            "overlap": st and ("scan kernels took %.1f %% of the host's fill time: hidden under it" % (100.0 * st["scan_kernel_ms"] / max(st["host_fill_ms"], 1e-9))),
            "counters_match_generator": True}
        rows["device_bytes_high_water"] = int(scfq.lib().scfq_device_bytes_high_water())
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return rows


def dedup_row(scfq, torch, nbytes=int(10e9), frac=0.2, reps=5):
    """NON-headline: `sc fq-dedup`'s device pipeline (/root/reference/src/fq_dedup.nim:14-84) over an HBM-resident input
    of the headline workload's shape in which one record in six re-appears later — the input of scripts/bench_dedup.py, whose CPU
    comparison and oracle check live there and in tests/test_gpu_dedup.py.  Checked here by what needs no oracle: every record is
    echoed or counted, the appended copies are all found, and de-duplicating the result changes nothing."""
    seed = 20260101
    uniq = int(nbytes / (1 + frac))
    plan = scfq.synth_plan(0, seed, uniq)
    again = scfq.synth_plan(0, seed, int(uniq * frac))            # the first records once more: every one a duplicate
    n = plan.bytes + again.bytes
    buf = torch.empty(n + 4096, dtype=torch.uint8, device="cuda")
    scfq.synth_device(0, seed, plan.records, buf.data_ptr(), plan.bytes)
    buf[plan.bytes:n] = buf[:again.bytes]
    out = torch.empty(n + 4096, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    walls = []
    for _ in range(reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nb, st = scfq.dedup_device(buf.data_ptr(), n, out.data_ptr(), n)
        torch.cuda.synchronize()
        walls.append(time.perf_counter() - t0)
    best = min(walls[1:])
    records = plan.records + again.records
    assert st.total_reads == records and st.duplicates >= again.records and st.records_out == records - st.duplicates and nb < n
    nb2, st2 = scfq.dedup_device(out.data_ptr(), nb, buf.data_ptr(), n)      # (the input is not needed any more)
    assert nb2 == nb and st2.duplicates == 0 and st2.total_reads == st.records_out
    return {"what": "fq-dedup, input and result resident in HBM: line index, header hashes, radix sort, exact compares, gather (DESIGN.md §4); "
                    "wall of one scfq_dedup_buffer call, the fastest of %d after a first one" % reps,
            "input_bytes": n, "records": records, "duplicates": st.duplicates, "bytes_out": nb, "hash_collisions": st.hash_collisions,
            "wall_ms": round(best * 1e3, 3), "first_call_wall_ms": round(walls[0] * 1e3, 3), "input_GBps": round(n / best / 1e9, 1),
            "records_per_s": round(records / best), "idempotent": True}


def crc32_combine(crc1, crc2, len2):
    """zlib's crc32_combine (not exposed by Python's zlib): CRC-32 of A || B from crc(A), crc(B) and len(B)"""
    def times(mat, vec):
        s_ = 0
        i = 0
        while vec:
            if vec & 1:
                s_ ^= mat[i]
            vec >>= 1
            i += 1
        return s_

    def square(mat):
        return [times(mat, mat[n]) for n in range(32)]
    if len2 <= 0:
        return crc1
    odd = [0xEDB88320] + [1 << n for n in range(31)]
    even = square(odd)
    odd = square(even)
    while True:
        even = square(odd)
        if len2 & 1:
            crc1 = times(even, crc1)
        len2 >>= 1
        if not len2:
            break
        odd = square(even)
        if len2 & 1:
            crc1 = times(odd, crc1)
        len2 >>= 1
        if not len2:
            break
    return crc1 ^ crc2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--bytes-per-gpu", type=float, default=0,
                    help="bytes of FASTQ resident per GPU (default: 10e9 at N=1 = configs[1]; 25e9 at N>1 = configs[2], 200 GB / 8)")
    ap.add_argument("--workload", choices=["illumina", "nanopore"], default="illumina")
    ap.add_argument("--cpu-sample-bytes", type=float, default=10.1e9)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--flags", type=int, default=0, help="extra SCFQ_* flags (1 = qual hist, 2 = struct check)")
    ap.add_argument("--no-verify", action="store_true", help="diagnostic (ablation builds): skip the counter checks")
    ap.add_argument("--backend", default="gloo",
                    help="torch.distributed backend of the CONTROL plane (the contract's barrier and the max over ranks): gloo by default — the "
                         "data path's collective is the C library's own RCCL communicator, and a second RCCL communicator per rank (torch's) "
                         "would only be one more thing that can fail while eight ranks come up; nccl is accepted")
    ap.add_argument("--exchange", choices=["lib", "torch"], default="lib",
                    help="lib = the C library's own communicator (scfq_comm_*: RCCL ncclAllGather inside libsc_fqcount_hip; default); "
                         "torch = the Python mirror over torch.distributed (--backend says over what)")
    ap.add_argument("--transport", choices=["rccl", "tcp"], default="rccl",
                    help="transport of the library communicator: rccl (default) or tcp (rehearsals with several ranks on ONE device, which RCCL refuses)")
    ap.add_argument("--ingest-bytes", type=float, default=10e9,
                    help="N=1 only, after the timed region: inflated size of the gzip member of the non-headline `ingest` object — 10e9 = BASELINE "
                         "configs[3]'s own size (0 = skip the object)")
    ap.add_argument("--ingest-bgzf-bytes", type=float, default=2e9, help="... and of its BGZF file (at most --ingest-bytes)")
    ap.add_argument("--exchange-timeout-s", type=float, default=120.0, help="deadline of one exchange: a stuck collective ends the run non-zero")
    ap.add_argument("--same-device", action="store_true",
                    help="rehearsal only: every rank uses cuda:0 (1-GPU box, gloo backend); the number is not a scaling result")
    ap.add_argument("--exchange-at-1", action="store_true",
                    help="rehearsal only: run the N>1 exchange path (helper thread + collective) with a single rank")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import scfq
    import scfq_dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    xdev = dev if args.backend == "nccl" else None     # gloo exchanges CPU tensors
    exchange = world > 1 or args.exchange_at_1
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
    elif exchange:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        kw = {"device_id": dev} if args.backend == "nccl" else {}
        dist.init_process_group(args.backend, rank=0, world_size=1, **kw)
    assert args.gpus == world, "--gpus must equal the number of launched ranks"
    tmo_ms = int(args.exchange_timeout_s * 1e3)

    kind = 0 if args.workload == "illumina" else 1
    # N=1: BASELINE configs[1] (10 GB, seed 20260101); N>1: configs[2] (25 GB per GPU = 200 GB / 8, seed 20260102)
    per = int(args.bytes_per_gpu) if args.bytes_per_gpu > 0 else int(10e9 if world == 1 else 25e9)
    # `--bytes-per-gpu 25e9` at N=1 is the like-for-like anchor of the configs[2] curve: rank 0's 25 GB shard of the seed-20260102 stream
    anchor = world == 1 and kind == 0 and args.bytes_per_gpu > 0 and per != int(10e9)
    seed = (SEED if (world == 1 and not anchor) else 20260102) if kind == 0 else 20260103
    lo, hi = rank * per, (rank + 1) * per   # this rank's byte range of the N x per stream: arbitrary cut points

    # ---- build this rank's shard directly in HBM (not timed) ------------------------------------------
    t0 = time.time()
    first_rec, first_start = scfq.synth_locate(kind, seed, lo)
    last_rec, last_start = scfq.synth_locate(kind, seed, hi - 1)
    nrec = last_rec - first_rec + 1
    # last rank ends the stream on a record boundary (the file ends after a complete record)
    plan = scfq.synth_plan(kind, seed, 1, first_record=last_rec)   # length of the last record
    cover = last_start + plan.bytes - first_start
    buf = torch.empty(cover + 8192, dtype=torch.uint8, device=dev)
    info = scfq.synth_device(kind, seed, nrec, buf.data_ptr(), cover, first_record=first_rec)
    assert info.bytes == cover
    if rank == world - 1:
        hi = first_start + cover
    shard_ptr = buf.data_ptr() + (lo - first_start)
    shard_n = hi - lo
    prev_byte = -1 if lo == 0 else int(buf[lo - first_start - 1].item()) if lo > first_start else 10
    gen_s = time.time() - t0

    flags = scfq.SCFQ_TIMING | args.flags

    class LibExchange:
        """the exchange inside the C library (scfq_comm_*: pinned staging, ncclAllGather of ncclUint64 on a private stream,
        rank-ordered fold) — its worker thread runs the collective while this thread is inside the C call that scans the next
        step, so submit() never blocks and wait() normally finds the result ready"""
        def __init__(self):
            # The RCCL unique id travels through the LIBRARY's own rendezvous (scfq_comm_init_rendezvous: rank 0 listens, the ranks
            # connect); torch.distributed only tells the ranks which port rank 0 found free (a public collective on the control plane)
            port = [0]
            if rank == 0:
                import socket
                s_ = socket.socket()
                # (probed on the address the library's rank 0 will bind — MASTER_ADDR — not on loopback whatever that is)
                s_.bind((os.environ.get("MASTER_ADDR", "127.0.0.1"), 0))
                port[0] = s_.getsockname()[1]
                s_.close()
            if world > 1:
                dist.broadcast_object_list(port, src=0)
            transport = scfq.SCFQ_COMM_TCP if args.transport == "tcp" else scfq.SCFQ_COMM_RCCL
            self.comm = scfq.Comm.init_rendezvous(os.environ.get("MASTER_ADDR", "127.0.0.1"), port[0], world, rank, device=local_rank,
                                                  transport=transport, timeout_ms=tmo_ms)
            if transport == scfq.SCFQ_COMM_TCP:
                self.what = "all-gather of 32 x u64 partials over the library's TCP transport (scfq_comm_*, rehearsal: several ranks on one device) + rank-ordered fold"
            else:
                self.what = "%s ncclAllGather of 32 x u64 partials inside libsc_fqcount_hip (scfq_comm_*) + rank-ordered fold" % self.comm.transport

        def submit(self, partial):
            self.comm.start(partial, timeout_ms=tmo_ms)

        def wait(self):
            return scfq.finalize(self.comm.finish(timeout_ms=tmo_ms))

        def close(self):
            self.comm.destroy()

    class TorchExchange:
        """the Python mirror (pyhost/scfq_dist.py over torch.distributed) on ONE persistent helper thread: all_gather,
        device->host copy and the host fold of step k run while the main thread is inside the (GIL-free) C call that scans
        step k+1.  Used with the gloo backend (rehearsals on one device) and with --exchange torch."""
        def __init__(self):
            import queue
            import threading
            self.q_in, self.q_out = queue.SimpleQueue(), queue.SimpleQueue()
            self.t = threading.Thread(target=self._run, daemon=True)
            self.t.start()
            self.what = "%s all_gather of 32 x u64 partials via torch.distributed + rank-ordered fold" % ("RCCL" if args.backend == "nccl" else args.backend)

        def _run(self):
            torch.cuda.set_device(dev)          # the current device is thread-local
            self.dx = scfq_dist.DeviceExchange(dev) if xdev is not None else None
            while True:
                partial = self.q_in.get()
                if partial is None:
                    return
                try:
                    if self.dx is not None:          # RCCL: asynchronous calls only (scfq_dist.DeviceExchange)
                        self.dx.start(partial)
                        acc = self.dx.finish()
                    else:
                        acc, _ = scfq_dist.finish_exchange(scfq_dist.start_exchange(partial, device=xdev))
                    self.q_out.put(scfq.finalize(acc))
                except BaseException as e:      # surfaces in wait()
                    self.q_out.put(e)

        def submit(self, partial):
            self.q_in.put(partial)

        def wait(self):
            r = self.q_out.get(timeout=args.exchange_timeout_s)     # a stuck collective fails the run instead of hanging it
            if isinstance(r, BaseException):
                raise r
            return r

        def close(self):
            self.q_in.put(None)
            self.t.join(timeout=10)

    exchanger = None
    exchange_note = None
    if exchange:
        use_lib = args.exchange == "lib"
        if use_lib:
            # every rank must end up on the same path: a rank whose library communicator failed takes all ranks to the mirror
            ok, err = 1, ""
            try:
                exchanger = LibExchange()
            except Exception as e:      # noqa: BLE001
                ok, err = 0, repr(e)
            if world > 1:
                t_ok = torch.tensor([ok], dtype=torch.int32, device=xdev)
                dist.all_reduce(t_ok, op=dist.ReduceOp.MIN)
                all_ok = int(t_ok.item())
            else:
                all_ok = ok
            if not all_ok:
                if exchanger is not None:
                    exchanger.close()
                exchanger = None
                exchange_note = "library communicator unavailable on at least one rank (%s): exchange ran through torch.distributed" % (err or "another rank")
                sys.stderr.write("bench.py rank %d: %s\n" % (rank, exchange_note))
        if exchanger is None:
            exchanger = TorchExchange()

    def step(pending):
        """one pass over this rank's shard.  The exchange of a step's partial (an RCCL all-gather issued right after that
        step, off this thread) overlaps the following scans: its kernel only gets CUs when the scan it overlaps drains
        (the scan fills every CU), so it completes just AFTER that scan; results are therefore collected two steps late and
        nothing of the exchange is on the critical path except the last two, which are drained inside the timed region"""
        p = scfq.partial_device(shard_ptr, shard_n, prev_byte, flags=flags)
        t = scfq.last_timing()
        if not exchange:
            return scfq.finalize(p), t, pending
        done = None
        if len(pending) >= 2:
            pending.pop(0)
            done = timed_wait()
        exchanger.submit(p)
        pending.append(1)
        return done, t, pending

    xwait = {"ms": 0.0, "n": 0, "max_ms": 0.0}      # this rank's time inside exchanger.wait(): what the exchange costs the scanning thread

    def timed_wait():
        t0 = time.perf_counter()
        r = exchanger.wait()
        ms = (time.perf_counter() - t0) * 1e3
        xwait["ms"] += ms
        xwait["n"] += 1
        xwait["max_ms"] = max(xwait["max_ms"], ms)
        return r

    def drain(pending):
        out = None
        while pending:
            pending.pop(0)
            out = timed_wait()
        return out

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    pending = []
    for _ in range(args.warmup):
        c_done, _t, pending = step(pending)
        if c_done is not None:
            counts = c_done
    if pending:
        counts = drain(pending)
    barrier()
    xwait.update(ms=0.0, n=0, max_ms=0.0)      # (the warm-up's waits do not count)
    t_start = time.perf_counter()
    kern_ms = 0.0
    fold_ms = 0.0
    seen = set()
    for _ in range(args.steps):
        c_done, t, pending = step(pending)
        kern_ms += t.scan_kernel_ms
        fold_ms += t.fold_kernel_ms
        if c_done is not None:
            counts = c_done
            seen.add((counts.reads, counts.gc_bases, counts.n_bases, counts.bases, counts.lines))
    if pending:                   # the exchanges still in flight finish inside the timed region
        counts = drain(pending)
        seen.add((counts.reads, counts.gc_bases, counts.n_bases, counts.bases, counts.lines))
    barrier()
    elapsed = time.perf_counter() - t_start
    own_elapsed = elapsed
    if world > 1:
        te = torch.tensor([elapsed], dtype=torch.float64, device=xdev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    # every rank's own view of the timed region, gathered on the control plane: a curve that bends must say WHERE (a slow shard
    # generation is not in it, a slow kernel on one device or a rank waiting for the exchange is)
    mine = {"rank": rank, "device": local_rank, "elapsed_s": round(own_elapsed, 5), "avg_kernel_ms": round(kern_ms / max(1, args.steps), 4),
            "avg_fold_ms": round(fold_ms / max(1, args.steps), 4), "exchange_wait_ms_per_step": round(xwait["ms"] / max(1, args.steps), 4),
            "exchange_wait_ms_max": round(xwait["max_ms"], 3), "exchanges_waited_for": xwait["n"], "shard_bytes": shard_n, "generation_s": round(gen_s, 2),
            "exchange_path": ("none" if not exchange else "library" if isinstance(exchanger, LibExchange) else "torch mirror"),
            "transport": (exchanger.comm.transport if exchange and isinstance(exchanger, LibExchange) else (args.backend if exchange else "none"))}
    per_rank = [mine]
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    assert len(seen) == 1, ("steps disagree with each other", seen)   # every timed step produced the same counters
    # ---- correctness outside the timed region: generator tallies (independent of the scan) -------------
    # a rank owns the records that START inside its byte range; the record straddling its lower cut point
    # (generated here too, because the shard begins inside it) is tallied by the previous rank
    own = [info.gc_bases, info.n_bases, info.bases, info.records]
    if first_start < lo:
        _, head = scfq.synth_host(kind, seed, 1, first_record=first_rec)
        own = [own[0] - head.gc_bases, own[1] - head.n_bases, own[2] - head.bases, own[3] - 1]
    tallies = torch.tensor(own, dtype=torch.int64, device=xdev)
    if world > 1:
        dist.all_reduce(tallies)
    exact = (counts.gc_bases, counts.n_bases, counts.bases, counts.reads) == tuple(tallies.tolist())
    assert exact or args.no_verify, ("scan disagrees with generator tallies", counts.gc_bases, counts.n_bases, counts.bases,
                   counts.reads, tallies.tolist())

    total_bases = counts.bases   # bases of the WHOLE job (all ranks' shards folded)
    value = total_bases * args.steps / elapsed / 1e9
    ms_per_step = elapsed / args.steps * 1e3
    avg_kernel_ms = kern_ms / args.steps
    achieved = shard_n / (avg_kernel_ms * 1e-3) / 1e9 if avg_kernel_ms > 0 else 0.0

    out = {
        "metric": "Gbases/s parsed (fq-count)",
        "value": round(value, 3),
        "unit": "Gbases/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {
            "workload": (("synthetic %.0f GB uncompressed 150 bp Illumina FASTQ per GPU, HBM-resident (BASELINE configs[1]; the N>1 lines of this "
                          "bench hold 25 GB per GPU — `--bytes-per-gpu 25e9` at N=1 is their like-for-like anchor)" % (per / 1e9))
                         if (world == 1 and not anchor) else
                         ("synthetic %.0f GB uncompressed 150 bp Illumina FASTQ on 1 GPU, HBM-resident: the N=1 anchor of the configs[2] curve "
                          "(one shard of the seed-20260102 stream at the shard size of the N>1 lines; not BASELINE configs[1])" % (per / 1e9))
                         if world == 1 else
                         ("synthetic %.0f GB uncompressed 150 bp Illumina FASTQ byte-sharded across %d GPUs, %.0f GB per GPU, HBM-resident "
                          "(BASELINE configs[2]: 200 GB across 8 x MI355X + RCCL exchange%s)"
                          % (per * world / 1e9, world, per / 1e9, "" if (world == 8 and per == int(25e9)) else "; same shard size per GPU as its N=8 case"))) if kind == 0 else
                        ("synthetic %.0f GB Nanopore-style 500 bp-50 kb FASTQ per GPU, HBM-resident (BASELINE configs[4])" % (per / 1e9)),
            "bytes_per_gpu": shard_n, "seed": seed, "shards": "byte ranges at arbitrary (unaligned) cut points",
            "exchange": "none" if not exchange else exchanger.what,
            "flags": args.flags,
        },
        "roofline": {
            "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None,
            "kernel": "fq_scan_tiles", "avg_kernel_ms": round(avg_kernel_ms, 4), "avg_fold_ms": round(fold_ms / args.steps, 4),
            "algorithmic_bytes_per_launch": shard_n,
        },
        "counters": {"reads": counts.reads, "gc_bases": counts.gc_bases, "n_bases": counts.n_bases,
                     "bases": counts.bases, "tsv": scfq.format_tsv(counts), "matches_generator_tally": exact},
        "setup_s": round(gen_s, 2),
    }
    if exchange_note:
        out["config"]["exchange_note"] = exchange_note
    if exchange:
        out["config"]["per_rank"] = per_rank
    # `--exchange lib` is what the product ships; a rank that ended up on the Python mirror means the library's communicator did not come
    # up — the line is printed (its numbers are real), and the run then ends non-zero so that nobody reads it as the library's result
    mirror_fallback = bool(exchange and args.exchange == "lib" and any(r["exchange_path"] != "library" for r in per_rank))
    # the same device's read-stream ceiling: the scan kernel's load structure with no compute (diagnostic kernel)
    al = (-shard_ptr) % 4096
    if shard_n > al + (1 << 20):
        sm = scfq.debug_stream_ms(shard_ptr + al, shard_n - al, 5)
        if sm > 0:
            sbytes = (shard_n - al) // 4096 * 4096
            out["roofline"]["stream_ceiling"] = {"GBps": round(sbytes / (sm * 1e-3) / 1e9, 1), "ms": round(sm, 4),
                                                 "what": "same LDS-DMA ring and ranges, no classification/accounting, same buffer"}
            out["roofline"]["frac_of_stream_ceiling"] = round(achieved / (sbytes / (sm * 1e-3) / 1e9), 4)
    # PMC traffic measured offline with rocprofv3 (separate --pmc pass), if committed for this workload
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc) and args.flags == 0 and kind == 0:
        try:
            with open(pmc) as f:
                j = json.load(f)
            # the committed counters belong to ONE kernel instance at ONE range geometry: attach them only to a run of exactly that
            # (SCFQ_RING / SCFQ_NT / SCFQ_TILES_PER_RANGE change the load structure, and with it what may be re-read)
            ran = scfq.debug_last_scan_kernel()
            out["roofline"]["kernel_instance"] = ran
            same_kernel = ran.split(" tiles_per_range=")[0] == j.get("kernel") and ran.endswith("tiles_per_range=%s" % j.get("tiles_per_range"))
            if int(j.get("bytes_per_launch", -1)) == shard_n and same_kernel:
                out["roofline"]["traffic"] = j.get("hbm_bytes_per_launch")
                out["roofline"]["traffic_source"] = "profiles/pmc_traffic.json (offline rocprofv3 --pmc passes over this same workload and kernel: %s; not measured in this run)" % j.get("source", "")
        except Exception:
            pass

    # ---- CPU baseline: reference-shaped restatement on ONE host core, bounded sample of the same bytes ---
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import numpy as np
        L, OC = load_oracle()
        sample = int(min(args.cpu_sample_bytes, shard_n))
        host = buf[(lo - first_start):(lo - first_start) + sample].cpu().numpy()
        # cut the sample at a record boundary so that it is a well-formed file
        cut = sample
        nl = 0
        while cut > 0 and nl < 1:
            cut -= 1
            if host[cut] == 10:
                nl += 1
        oc = OC()
        tc = time.perf_counter()
        L.oracle_count_lines(host.ctypes.data, cut + 1, ctypes.byref(oc))
        cpu_s = time.perf_counter() - tc
        # the same sample through the HIP path must agree bit-exactly
        gc = scfq.count_device(shard_ptr, cut + 1)
        assert (gc.reads, gc.gc_bases, gc.n_bases, gc.bases, gc.lines) == (oc.reads, oc.gc_bases, oc.n_bases, oc.bases, oc.lines), \
            "HIP path and CPU oracle disagree on the sample"
        out["cpu_baseline"] = {
            "value": round(oc.bases / cpu_s / 1e9, 4), "unit": "Gbases/s", "cores": 1, "kind": "port",
            "sample": "first %.2f GB of the same workload, page-resident host memory, oracle_count_lines "
                      "(line loop + three memchr count passes, src/fq_count.nim:38-45), %.1f s; host has %d cores"
                      % ((cut + 1) / 1e9, cpu_s, os.cpu_count()),
            "bytes_per_s_GB": round((cut + 1) / cpu_s / 1e9, 3),
            "matches_hip_path": True,
        }
        threads = min(os.cpu_count() or 1, 64)
        all_s, acc = cpu_all_cores(L, host, cut + 1, threads)
        assert (acc[2], acc[6], acc[10]) == (oc.gc_bases, oc.n_bases, oc.bases)
        out["cpu_all_cores"] = {"value": round(oc.bases / all_s / 1e9, 3), "unit": "Gbases/s", "threads": threads,
                                "note": "optimised CPU restatement: byte-range shards + the same ordered fold, byte-serial scan per shard"}
        del host
    side_leg_failed = False
    if rank == 0 and world == 1 and args.ingest_bytes > 0 and kind == 0 and args.flags == 0:
        del buf          # (the 10 GB workload: the ingest legs measure processes of their own, beside a parent that holds little)
        torch.cuda.empty_cache()
        # (non-headline: a failure here — no room for the files, a host short of memory — is reported in the object, never allowed to take
        # the headline line with it)
        # Only RESOURCE trouble is forgiven; the legs' assertions are correctness checks (counters == generator tallies, dedup idempotent):
        # a wrong result is recorded in the object too, the line is still printed, and the run then ends non-zero.
        forgiven = (OSError, MemoryError, subprocess.SubprocessError, torch.OutOfMemoryError)
        try:
            out["ingest"] = ingest_rows(scfq, int(args.ingest_bytes), int(min(args.ingest_bgzf_bytes, args.ingest_bytes)))
        except forgiven as e:
            import traceback
            out["ingest"] = {"error": "%s: %s" % (type(e).__name__, e), "traceback_tail": traceback.format_exc()[-1500:]}
        except Exception as e:      # noqa: BLE001
            import traceback
            out["ingest"] = {"error": "%s: %s" % (type(e).__name__, e), "traceback_tail": traceback.format_exc()[-1500:], "fatal": True}
            side_leg_failed = True
        # (behind the ingest legs: its pool keeps the scratch of a 10 GB call, which the legs' processes need not find in their way)
        torch.cuda.empty_cache()
        try:
            out["dedup"] = dedup_row(scfq, torch)
        except forgiven as e:
            out["dedup"] = {"error": "%s: %s" % (type(e).__name__, e)}
        except Exception as e:      # noqa: BLE001
            out["dedup"] = {"error": "%s: %s" % (type(e).__name__, e), "fatal": True}
            side_leg_failed = True
    if rank == 0:
        print(json.dumps(out), flush=True)
    if exchange:
        exchanger.close()
        dist.destroy_process_group()
    if mirror_fallback:
        sys.stderr.write("bench.py: --exchange lib was asked for and at least one rank ran the torch.distributed mirror (%s): exit 1\n" % (exchange_note or "see config.per_rank"))
        sys.exit(1)
    if side_leg_failed:
        sys.stderr.write("bench.py: a non-headline leg produced a WRONG RESULT or an unexpected error (see the line's \"fatal\" object): exit 1\n")
        sys.exit(1)


if __name__ == "__main__":
    try:
        main()
    except BaseException as e:      # noqa: BLE001
        if isinstance(e, SystemExit) and not e.code:
            raise
        # a failed or stuck collective must END the rank (non-zero), never leave it waiting in an atexit hook or a join
        import traceback
        traceback.print_exc()
        sys.stderr.flush()
        sys.stdout.flush()
        os._exit(1)
