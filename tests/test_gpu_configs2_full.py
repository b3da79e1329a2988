"""BASELINE configs[2] at its FULL size on one MI355X: the 200 GB stream of seed 20260102 (SURVEY.md §8d input 3) resident in the
288 GB of one device, cut into the eight byte-range shards the 8-GPU run gives its ranks — `[r·S/8, (r+1)·S/8)`, arbitrary
(unaligned) cut points, none of them on a record boundary — every shard scanned by the real kernels with its look-behind byte in
memory, every shard's 32-word partial sent through ONE RCCL communicator (world 1: dlopen, ncclCommInitRank, pinned staging,
ncclAllGather of ncclUint64, fold — the payload path of the N>1 run), and the eight partials folded in rank order with the shard
monoid.  All ≈556 M records and all seven interior cuts against the generator's own tallies.

What this does NOT show: RCCL with more than one rank (hardware: there is one GPU per box here).  The reference seam is one call per
file (`/root/reference/sc.nim:114-116`, `src/fq_count.nim:14-53`): the eight shards together are that one call."""
import time

import pytest

pytestmark = pytest.mark.gpu

SEED_CONFIGS2 = 20260102
TOTAL = int(200e9)
WORLD = 8


def test_configs2_200gb_eight_shards_one_device(gpu, scfq):
    torch = gpu
    free, total = torch.cuda.mem_get_info()
    assert free > TOTAL + (8 << 30), "this test needs the 288 GB device (free: %.1f GB)" % (free / 1e9)
    t0 = time.time()
    plan = scfq.synth_plan(0, SEED_CONFIGS2, TOTAL)
    S = plan.bytes
    assert TOTAL <= S < TOTAL + 400 and plan.records > 550_000_000
    buf = torch.empty(S + 8192, dtype=torch.uint8, device="cuda:0")
    base = buf.data_ptr()
    # the generator in launches of at most 8 M records (2.9 GB): every launch reports what it wrote, the next one starts behind it
    tallies = [0, 0, 0, 0]           # gc, n, bases, records
    at, rec = 0, 0
    while rec < plan.records:
        n = min(8_000_000, plan.records - rec)
        info = scfq.synth_device(0, SEED_CONFIGS2, n, base + at, S + 8192 - at, first_record=rec)
        assert info.records == n
        tallies = [tallies[0] + info.gc_bases, tallies[1] + info.n_bases, tallies[2] + info.bases, tallies[3] + n]
        at += info.bytes
        rec += n
    assert at == S
    torch.cuda.synchronize()
    t_gen = time.time() - t0
    # a cut that fell on a record boundary would not test the monoid: none of the seven does (checked, not assumed)
    cuts = [r * S // WORLD for r in range(WORLD + 1)]
    for c in cuts[1:-1]:
        _, start = scfq.synth_locate(0, SEED_CONFIGS2, c)
        assert start != c

    comm = scfq.Comm.init_rank(scfq.Comm.unique_id(), 1, 0, 0, timeout_ms=120000)
    assert comm.transport.startswith("RCCL 2.")
    try:
        t1 = time.time()
        acc = scfq.identity()
        acc_direct = scfq.identity()
        scan_ms = 0.0
        for r in range(WORLD):
            lo, hi = cuts[r], cuts[r + 1]
            flags = scfq.SCFQ_TIMING | (scfq.SCFQ_PREV_IN_MEMORY if r else 0)
            p = scfq.partial_device(base + lo, hi - lo, prev_byte=-1, flags=flags)
            scan_ms += scfq.last_timing().scan_kernel_ms
            assert p.bytes == hi - lo
            # the 32-word payload through the communicator, exactly as a rank of the N>1 run sends it
            q = comm.exchange(p, timeout_ms=60000)
            assert q.words() == p.words()
            scfq.combine(acc, q)
            scfq.combine(acc_direct, p)
        t_scan = time.time() - t1
    finally:
        comm.destroy()
    c = scfq.finalize(acc)
    assert (c.gc_bases, c.n_bases, c.bases, c.reads) == tuple(tallies), (c.gc_bases, c.n_bases, c.bases, c.reads, tallies)
    assert c.reads == plan.records and c.input_bytes == S and c.lines == 4 * plan.records
    assert scfq.finalize(acc_direct).bases == c.bases
    # one launch over the whole 200 GB is the same count (a single scan launch covers up to 16 TiB)
    whole = scfq.count_device(base, S, flags=scfq.SCFQ_TIMING)
    whole_ms = scfq.last_timing().scan_kernel_ms
    assert (whole.gc_bases, whole.n_bases, whole.bases, whole.reads, whole.lines) == (c.gc_bases, c.n_bases, c.bases, c.reads, c.lines)
    # structure check and quality histogram folded over the same eight shards: every header starts with '@', every separator with '+',
    # and the histogram holds exactly one quality byte per base
    import ctypes
    acc2, h2 = scfq.Partial(), (ctypes.c_uint64 * scfq.HIST_WORDS)()
    scfq.lib().scfq_partial_identity(ctypes.byref(acc2), ctypes.byref(h2))
    for r in range(WORLD):
        lo, hi = cuts[r], cuts[r + 1]
        flags = scfq.SCFQ_QUAL_HIST | scfq.SCFQ_STRUCT_CHECK | (scfq.SCFQ_PREV_IN_MEMORY if r else 0)
        p, h = scfq.partial_device(base + lo, hi - lo, prev_byte=-1, flags=flags, want_hist=True)
        scfq.combine(acc2, p, h2, h)
    c2 = scfq.finalize(acc2, h2)
    assert (c2.gc_bases, c2.n_bases, c2.bases, c2.reads) == tuple(tallies)
    assert c2.bad_at == 0 and c2.bad_plus == 0
    assert sum(c2.qual_hist) == c2.bases and {v for v in range(256) if c2.qual_hist[v]} == {ord(ch) for ch in "F:,#"}
    print("configs[2] at 200 GB on one device: %d records, generated in %.1f s; eight shards scanned + exchanged + folded in %.2f s "
          "(scan kernels %.1f ms = %.2f TB/s); one launch over all of it %.1f ms = %.2f TB/s"
          % (c.reads, t_gen, t_scan, scan_ms, S / scan_ms / 1e9, whole_ms, S / whole_ms / 1e9))
