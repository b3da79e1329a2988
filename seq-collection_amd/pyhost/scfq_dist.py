"""C1 — the cross-rank exchange of shard partials (one process per GPU, torch.distributed; backend "nccl" is
RCCL over xGMI on ROCm, "gloo" in the CPU tests).

The reference is single-process and has no collective; the exchange exists because the input
is byte-range sharded across GPUs.  The shard combine is ORDERED and non-commutative (the class rotation by the
left operand's newline count, include/sc_fqcount.h), so a sum-allreduce of final counters would be wrong:
every rank all_gathers the 32-word partials and performs the same rank-ordered fold.
Payload: world x 256 B (+ world x 8 KiB with the quality histogram) — latency-bound.
"""
import torch
import torch.distributed as dist

import scfq

_MASK = (1 << 64) - 1


def _to_i64(words):
    return [w - (1 << 64) if w >= (1 << 63) else w for w in words]


def _from_i64(vals):
    return [int(v) & _MASK for v in vals]


def shard_bounds(total_bytes, world, rank):
    """Byte range of `rank`: contiguous, arbitrary (unaligned) cut points, SURVEY.md §8(e)."""
    return total_bytes * rank // world, total_bytes * (rank + 1) // world


class PendingExchange:
    """An all_gather in flight (start_exchange): lets a host overlap the exchange of shard k with the scan of shard k+1."""
    def __init__(self, work, out, world, width, with_hist):
        self.work, self.out, self.world, self.width, self.with_hist = work, out, world, width, with_hist


def start_exchange(partial, device=None, group=None, hist=None):
    world = dist.get_world_size(group)
    words = _to_i64(partial.words())
    if hist is not None:
        words = words + _to_i64(list(hist))
    mine = torch.tensor(words, dtype=torch.int64, device=device)
    out = torch.empty(world * mine.numel(), dtype=torch.int64, device=device)
    work = dist.all_gather_into_tensor(out, mine, group=group, async_op=True)
    return PendingExchange(work, out, world, mine.numel(), hist is not None)


def finish_exchange(pending):
    """Wait for the all_gather, fold in rank order. Returns (Partial, hist or None), identical on every rank."""
    import ctypes
    pending.work.wait()
    rows = pending.out.view(pending.world, pending.width).cpu().tolist()
    acc = scfq.identity()
    acc_h = (ctypes.c_uint64 * scfq.HIST_WORDS)() if pending.with_hist else None
    for r in range(pending.world):
        p = scfq.Partial.from_words(_from_i64(rows[r][:scfq.PARTIAL_WORDS]))
        h = (ctypes.c_uint64 * scfq.HIST_WORDS)(*_from_i64(rows[r][scfq.PARTIAL_WORDS:])) if pending.with_hist else None
        scfq.combine(acc, p, acc_h, h)
    return acc, acc_h


def exchange_partials(partial, device=None, group=None, hist=None):
    """all_gather every rank's partial (and optional [4][256] histogram), fold in rank order.

    Returns (folded Partial, folded hist or None); identical on every rank."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return partial, hist
    words = _to_i64(partial.words())
    if hist is not None:
        words = words + _to_i64(list(hist))        # one message per rank: partial (+ histogram)
    mine = torch.tensor(words, dtype=torch.int64, device=device)
    out = torch.empty(world * mine.numel(), dtype=torch.int64, device=device)
    try:
        dist.all_gather_into_tensor(out, mine, group=group)      # one collective, one contiguous result
    except (RuntimeError, NotImplementedError):
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine, group=group)
        out = torch.cat(parts)
    rows = out.view(world, mine.numel()).cpu().tolist()           # a single device->host copy
    acc = scfq.identity()
    import ctypes
    acc_h = (ctypes.c_uint64 * scfq.HIST_WORDS)() if hist is not None else None
    for r in range(world):
        p = scfq.Partial.from_words(_from_i64(rows[r][:scfq.PARTIAL_WORDS]))
        h = (ctypes.c_uint64 * scfq.HIST_WORDS)(*_from_i64(rows[r][scfq.PARTIAL_WORDS:])) if hist is not None else None
        scfq.combine(acc, p, acc_h, h)
    return acc, acc_h


class DeviceExchange:
    """The same exchange for a host that keeps scanning while it runs (bench.py's helper thread): every HIP call it makes is
    asynchronous (pinned staging, a side stream, event polling with short sleeps), because a blocking call issued from a
    second thread (torch.tensor(list, device=...), .cpu()) was measured to delay the scanning thread's own launches by
    ~0.1 ms per step through the runtime's locks."""

    def __init__(self, device, group=None, width=scfq.PARTIAL_WORDS):
        self.group = group
        self.world = dist.get_world_size(group)
        self.width = width
        self.stream = torch.cuda.Stream(device=device)
        self.h_in = torch.empty(width, dtype=torch.int64).pin_memory()
        self.h_out = torch.empty(self.world * width, dtype=torch.int64).pin_memory()
        self.d_in = torch.empty(width, dtype=torch.int64, device=device)
        self.d_out = torch.empty(self.world * width, dtype=torch.int64, device=device)
        self.done = torch.cuda.Event()

    def start(self, partial):
        self.h_in.copy_(torch.tensor(_to_i64(partial.words()), dtype=torch.int64))
        with torch.cuda.stream(self.stream):
            self.d_in.copy_(self.h_in, non_blocking=True)
            dist.all_gather_into_tensor(self.d_out, self.d_in, group=self.group)     # enqueued behind the copy on this stream
            self.h_out.copy_(self.d_out, non_blocking=True)
            self.done.record(self.stream)

    def finish(self, poll_s=5e-5):
        import time
        while not self.done.query():
            time.sleep(poll_s)
        rows = self.h_out.view(self.world, self.width).tolist()
        acc = scfq.identity()
        for r in range(self.world):
            scfq.combine(acc, scfq.Partial.from_words(_from_i64(rows[r])))
        return acc
