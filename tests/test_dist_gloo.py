"""N>1 path on CPU: world_size-2 (and 3) gloo processes, byte-range shards at arbitrary cut points, the
all_gather + rank-ordered fold of scfq_dist.exchange_partials.  Shard partials come from the oracle here (no
GPU in this container); the exchange, the ordered combine and the finalisation are the product's code."""
import ctypes
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import PKG, ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, data_bytes, want_hist, q):
    sys.path.insert(0, os.path.join(PKG, "pyhost"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import scfq
    import scfq_dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    data = np.frombuffer(data_bytes, dtype=np.uint8)
    lo, hi = scfq_dist.shard_bounds(data.size, world, rank)
    O = ctypes.CDLL(os.path.join(ROOT, "oracle", "libfqcount_oracle.so"))
    O.oracle_partial.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64), ctypes.c_void_p]
    w = (ctypes.c_uint64 * 27)()
    h = (ctypes.c_uint64 * 1024)()
    shard = np.ascontiguousarray(data[lo:hi])
    O.oracle_partial(shard.ctypes.data if shard.size else None, shard.size, int(data[lo - 1]) if lo else -1, w, ctypes.byref(h))
    p = scfq.Partial.from_words(list(w) + [0] * 5)
    acc, acc_h = scfq_dist.exchange_partials(p, hist=h if want_hist else None)
    c = scfq.finalize(acc, acc_h)
    # the pipelined form (start now, finish later) must give the same fold
    pend = scfq_dist.start_exchange(p, hist=h if want_hist else None)
    acc2, acc2_h = scfq_dist.finish_exchange(pend)
    assert acc2.words() == acc.words()
    assert (acc2_h is None) == (acc_h is None) and (acc_h is None or list(acc2_h) == list(acc_h))
    q.put((rank, c.reads, c.gc_bases, c.n_bases, c.bases, c.lines, c.bad_at, c.bad_plus, list(c.qual_hist), scfq.format_tsv(c)))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,want_hist", [(2, False), (2, True), (3, True)])
def test_sharded_exchange_matches_oracle(oracle, world, want_hist):
    rng = np.random.default_rng(world * 7 + want_hist)
    rec = b"@r x\r\nACGTNNGCGC\r\n+\r\nIIII#III@+\r\n@r2\nGGCCN\n+r2\n!!!!!\n"
    data = rec * 37 + bytes(rng.choice(np.frombuffer(b"ACGTN\n\r@+", dtype=np.uint8), 1001))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, data, want_hist, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    oc = oracle.count(np.frombuffer(data, dtype=np.uint8), "bytes")
    for res in results:
        rank, reads, gc, n, bases, lines, bad_at, bad_plus, hist, tsv = res
        assert (reads, gc, n, bases, lines, bad_at, bad_plus) == (oc.reads, oc.gc_bases, oc.n_bases, oc.bases, oc.lines, oc.bad_at, oc.bad_plus), rank
        assert tsv == oracle.tsv(oc)
        if want_hist:
            assert hist == list(oc.qual_hist)


def test_shard_bounds_cover_without_overlap():
    sys.path.insert(0, os.path.join(PKG, "pyhost"))
    import scfq_dist
    for total in (0, 1, 7, 1000, 10**10 + 3):
        for world in (1, 2, 3, 8):
            edges = [scfq_dist.shard_bounds(total, world, r) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(edges[:-1], edges[1:]))
