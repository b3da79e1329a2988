"""C1 inside the C library (scfq_comm_*, csrc/scfq_comm.cpp) on the CPU: one PROCESS per rank, world 2 and 3, the library's
own rendezvous and its SCFQ_COMM_TCP transport (host sockets: no device needed).  Shard partials come from the oracle (no GPU
in this container); the rendezvous, the all-gather, the rank-ordered fold, the two-halves (start / finish) form, the
allgather_u64 helper and the deadlines are the product's code.  The RCCL transport of the same object runs in
tests/test_gpu_comm.py."""
import ctypes
import multiprocessing as mp
import os
import socket
import sys
import time

import numpy as np
import pytest

from conftest import PKG, ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_partial(data, lo, hi, want_hist):
    O = ctypes.CDLL(os.path.join(ROOT, "oracle", "libfqcount_oracle.so"))
    O.oracle_partial.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64), ctypes.c_void_p]
    w = (ctypes.c_uint64 * 27)()
    h = (ctypes.c_uint64 * 1024)()
    shard = np.ascontiguousarray(data[lo:hi])
    O.oracle_partial(shard.ctypes.data if shard.size else None, shard.size, int(data[lo - 1]) if lo else -1, w, ctypes.byref(h))
    return list(w) + [0] * 5, (h if want_hist else None)


def _worker(rank, world, port, data_bytes, want_hist, q):
    sys.path.insert(0, os.path.join(PKG, "pyhost"))
    import scfq
    try:
        data = np.frombuffer(data_bytes, dtype=np.uint8)
        comm = scfq.Comm.init_rendezvous("127.0.0.1", port, world, rank, transport=scfq.SCFQ_COMM_TCP, timeout_ms=60000)
        assert (comm.world, comm.rank, comm.transport) == (world, rank, "tcp")
        lo, hi = data.size * rank // world, data.size * (rank + 1) // world
        words, h = _oracle_partial(data, lo, hi, want_hist)
        p = scfq.Partial.from_words(words)
        if want_hist:
            acc, acc_h = comm.exchange(p, h, timeout_ms=60000)
        else:
            acc, acc_h = comm.exchange(p, timeout_ms=60000), None
        c = scfq.finalize(acc, acc_h)
        # the pipelined form: two exchanges in flight, finishes answer starts in order
        p2 = scfq.Partial.from_words(words)
        p2.bytes += 1000 * (rank + 1)
        comm.start(p, h, timeout_ms=60000)
        comm.start(p2, h, timeout_ms=60000)
        a1 = comm.finish(want_hist, timeout_ms=60000)
        a2 = comm.finish(want_hist, timeout_ms=60000)
        a1p, a2p = (a1[0], a2[0]) if want_hist else (a1, a2)
        assert a1p.words() == acc.words()
        assert a2p.bytes == acc.bytes + 1000 * world * (world + 1) // 2
        rows = comm.allgather_u64([rank, 7 * rank + 1, 2**64 - 1 - rank], timeout_ms=60000)
        assert rows == [[r, 7 * r + 1, 2**64 - 1 - r] for r in range(world)]
        comm.destroy()
        q.put((rank, "ok", c.reads, c.gc_bases, c.n_bases, c.bases, c.lines, c.bad_at, c.bad_plus, list(c.qual_hist), scfq.format_tsv(c)))
    except BaseException as e:      # noqa: BLE001 — the parent asserts on the text
        q.put((rank, "error: %r" % (e,)))


@pytest.mark.parametrize("world,want_hist", [(2, False), (2, True), (3, True)])
def test_tcp_transport_exchange_matches_oracle(oracle, world, want_hist):
    rng = np.random.default_rng(world * 11 + want_hist)
    rec = b"@r x\r\nACGTNNGCGC\r\n+\r\nIIII#III@+\r\n@r2\nGGCCN\n+r2\n!!!!!\n"
    data = rec * 41 + bytes(rng.choice(np.frombuffer(b"ACGTN\n\r@+", dtype=np.uint8), 997))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, data, want_hist, q)) for r in range(world)]
    for p in procs[1:]:          # rank 0 (the listener) last: the others must retry until it is up
        p.start()
    time.sleep(0.3)
    procs[0].start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    oc = oracle.count(np.frombuffer(data, dtype=np.uint8), "bytes")
    for res in results:
        assert res[1] == "ok", res
        rank, _, reads, gc, n, bases, lines, bad_at, bad_plus, hist, tsv = res
        assert (reads, gc, n, bases, lines, bad_at, bad_plus) == (oc.reads, oc.gc_bases, oc.n_bases, oc.bases, oc.lines, oc.bad_at, oc.bad_plus), rank
        assert tsv == oracle.tsv(oc)
        if want_hist:
            assert hist == list(oc.qual_hist)


def test_missing_rank_is_an_error_not_a_hang(scfq):
    """rank 1 of 2 never shows up: rank 0's rendezvous must come back with SCFQ_ERCCL at the deadline; a rank that finds
    nobody listening likewise"""
    t0 = time.time()
    with pytest.raises(scfq.ScfqError) as e:
        scfq.Comm.init_rendezvous("127.0.0.1", _free_port(), 2, 0, transport=scfq.SCFQ_COMM_TCP, timeout_ms=700)
    assert e.value.rc == scfq.SCFQ_ERCCL and "1 of 2 ranks" in str(e.value)
    with pytest.raises(scfq.ScfqError) as e:
        scfq.Comm.init_rendezvous("127.0.0.1", _free_port(), 2, 1, transport=scfq.SCFQ_COMM_TCP, timeout_ms=700)
    assert e.value.rc == scfq.SCFQ_ERCCL and "did not answer" in str(e.value)
    assert time.time() - t0 < 20


def _token_worker(rank, world, port, token, timeout_ms, q):
    sys.path.insert(0, os.path.join(PKG, "pyhost"))
    if token:
        os.environ["SCFQ_RENDEZVOUS_TOKEN"] = token
    import scfq
    t0 = time.time()
    try:
        comm = scfq.Comm.init_rendezvous("127.0.0.1", port, world, rank, transport=scfq.SCFQ_COMM_TCP, timeout_ms=timeout_ms)
        q.put((rank, "up", time.time() - t0))
        comm.destroy()
    except scfq.ScfqError as e:
        q.put((rank, "rc=%d %s" % (e.rc, e), time.time() - t0))


def test_a_rank_that_is_turned_away_learns_it_at_once(scfq):
    """rank 0 answers every hello with a verdict word: a rank with another launch's token, and a rank number outside rank 0's world, fail
    FAST with the reason (not at the deadline, and not later with an unrelated message of the id exchange), while rank 0 keeps
    waiting for its real rank — which then arrives and is admitted"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    r0 = ctx.Process(target=_token_worker, args=(0, 2, port, "launch-A", 20000, q))
    r0.start()
    time.sleep(0.5)
    wrong = ctx.Process(target=_token_worker, args=(1, 2, port, "launch-B", 20000, q))
    wrong.start()
    rank, what, dt = q.get(timeout=30)
    assert rank == 1 and "rc=%d" % scfq.SCFQ_ERCCL in what and "SCFQ_RENDEZVOUS_TOKEN differs" in what and dt < 5, (what, dt)
    stray = ctx.Process(target=_token_worker, args=(2, 3, port, "launch-A", 20000, q))      # (a rank of a launch with another --shard-world)
    stray.start()
    rank, what, dt = q.get(timeout=30)
    assert rank == 2 and "no such rank" in what and dt < 5, (what, dt)
    good = ctx.Process(target=_token_worker, args=(1, 2, port, "launch-A", 20000, q))
    good.start()
    got = sorted(q.get(timeout=30)[:2] for _ in range(2))
    assert got == [(0, "up"), (1, "up")], got
    for p in (r0, wrong, stray, good):
        p.join(timeout=30)


def _half_worker(rank, port, q):
    sys.path.insert(0, os.path.join(PKG, "pyhost"))
    import scfq
    comm = scfq.Comm.init_rendezvous("127.0.0.1", port, 2, rank, transport=scfq.SCFQ_COMM_TCP, timeout_ms=30000)
    if rank == 1:
        q.put("rank 1 up")
        time.sleep(4)           # never takes part in the exchange
        return
    try:
        comm.exchange(scfq.identity(), timeout_ms=800)
        q.put("no error")
    except scfq.ScfqError as e:
        q.put("rc=%d %s" % (e.rc, e))
    # a communicator that failed once keeps failing quickly instead of waiting again
    t0 = time.time()
    try:
        comm.exchange(scfq.identity(), timeout_ms=800)
        q.put("no error")
    except scfq.ScfqError as e:
        q.put("again rc=%d in %.1fs" % (e.rc, time.time() - t0))
    comm.destroy()


def test_stuck_exchange_times_out(scfq):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_half_worker, args=(r, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    msgs = [q.get(timeout=60) for _ in range(3)]
    for p in procs:
        p.join(timeout=60)
    first = [m for m in msgs if m.startswith("rc=")]
    assert first and first[0].startswith("rc=%d" % scfq.SCFQ_ERCCL), msgs
    again = [m for m in msgs if m.startswith("again")]
    assert again and again[0].startswith("again rc=%d" % scfq.SCFQ_ERCCL), msgs


def _late_answer_worker(q):
    os.environ["SCFQ_COMM_TEST_DELAY_MS"] = "1200"      # the worker "hangs" for 1.2 s in every gather
    os.environ["SCFQ_COMM_TAKE_SLACK_MS"] = "50"        # and the caller gives up 50 ms after the exchange's own deadline
    sys.path.insert(0, os.path.join(PKG, "pyhost"))
    import scfq
    comm = scfq.Comm.init_rendezvous(None, 5000, 1, 0, transport=scfq.SCFQ_COMM_TCP, timeout_ms=5000)
    first = scfq.identity()
    first.nl, first.bytes = 3, 1111
    try:
        comm.exchange(first, timeout_ms=100)
        q.put("first: no error")
    except scfq.ScfqError as e:
        q.put("first rc=%d" % e.rc)
    q.put("broken=%d" % comm.broken)
    time.sleep(1.6)             # the late answer to the FIRST exchange is in the queue by now
    second = scfq.identity()
    second.nl, second.bytes = 5, 2222
    t0 = time.time()
    try:
        got = comm.exchange(second, timeout_ms=100)
        q.put("second: answered with bytes=%d" % got.bytes)      # 1111 here would be the previous exchange's rows
    except scfq.ScfqError as e:
        q.put("second rc=%d in %.2fs" % (e.rc, time.time() - t0))
    try:
        got = comm.finish(timeout_ms=100)
        q.put("stray finish: answered with bytes=%d" % got.bytes)
    except scfq.ScfqError as e:
        q.put("stray finish rc=%d" % e.rc)
    comm.destroy()


def test_late_answer_is_never_handed_to_the_next_exchange(scfq):
    """An exchange whose caller gave up at its deadline leaves an answer behind when the worker completes after all: that
    answer must be dropped — the next start / finish pair may fail, it may never return the previous exchange's fold with rc 0."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_late_answer_worker, args=(q,))
    p.start()
    msgs = [q.get(timeout=60) for _ in range(4)]
    p.join(timeout=60)
    assert msgs[0] == "first rc=%d" % scfq.SCFQ_ERCCL, msgs
    assert msgs[1] == "broken=1", msgs
    assert msgs[2].startswith("second rc=%d" % scfq.SCFQ_ERCCL) and float(msgs[2].split(" in ")[1][:-1]) < 0.5, msgs      # fails fast
    assert msgs[3].startswith("stray finish rc="), msgs


def test_argument_checks(scfq):
    L = scfq.lib()
    h = ctypes.c_void_p()
    assert L.scfq_comm_init_rendezvous(None, 0, 2, 0, 0, scfq.SCFQ_COMM_TCP, 100, ctypes.byref(h)) == scfq.SCFQ_EARG      # port
    assert L.scfq_comm_init_rendezvous(None, 5000, 2, 2, 0, scfq.SCFQ_COMM_TCP, 100, ctypes.byref(h)) == scfq.SCFQ_EARG   # rank >= world
    assert L.scfq_comm_init_rendezvous(None, 5000, 2, 0, 0, 7, 100, ctypes.byref(h)) == scfq.SCFQ_EARG                    # transport
    assert L.scfq_comm_exchange(None, None, None, None, None, 0) == scfq.SCFQ_EARG
    # world 1 over tcp needs no socket at all
    c = scfq.Comm.init_rendezvous(None, 5000, 1, 0, transport=scfq.SCFQ_COMM_TCP, timeout_ms=1000)
    p = scfq.identity()
    p.nl, p.bytes = 5, 77
    assert c.exchange(p).words() == p.words()
    c.destroy()
