"""C1 on the GPU box: the C library's communicator over RCCL (single rank — one device is all a test box has — which
still runs every step: dlopen of librccl, ncclCommInitRank / ncclCommInitAll, pinned staging, ncclAllGather on the private
stream, the fold), the sharded file entry, the CLI's one-process-per-GPU mode with real kernels on both ranks (TCP
transport: RCCL refuses two ranks on one device), and bench.py's N>1 code paths exactly as the driver launches them."""
import ctypes
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import PKG, ROOT
from test_ingest_sources import fastq_bytes

pytestmark = pytest.mark.gpu

REF_FIELDS = ("reads", "gc_bases", "n_bases", "bases", "lines")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _same(c, oc, fields=REF_FIELDS):
    for f in fields:
        assert getattr(c, f) == getattr(oc, f), (f, getattr(c, f), getattr(oc, f))


def test_rccl_single_rank_exchange(gpu, scfq, oracle):
    data = np.frombuffer(fastq_bytes(300_000, seed=3), dtype=np.uint8)
    words, hist = oracle.partial(data, -1, want_hist=True)
    p = scfq.Partial.from_words(words + [0] * 5)
    h = (ctypes.c_uint64 * scfq.HIST_WORDS)(*hist)
    for comm in (scfq.Comm.init_all([0])[0], scfq.Comm.init_rank(scfq.Comm.unique_id(), 1, 0, 0, timeout_ms=120000)):
        assert comm.world == 1 and comm.rank == 0 and comm.transport.startswith("RCCL 2.")
        assert comm.exchange(p, timeout_ms=60000).words() == p.words()
        acc, acc_h = comm.exchange(p, h, timeout_ms=60000)
        assert acc.words() == p.words() and list(acc_h) == hist
        for k in range(3):      # several in flight
            comm.start(p, timeout_ms=60000)
        assert all(comm.finish(timeout_ms=60000).words() == p.words() for _ in range(3))
        assert comm.allgather_u64([1, 2, 3]) == [[1, 2, 3]]
        c = scfq.finalize(acc, acc_h)
        _same(c, oracle.count(data))
        comm.destroy()


def test_sharded_file_entry_single_rank(gpu, scfq, oracle, tmp_path):
    import gzip
    data = fastq_bytes(5_000_000, seed=8)
    plain = tmp_path / "a.fq"
    plain.write_bytes(data)
    gz = tmp_path / "a.fq.gz"
    gz.write_bytes(gzip.compress(data, 6))
    oc = oracle.count(np.frombuffer(data, dtype=np.uint8))
    comm = scfq.Comm.init_all([0])[0]
    for path in (plain, gz):
        c = comm.count_file(str(path), flags=scfq.SCFQ_QUAL_HIST | scfq.SCFQ_STRUCT_CHECK, chunk_bytes=1 << 20)
        _same(c, oc, REF_FIELDS + ("bad_at", "bad_plus"))
        assert list(c.qual_hist) == list(oc.qual_hist)
    with pytest.raises(scfq.ScfqError) as e:
        comm.count_file(str(tmp_path / "missing.fq"))
    assert e.value.rc == scfq.SCFQ_EOPEN
    comm.destroy()


def test_multi_device_list_goes_through_rccl(gpu, scfq, oracle, tmp_path):
    """`sc fq-count --devices=0 big.fq` with SCFQ_EXCHANGE_AT_1=1: shard -> K1/K2 -> ncclCommInitAll + ncclAllGather -> fold,
    no Python in the process"""
    data = fastq_bytes(6_000_000, seed=13)
    f = tmp_path / "big.fq"
    f.write_bytes(data)
    oc = oracle.count(np.frombuffer(data, dtype=np.uint8))
    env = dict(os.environ, SCFQ_EXCHANGE_AT_1="1", SCFQ_VERBOSE="1")
    r = subprocess.run([os.path.join(PKG, "sc"), "fq-count", "--devices=0", str(f)], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr
    assert r.stdout == oracle.tsv(oc) + "\n"


def test_cli_one_process_per_rank_real_kernels(gpu, scfq, oracle, tmp_path):
    """two `sc fq-count` processes, ranks 0 and 1 of 2, both on device 0: each scans its byte range (cut at an arbitrary
    offset, CRLF input so the look-behind byte matters) with the HIP kernels, the partials cross the library's TCP transport,
    rank 0 prints the row"""
    rec = b"@r x\r\nACGTNNGCGC\r\n+\r\nIIII#III@+\r\n@r2\nGGCCN\n+r2\n!!!!!\n"
    data = rec * 60_001 + b"@tail\nACGT"
    f = tmp_path / "crlf.fq"
    f.write_bytes(data)
    oc = oracle.count(np.frombuffer(data, dtype=np.uint8))
    port = _free_port()
    sc = os.path.join(PKG, "sc")
    procs = [subprocess.Popen([sc, "fq-count", "--shard-rank=%d" % r, "--shard-world=2", "--rendezvous=127.0.0.1:%d" % port,
                               "--transport=tcp", "--devices=0", "-b", str(f)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in (1, 0)]
    outs = [p.communicate(timeout=300) for p in procs]
    assert [p.returncode for p in procs] == [0, 0], outs
    assert outs[0][0] == ""                                             # rank 1 prints nothing
    assert outs[1][0] == oracle.tsv(oc) + "\tcrlf.fq\n"                 # rank 0: the reference's row


def _bench(args, nproc=0, timeout=600):
    env = dict(os.environ)
    if nproc:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py")] + args
    else:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
        env["MASTER_PORT"] = str(_free_port())
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_exchange_path_single_rank_rccl(gpu):
    """bench.py's N>1 code (library communicator, exchanges two steps in flight, drain) with one RCCL rank"""
    j = _bench(["--gpus", "1", "--exchange-at-1", "--bytes-per-gpu", "3e8", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"])
    assert j["counters"]["matches_generator_tally"] is True
    assert j["config"]["exchange"].startswith("RCCL 2.") and "scfq_comm" in j["config"]["exchange"]
    assert "exchange_note" not in j["config"]
    # and the torch.distributed mirror on the same path
    j2 = _bench(["--gpus", "1", "--exchange-at-1", "--exchange", "torch", "--bytes-per-gpu", "3e8", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"])
    assert j2["counters"] == j["counters"] and "torch.distributed" in j2["config"]["exchange"]


def test_bench_full_size_shard_of_the_scaling_run(gpu):
    """one rank's share of BASELINE configs[2] — 25 GB, 69.5 M records — generated, scanned and exchanged exactly as in the 8-GPU
    run (at this size the generator's one-wave-per-record launch once exceeded 2^32 threads and silently wrote 3 % of the shard)"""
    j = _bench(["--gpus", "1", "--exchange-at-1", "--bytes-per-gpu", "25e9", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"], timeout=900)
    assert j["counters"]["matches_generator_tally"] is True
    assert j["config"]["bytes_per_gpu"] >= 25_000_000_000 and j["counters"]["reads"] > 69_000_000
    assert j["roofline"]["frac"] > 0.5


def test_bench_two_ranks_launched_like_the_driver(gpu):
    """`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` (gloo on one device: RCCL cannot put two ranks on
    one GPU): two shards of one record stream cut at an arbitrary byte, exchange, counters == generator tallies"""
    j = _bench(["--gpus", "2", "--backend", "gloo", "--same-device", "--bytes-per-gpu", "2.5e8", "--steps", "4", "--warmup", "2"], nproc=2)
    assert j["n_gpus"] == 2 and j["scaling"] == "weak"
    assert j["counters"]["matches_generator_tally"] is True
    assert "configs[2]" in j["config"]["workload"] and j["config"]["seed"] == 20260102
    assert j["counters"]["bases"] > 2 * 2.5e8 / 2.5          # both shards are in the folded result
