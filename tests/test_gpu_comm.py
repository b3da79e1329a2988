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


def test_bgzf_input_shards_across_ranks(gpu, scfq, oracle, tmp_path):
    """a BGZF file is cut where its members are: 2 and 3 `sc fq-count --shard-rank` processes (TCP transport, one device), CRLF
    records so that the byte in front of a rank's first inflated byte matters, structure check on (its line-start accounting
    needs that byte too); every rank inflates and scans its own members (h2d_bytes > 0), rank 0 prints the oracle's row"""
    from test_ingest_sources import bgzf_file
    rec = b"@r x\r\nACGTNNGCGC\r\n+\r\nIIII#III@+\r\n@r2\nGGCCN\n+r2\n!!!!!\n"
    data = rec * 150_001 + b"@tail\nACGT"
    f = tmp_path / "crlf.fq.gz"
    f.write_bytes(bgzf_file(data))
    oc = oracle.count(np.frombuffer(data, dtype=np.uint8))
    sc = os.path.join(PKG, "sc")
    for world in (2, 3):
        port = _free_port()
        procs = [subprocess.Popen([sc, "fq-count", "--shard-rank=%d" % r, "--shard-world=%d" % world, "--rendezvous=127.0.0.1:%d" % port,
                                   "--transport=tcp", "--devices=0", "--stats", "--struct-check", str(f)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
                 for r in reversed(range(world))]
        outs = [p.communicate(timeout=300) for p in procs]
        assert [p.returncode for p in procs] == [0] * world, outs
        assert outs[-1][0] == oracle.tsv(oc) + "\n"                          # rank 0: the reference's row
        assert "bad_at=%d\tbad_plus=%d" % (oc.bad_at, oc.bad_plus) in outs[-1][1]
        shares = []
        for so, se in outs:
            st = [json.loads(ln) for ln in se.splitlines() if ln.startswith("{")]
            assert st and st[-1]["h2d_bytes"] > 0 and st[-1]["scan_launches"] > 0, se      # every rank moved compressed bytes and scanned
            shares.append(st[-1]["h2d_bytes"])
        assert max(shares) < 0.75 * sum(shares)                              # nobody did (nearly) all of it
    # SCFQ_SHARD_BGZF=0 and a non-BGZF .gz: rank 0 alone, same row
    import gzip
    g = tmp_path / "plain.fq.gz"
    g.write_bytes(gzip.compress(data, 6))
    for path, env in ((f, {"SCFQ_SHARD_BGZF": "0"}), (g, {})):
        port = _free_port()
        procs = [subprocess.Popen([sc, "fq-count", "--shard-rank=%d" % r, "--shard-world=2", "--rendezvous=127.0.0.1:%d" % port,
                                   "--transport=tcp", "--devices=0", str(path)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                                  env=dict(os.environ, **env)) for r in (1, 0)]
        outs = [p.communicate(timeout=300) for p in procs]
        assert [p.returncode for p in procs] == [0, 0] and outs[1][0] == oracle.tsv(oc) + "\n", outs


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
    j = _bench(["--gpus", "1", "--exchange-at-1", "--bytes-per-gpu", "3e8", "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--ingest-bytes", "0"])
    assert j["counters"]["matches_generator_tally"] is True
    assert j["config"]["exchange"].startswith("RCCL 2.") and "scfq_comm" in j["config"]["exchange"]
    assert "exchange_note" not in j["config"]
    # and the torch.distributed mirror on the same path
    j2 = _bench(["--gpus", "1", "--exchange-at-1", "--exchange", "torch", "--bytes-per-gpu", "3e8", "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--ingest-bytes", "0"])
    assert j2["counters"] == j["counters"] and "torch.distributed" in j2["config"]["exchange"]


def test_bench_full_size_shard_of_the_scaling_run(gpu):
    """one rank's share of BASELINE configs[2] — 25 GB, 69.5 M records — generated, scanned and exchanged exactly as in the 8-GPU
    run (at this size the generator's one-wave-per-record launch once exceeded 2^32 threads and silently wrote 3 % of the shard)"""
    j = _bench(["--gpus", "1", "--exchange-at-1", "--bytes-per-gpu", "25e9", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--ingest-bytes", "0"], timeout=900)
    assert j["counters"]["matches_generator_tally"] is True
    assert j["config"]["bytes_per_gpu"] >= 25_000_000_000 and j["counters"]["reads"] > 69_000_000
    assert j["roofline"]["frac"] > 0.5


def test_bench_two_ranks_launched_like_the_driver(gpu):
    """`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` on one device: two shards of one record stream cut at
    an arbitrary byte, the LIBRARY's communicator (its own rendezvous; TCP transport, because RCCL cannot put two ranks on one
    GPU), counters == generator tallies; and the torch.distributed mirror on the same path"""
    j = _bench(["--gpus", "2", "--same-device", "--transport", "tcp", "--bytes-per-gpu", "2.5e8", "--steps", "4", "--warmup", "2"], nproc=2)
    assert j["n_gpus"] == 2 and j["scaling"] == "weak"
    assert j["counters"]["matches_generator_tally"] is True
    assert "configs[2]" in j["config"]["workload"] and j["config"]["seed"] == 20260102
    assert j["counters"]["bases"] > 2 * 2.5e8 / 2.5          # both shards are in the folded result
    assert "scfq_comm" in j["config"]["exchange"] and "exchange_note" not in j["config"]
    # every rank's own view of the timed region rides on the line (what makes a first 8-GPU curve explain itself)
    pr = j["config"]["per_rank"]
    assert [r["rank"] for r in pr] == [0, 1] and all(r["exchange_path"] == "library" and r["transport"].startswith("tcp") for r in pr), pr
    assert all(r["avg_kernel_ms"] > 0 and r["exchanges_waited_for"] == 4 and r["exchange_wait_ms_per_step"] >= 0 and r["shard_bytes"] > 2e8 for r in pr), pr
    assert max(r["elapsed_s"] for r in pr) * 1e3 / 4 == pytest.approx(j["ms_per_step"], rel=0.02)
    j2 = _bench(["--gpus", "2", "--same-device", "--exchange", "torch", "--bytes-per-gpu", "2.5e8", "--steps", "4", "--warmup", "2"], nproc=2)
    assert j2["counters"] == j["counters"] and "torch.distributed" in j2["config"]["exchange"]
    assert all(r["exchange_path"] == "torch mirror" for r in j2["config"]["per_rank"])


def test_bench_configs2_shape_at_full_shard_size(gpu):
    """BASELINE configs[2] at its real shard size: 25 GB per rank of ONE record stream (seed 20260102) cut at arbitrary bytes, four
    ranks on this one device (100 GB of its 288 GB; the GPU box allows six processes on the card, and pytest and the launcher are
    two of them — with five ranks its process guard killed the run; the 8-rank, 200 GB form needs the 8-GPU node), launched as the
    driver launches the scaling run; every rank scans its own shard with the real kernels, the partials cross the library's
    communicator (own rendezvous, TCP transport: RCCL refuses several ranks on one device), counters of all 278 M records == the
    generator's tallies"""
    j = _bench(["--gpus", "4", "--same-device", "--transport", "tcp", "--steps", "3", "--warmup", "1"], nproc=4, timeout=900)
    assert j["n_gpus"] == 4 and j["scaling"] == "weak"
    assert j["counters"]["matches_generator_tally"] is True
    assert j["config"]["bytes_per_gpu"] >= 25_000_000_000 - 1000 and "configs[2]" in j["config"]["workload"] and j["config"]["seed"] == 20260102
    assert j["counters"]["reads"] > 4 * 69_000_000 and j["counters"]["bases"] > 4 * 25e9 / 2.5
    assert "scfq_comm" in j["config"]["exchange"]


def test_bench_ingest_object(gpu):
    """the non-headline `ingest` object of the default N=1 line: a gzip member and a BGZF file written in setup, a cold process and
    the warm call, counters == generator tallies (here at a fifth of the default size)"""
    j = _bench(["--gpus", "1", "--bytes-per-gpu", "3e8", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--ingest-bytes", "4e8"])
    ing = j["ingest"]
    for k in ("gzip_member", "bgzf"):
        assert ing[k]["counters_match_generator"] is True and ing[k]["compressed_bytes"] < ing[k]["inflated_bytes"] // 2
        assert 0 < ing[k]["warm_wall_s"] <= ing[k]["cold_process_wall_s"]
        marks = [m[0] for m in ing[k]["cold_stages_ms"]["median_run"]["marks"]]      # where a fresh process's time went
        assert "runtime initialised (hipGetDevice returned)" in marks and "context up" in marks and "session folded" in marks
        w = ing[k]["cold_process_walls_s"]
        assert w["min"] <= w["median"] <= w["max"]
        # every cold run says which side it was slow on, and three further processes ran 2 s apart
        assert all(r["scan_kernel_ms"] is not None and r["ingest_wall_ms"] > 0 for r in ing[k]["cold_runs_in_order"])
        assert len(ing[k]["cold_process_walls_2s_apart_s"]["in_order"]) == 3 and ing[k]["cold_process_walls_2s_apart_s"]["median"] > 0
    ho = ing["gzip_member_host_inflate_overlap"]          # the same member inflated on the host, overlapped with copy + scan
    assert ho["counters_match_generator"] is True and ho["scan_kernel_ms"] < 0.5 * ho["host_fill_ms"]
    assert ing["device_bytes_high_water"] > 0


def _gz_members(data, cuts, level=6):
    import zlib
    out = []
    for a, b in zip([0] + cuts, cuts + [len(data)]):
        co = zlib.compressobj(level, zlib.DEFLATED, 31)
        out.append(co.compress(data[a:b]) + co.flush())
    return out


def _run_ranks(sc, world, path, extra=(), env=None):
    port = _free_port()
    procs = [subprocess.Popen([sc, "fq-count", "--shard-rank=%d" % r, "--shard-world=%d" % world, "--rendezvous=127.0.0.1:%d" % port,
                               "--transport=tcp", "--devices=0", "--stats"] + list(extra) + [str(path)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True, env=dict(os.environ, **(env or {}))) for r in reversed(range(world))]
    outs = [p.communicate(timeout=300) for p in procs]
    assert [p.returncode for p in procs] == [0] * world, outs
    stats = []
    for so, se in outs:
        st = [json.loads(ln) for ln in se.splitlines() if ln.startswith("{")]
        stats.append(st[-1] if st else None)
    return outs, stats


def test_gzip_members_shard_across_ranks(gpu, scfq, oracle, tmp_path):
    """an ordinary gzip file of 30 members (`cat a.gz b.gz ...`, pigz -i, a sequencer's writer) is cut where members START: 2 and 3
    `sc fq-count --shard-rank` processes over the TCP transport, each inflating and scanning the members that begin in its byte
    range (device path, and the host's decoder for small stretches), row == oracle.  The members are cut at arbitrary bytes of a CRLF
    record stream — some between a '\\r' and its '\\n', one member is empty — with the structure check on: the byte in front of a rank's
    first inflated byte is only known once the rank before it is through, and is put right at the fold (gz_shard_fix)."""
    # (random bases and qualities: a file that compresses like FASTQ — 3 to 4 times —, so that a rank's stretch is megabytes of
    # deflate data and goes through the device path; sequence and separator lines end in "\r\n", the others in "\n")
    data = fastq_bytes(30_000_000, seed=21).replace(b"\n+\n", b"\r\n+\r\n") + b"@tail\nACGT"
    rng = np.random.default_rng(5)
    cuts = sorted(set(int(x) for x in rng.integers(1, len(data) - 1, 27)))
    # two cuts that fall between a '\r' and its '\n', and one in front of a line start
    k = data.index(b"\r\n", len(data) // 3)
    cuts.append(k + 1)
    k = data.index(b"\r\n", 2 * len(data) // 3)
    cuts.append(k + 1)
    cuts.append(data.index(b"\n@r", len(data) // 2) + 1)
    cuts = sorted(set(cuts))
    members = _gz_members(data, cuts)
    members.insert(11, gzip_empty())
    f = tmp_path / "members.fq.gz"
    f.write_bytes(b"".join(members) + b"\0\0trailing bytes that are not a member")
    oc = oracle.count(np.frombuffer(data, dtype=np.uint8))
    sc = os.path.join(PKG, "sc")
    for world, env in ((2, {}), (3, {}), (3, {"SCFQ_GZ_DEVICE_MIN_MB": "1"})):
        outs, stats = _run_ranks(sc, world, f, ["--struct-check"], env)
        assert outs[-1][0] == oracle.tsv(oc) + "\n", (world, env, outs[-1])
        assert "bad_at=%d\tbad_plus=%d" % (oc.bad_at, oc.bad_plus) in outs[-1][1]
        shares = [st["h2d_bytes"] for st in stats]
        assert all(s > 0 for s in shares) and max(shares) < 0.75 * sum(shares), shares      # every rank inflated and scanned members of its own
    # the quality histogram over the same shards (the '\r' of a "\r\n" cut in two is taken back from the right bin)
    outs, stats = _run_ranks(sc, 3, f, ["--qual-hist"])
    assert outs[-1][0] == oracle.tsv(oc) + "\n"
    want = "\t".join("%d:%d" % (v, oc.qual_hist[v]) for v in range(256) if oc.qual_hist[v])
    assert want in outs[-1][1], (want, outs[-1][1][-600:])
    # SCFQ_SHARD_GZ=0: rank 0 alone, same row
    outs, stats = _run_ranks(sc, 2, f, ["--struct-check"], {"SCFQ_SHARD_GZ": "0"})
    assert outs[-1][0] == oracle.tsv(oc) + "\n" and stats[0]["h2d_bytes"] == 0


def gzip_empty():
    import zlib
    co = zlib.compressobj(6, zlib.DEFLATED, 31)
    return co.compress(b"") + co.flush()


def test_gzip_shard_cut_that_is_no_member_start(gpu, scfq, oracle, tmp_path):
    """what looks like a member start to the rank that begins there — magic, header, deflate data that inflates — but lies INSIDE
    another member's stored block: the rank before it decodes straight through it, the stretch that begins there ends in bytes that
    are no member; every rank learns it from the gathered rows and rank 0 reads the whole file: the row is still the oracle's"""
    import zlib
    rec = b"@r x\nACGTNNGCGC\n+\nIIII#III@+\n"
    inner = gzip_empty()[:-8]                    # header + an empty final block: a "member" whose trailer never comes
    inner_full = _gz_members(b"@in\nAC\n+\n!!\n" * 50, [])[0]
    part1 = rec * 150_000
    part2 = rec * 120_000 + b"@bin " + inner_full + b" x\nACGT\n+\nIIII\n" + rec * 30_000
    assert inner  # (kept for clarity: the embedded member is a complete one, so the rank that starts there decodes it cleanly)
    m1 = _gz_members(part1, [], 6)[0]
    co = zlib.compressobj(0, zlib.DEFLATED, 31)          # level 0: stored blocks, the embedded bytes appear verbatim
    m2 = co.compress(part2) + co.flush()
    blob = m1 + m2
    at = blob.find(inner_full)
    if at < 0:
        pytest.skip("the embedded member straddles a stored-block boundary in this zlib build")
    f = tmp_path / "embedded.fq.gz"
    f.write_bytes(blob)
    data = part1 + part2
    oc = oracle.count(np.frombuffer(data, dtype=np.uint8))
    # two ranks: the nominal cut must lie inside m2 but in front of the embedded member, with no true member start in between
    assert len(m1) < len(blob) // 2 < at, (len(m1), len(blob) // 2, at)
    sc = os.path.join(PKG, "sc")
    outs, stats = _run_ranks(sc, 2, f, [], {"SCFQ_VERBOSE": "1"})
    assert outs[-1][0] == oracle.tsv(oc) + "\n", outs[-1]


def test_one_gzip_member_shards_across_ranks(gpu, scfq, oracle, tmp_path):
    """the common case — ONE gzip member (gzip / pigz output) — over 2 and 3 ranks: the deflate stream is cut where blocks start; every
    rank searches, decodes and proves its stretch, folds what it does to the 32 KiB window into a MAP and keeps the proven symbols; the
    maps cross the communicator, composed in rank order they give every rank the window in front of its stretch, and the kept symbols
    become bytes, CRC tiles and a scan (or, SCFQ_SHARD_GZ_KEEP=0, the stretch is decoded a second time); the member's CRC-32 / ISIZE are
    checked against the join of the stretches', and the partials fold with the byte in front of each stretch put right.  CRLF records, structure check and quality
    histogram on: row, bad_at / bad_plus and the histogram are the oracle's; every rank moved compressed bytes and scanned."""
    import gzip
    data = fastq_bytes(90_000_000, seed=33).replace(b"\n+\n", b"\r\n+\r\n") + b"@tail\nACGT"
    f = tmp_path / "one_member.fq.gz"
    f.write_bytes(gzip.compress(data, 6) + b"\0\0 trailing bytes")
    oc = oracle.count(np.frombuffer(data, dtype=np.uint8))
    sc = os.path.join(PKG, "sc")
    want_hist = "\t".join("%d:%d" % (v, oc.qual_hist[v]) for v in range(256) if oc.qual_hist[v])
    # (SCFQ_SHARD_GZ_KEEP=0: two passes per rank, the second decoding again; the default keeps the proven symbols between map and bytes)
    for world, keep in ((2, "1"), (3, "1"), (2, "0"), (3, "0")):
        outs, stats = _run_ranks(sc, world, f, ["--struct-check", "--qual-hist"], {"SCFQ_VERBOSE": "1", "SCFQ_SHARD_GZ_KEEP": keep})
        assert outs[-1][0] == oracle.tsv(oc) + "\n", (world, keep, outs[-1][0], outs[-1][1][-2500:])
        assert "bad_at=%d\tbad_plus=%d" % (oc.bad_at, oc.bad_plus) in outs[-1][1] and want_hist in outs[-1][1]
        shares = [st["h2d_bytes"] for st in stats]
        assert all(s > 0 for s in shares) and max(shares) < 0.75 * sum(shares), shares
        assert "the block cuts of a one-member file did not join up" not in outs[-1][1]
    # SCFQ_SHARD_GZ_BLOCKS=0: rank 0 alone (the member scheme finds no second member), same row
    outs, stats = _run_ranks(sc, 2, f, ["--struct-check"], {"SCFQ_SHARD_GZ_BLOCKS": "0"})
    assert outs[-1][0] == oracle.tsv(oc) + "\n" and stats[0]["h2d_bytes"] == 0
    # a damaged member (one bit flipped in the second half): the joined CRC-32 does not match, rank 0 reads the whole file and
    # reports what gzread reports; nobody prints a row
    blob = bytearray(f.read_bytes())
    blob[len(blob) * 2 // 3] ^= 0x10
    bad = tmp_path / "damaged.fq.gz"
    bad.write_bytes(bytes(blob))
    port = _free_port()
    procs = [subprocess.Popen([sc, "fq-count", "--shard-rank=%d" % r, "--shard-world=2", "--rendezvous=127.0.0.1:%d" % port, "--transport=tcp", "--devices=0", str(bad)],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in (1, 0)]
    outs = [p.communicate(timeout=300) for p in procs]
    single = subprocess.run([sc, "fq-count", str(bad)], capture_output=True, text=True, timeout=300)
    assert single.returncode != 0 and procs[1].returncode == single.returncode and outs[1][0] == single.stdout == ""
    assert procs[0].returncode != 0 and outs[0][0] == ""


def test_a_few_big_gzip_members_shard_across_ranks(gpu, scfq, oracle, tmp_path):
    """`cat lane1.fq.gz lane2.fq.gz`: two members of 20 MB (compressed) each over 2, 3 and 4 ranks.  The member scheme would leave the ranks
    in whose share no member starts without work; with at most one member start per share the ranks cut where BLOCKS start, a member
    start being a cut of its own: a stretch never crosses a member's end, the window maps compose inside a member only, and each member's
    CRC-32 is the join of the stretches that hold it.  Row, structure check and histogram == oracle; every rank moved bytes and scanned."""
    import gzip
    # (the first member the longer one: the second then starts inside the second rank's share whatever the number of ranks here — a share
    # that holds two member starts, or a second one in rank 0's, sends the file to the member scheme)
    a = fastq_bytes(66_000_000, seed=41).replace(b"\n+\n", b"\r\n+\r\n")
    b = fastq_bytes(56_000_000, seed=42)
    data = a + b
    f = tmp_path / "two_lanes.fq.gz"
    f.write_bytes(gzip.compress(a, 6) + gzip.compress(b, 6))
    oc = oracle.count(np.frombuffer(data, dtype=np.uint8))
    sc = os.path.join(PKG, "sc")
    want_hist = "\t".join("%d:%d" % (v, oc.qual_hist[v]) for v in range(256) if oc.qual_hist[v])
    for world in (2, 3, 4):
        outs, stats = _run_ranks(sc, world, f, ["--struct-check", "--qual-hist"], {"SCFQ_VERBOSE": "1"})
        assert outs[-1][0] == oracle.tsv(oc) + "\n", (world, outs[-1][0], outs[-1][1][-2500:])
        assert "bad_at=%d\tbad_plus=%d" % (oc.bad_at, oc.bad_plus) in outs[-1][1] and want_hist in outs[-1][1]
        shares = [st["h2d_bytes"] for st in stats]
        assert all(s > 0 for s in shares) and max(shares) < 0.75 * sum(shares), (world, shares)
        assert "did not join up" not in outs[-1][1]
