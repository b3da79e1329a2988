"""Parity of the HIP path (through the C ABI) against the CPU oracle: bit-exact integer counters."""
import ctypes
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden_rows

pytestmark = pytest.mark.gpu

REF_FIELDS = ("reads", "gc_bases", "n_bases", "bases", "lines", "newlines", "input_bytes")


def to_dev(torch, arr, offset=0, pad=8192):
    """device copy of arr placed `offset` bytes past a 4 KiB-aligned address, with guard bytes around"""
    t = torch.full((pad + offset + arr.size + pad,), 0x47, dtype=torch.uint8, device="cuda")   # guards are 'G'
    base = t.data_ptr()
    start = ((base + pad + 4095) // 4096) * 4096 - base + offset
    t[start:start + arr.size] = torch.from_numpy(arr.copy())
    torch.cuda.synchronize()
    return t, base + start


def random_fastq_like(rng, n, kind):
    if kind == "uniform":      # every byte value, newlines frequent
        a = rng.integers(0, 256, n, dtype=np.uint8)
    elif kind == "ascii":
        a = rng.choice(np.frombuffer(b"ACGTNacgtn@+FI#:,\r\n\n\n", dtype=np.uint8), n)
    elif kind == "dense_nl":
        a = rng.choice(np.frombuffer(b"\n\n\n\nG\r", dtype=np.uint8), n)
    elif kind == "sparse_nl":
        a = rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), n)
        if n:
            idx = rng.integers(0, n, max(1, n // 3000))
            a[idx] = 10
    elif kind == "crlf":
        a = rng.choice(np.frombuffer(b"ACGTN@+I", dtype=np.uint8), n)
        if n > 4:
            idx = np.sort(rng.choice(n - 1, max(1, n // 40), replace=False))
            a[idx] = 13
            a[idx + 1] = 10
    else:
        raise ValueError(kind)
    return a.astype(np.uint8)


def assert_partial_equal(p, ow, ctx=""):
    got = p.words()[:25]
    assert got == [int(x) for x in ow[:25]], ctx


def test_golden_files_through_count_file(gpu, scfq, oracle):
    for row in golden_rows():
        path = os.path.join(GOLDEN, row["name"])
        c = scfq.count_file(path)
        assert (c.reads, c.gc_bases, c.n_bases, c.bases) == (row["reads"], row["gc_bases"], row["n_bases"], row["bases"]), row["name"]
        assert scfq.format_tsv(c).split("\t")[1] == row["gc_content"], row["name"]
        rc, oc = oracle.count_file(path)
        assert rc == 0
        for f in REF_FIELDS:
            assert getattr(c, f) == getattr(oc, f), (row["name"], f)


@pytest.mark.parametrize("kind", ["uniform", "ascii", "dense_nl", "sparse_nl", "crlf"])
def test_random_buffers_device_resident(gpu, scfq, oracle, kind):
    # (a fixed seed per kind: Python's string hash is randomised per process, and a failure must be reproducible on the next box)
    rng = np.random.default_rng({"uniform": 20260501, "ascii": 20260502, "dense_nl": 20260503, "sparse_nl": 20260504, "crlf": 20260505}[kind])
    sizes = [0, 1, 2, 15, 16, 17, 63, 64, 65, 255, 1023, 4095, 4096, 4097, 8191, 12288, 12289, 40000, 70001, 300000, 1 << 20]
    for n in sizes:
        for offset in (0, 1, 17, 63, 64, 1000, 4095):
            a = random_fastq_like(rng, n, kind)
            t, ptr = to_dev(gpu, a, offset)
            for prev in (-1, 10, 13, 65):
                ow = oracle.partial(a, prev)
                p = scfq.partial_device(ptr, n, prev)
                assert_partial_equal(p, [ow[k] if k not in range(13, 25) else 0 for k in range(27)], (kind, n, offset, prev))
                ps = scfq.partial_simple_device(ptr, n, prev)
                assert ps.words()[:25] == [int(x) for x in ow[:25]], ("simple", kind, n, offset, prev)
            if n:
                assert p.bytes == n and p.last_byte == int(a[-1])


@pytest.mark.parametrize("kind", ["uniform", "ascii", "crlf"])
def test_struct_and_hist_variants(gpu, scfq, oracle, kind):
    rng = np.random.default_rng(7 + len(kind))
    for n in (0, 1, 100, 4096, 5000, 70001, 400000):
        for offset in (0, 33, 4095):
            a = random_fastq_like(rng, n, kind)
            t, ptr = to_dev(gpu, a, offset)
            for prev in (-1, 10, 13):
                ow, oh = oracle.partial(a, prev, want_hist=True)
                p = scfq.partial_device(ptr, n, prev, flags=scfq.SCFQ_STRUCT_CHECK)
                assert_partial_equal(p, ow, ("struct", kind, n, offset, prev))
                p2, h2 = scfq.partial_device(ptr, n, prev, flags=scfq.SCFQ_QUAL_HIST | scfq.SCFQ_HIST_EXACT | scfq.SCFQ_STRUCT_CHECK, want_hist=True)
                assert_partial_equal(p2, ow, ("hist+struct", kind, n, offset, prev))
                assert list(h2) == oh and p2.hist_class == 0, ("hist", kind, n, offset, prev)
                # speculative form (default): the class it reports complete is exact; 0 = all four are
                p3, h3 = scfq.partial_device(ptr, n, prev, flags=scfq.SCFQ_QUAL_HIST | scfq.SCFQ_STRUCT_CHECK, want_hist=True)
                assert_partial_equal(p3, ow, ("spec hist+struct", kind, n, offset, prev))
                assert p3.hist_class in (0, 1, 2, 3, 4)
                if p3.hist_class == 0:
                    assert list(h3) == oh, ("spec hist", kind, n, offset, prev)
                else:
                    k = p3.hist_class - 1
                    assert list(h3)[k * 256:(k + 1) * 256] == oh[k * 256:(k + 1) * 256], ("spec hist class", kind, n, offset, prev, k)


def test_shard_boundaries_every_offset(gpu, scfq, oracle):
    """cut one buffer at every byte offset of a few records (incl. inside \\r\\n): combine == whole"""
    rec = b"@r1 x\r\nACGTNNGCGC\r\n+\r\nIIII#III@+\r\n@r2\nGGCC\n+r2\n!!!!\n"
    data = np.frombuffer(rec * 3, dtype=np.uint8)
    t, ptr = to_dev(gpu, data, 5)
    whole = oracle.partial(data, -1)
    for cut in range(0, data.size + 1):
        a = scfq.partial_device(ptr, cut, -1, flags=scfq.SCFQ_STRUCT_CHECK)
        b = scfq.partial_device(ptr + cut, data.size - cut, int(data[cut - 1]) if cut else -1, flags=scfq.SCFQ_STRUCT_CHECK)
        acc = scfq.identity()
        scfq.combine(acc, a)
        scfq.combine(acc, b)
        assert acc.words()[:25] == whole[:25], cut
        # the halo can also be read from memory
        if cut:
            b2 = scfq.partial_device(ptr + cut, data.size - cut, 0, flags=scfq.SCFQ_STRUCT_CHECK | scfq.SCFQ_PREV_IN_MEMORY)
            assert b2.words()[:25] == b.words()[:25]


def test_host_buffer_chunked_ingest(gpu, scfq, oracle):
    rng = np.random.default_rng(11)
    a = random_fastq_like(rng, 3_000_000, "crlf")
    oc = oracle.count(a)
    for chunk in (4096, 65536, 1 << 20, 0):
        c = scfq.count_host(a, chunk_bytes=chunk, flags=scfq.SCFQ_QUAL_HIST | scfq.SCFQ_STRUCT_CHECK)
        for f in REF_FIELDS + ("bad_at", "bad_plus"):
            assert getattr(c, f) == getattr(oc, f), (chunk, f)
        assert list(c.qual_hist) == list(oc.qual_hist)
        assert scfq.format_tsv(c) == oracle.tsv(oc)


def test_synthetic_workloads_match_oracle_and_generator(gpu, scfq, oracle):
    torch = gpu
    for kind, seed, nbytes in ((scfq_kind, s, b) for scfq_kind, s, b in ((0, 20260101, 64 << 20), (1, 20260103, 64 << 20))):
        plan = scfq.synth_plan(kind, seed, nbytes)
        t = torch.empty(plan.bytes + 4096, dtype=torch.uint8, device="cuda")
        info = scfq.synth_device(kind, seed, plan.records, t.data_ptr(), plan.bytes)
        assert info.bytes == plan.bytes
        host, hinfo = scfq.synth_host(kind, seed, plan.records)
        assert np.array_equal(t[:plan.bytes].cpu().numpy(), host), "device and host generators differ"
        assert (info.gc_bases, info.n_bases, info.bases) == (hinfo.gc_bases, hinfo.n_bases, hinfo.bases)
        c = scfq.count_device(t.data_ptr(), plan.bytes, flags=scfq.SCFQ_STRUCT_CHECK)
        oc = oracle.count(host, "lines")
        for f in REF_FIELDS:
            assert getattr(c, f) == getattr(oc, f), (kind, f)
        assert (c.reads, c.gc_bases, c.n_bases, c.bases) == (plan.records, info.gc_bases, info.n_bases, info.bases)
        assert c.bad_at == 0 and c.bad_plus == 0


def test_full_size_config_properties(gpu, scfq, oracle):
    """BASELINE configs[1] at full size (10 GB Illumina, HBM-resident): size-independent properties.
    exact generator tallies, shard additivity at arbitrary cut points, idempotence, structure and histogram sums."""
    torch = gpu
    kind, seed = 0, 20260101
    plan = scfq.synth_plan(kind, seed, 10_000_000_000)
    buf = torch.empty(plan.bytes + 4096, dtype=torch.uint8, device="cuda")
    info = scfq.synth_device(kind, seed, plan.records, buf.data_ptr(), plan.bytes)
    assert info.bytes == plan.bytes
    ptr, n = buf.data_ptr(), plan.bytes
    whole = scfq.partial_device(ptr, n, -1)
    c = scfq.finalize(whole)
    assert (c.reads, c.gc_bases, c.n_bases, c.bases) == (plan.records, info.gc_bases, info.n_bases, info.bases)
    assert c.lines == 4 * plan.records and c.newlines == c.lines and c.input_bytes == n
    assert scfq.partial_device(ptr, n, -1).words() == whole.words()                     # idempotent
    # shard additivity: 8 shards at arbitrary (odd, unaligned) cut points, halo read from memory
    rng = np.random.default_rng(3)
    cuts = [0] + sorted(int(x) for x in rng.integers(1, n, 7)) + [n]
    acc = scfq.identity()
    for a, b in zip(cuts[:-1], cuts[1:]):
        flags = scfq.SCFQ_PREV_IN_MEMORY if a else 0
        scfq.combine(acc, scfq.partial_device(ptr + a, b - a, -1, flags=flags))
    assert acc.words()[:27] == whole.words()[:27]
    # the first 64 MiB against the CPU oracle, bit-exact
    m = 64 << 20
    host = buf[:m].cpu().numpy()
    assert scfq.partial_device(ptr, m, -1).words()[:13] == [int(x) for x in oracle.partial(host, -1)[:13]]
    # structure check + histogram variants at full size
    cs = scfq.count_device(ptr, n, flags=scfq.SCFQ_STRUCT_CHECK | scfq.SCFQ_QUAL_HIST)
    assert (cs.reads, cs.gc_bases, cs.n_bases, cs.bases, cs.bad_at, cs.bad_plus) == (c.reads, c.gc_bases, c.n_bases, c.bases, 0, 0)
    hist = list(cs.qual_hist)
    assert sum(hist) == c.bases                                                         # every quality line is as long as its read
    assert {v for v in range(256) if hist[v]} == {ord("F"), ord(":"), ord(","), ord("#")}
    assert abs(hist[ord("F")] / c.bases - 0.90) < 0.001


def test_long_read_workload(gpu, scfq, oracle):
    """BASELINE configs[4]: Nanopore-style 500 bp..50 kb reads (lines far longer than a 4 KiB tile)"""
    torch = gpu
    kind, seed = 1, 20260103
    plan = scfq.synth_plan(kind, seed, 1_000_000_000)
    buf = torch.empty(plan.bytes + 4096, dtype=torch.uint8, device="cuda")
    info = scfq.synth_device(kind, seed, plan.records, buf.data_ptr(), plan.bytes)
    c = scfq.count_device(buf.data_ptr(), plan.bytes, flags=scfq.SCFQ_STRUCT_CHECK | scfq.SCFQ_QUAL_HIST)
    assert (c.reads, c.gc_bases, c.n_bases, c.bases, c.bad_at, c.bad_plus) == (plan.records, info.gc_bases, info.n_bases, info.bases, 0, 0)
    hist = list(c.qual_hist)
    assert sum(hist) == c.bases and {v for v in range(256) if hist[v]} == set(range(34, 74))
    host = buf[:plan.bytes].cpu().numpy()
    oc = oracle.count(host, "lines")
    assert (c.reads, c.gc_bases, c.n_bases, c.bases, c.lines) == (oc.reads, oc.gc_bases, oc.n_bases, oc.bases, oc.lines)


def test_worst_case_newline_density(gpu, scfq, oracle):
    """pathological inputs: only newlines, alternating newline / base, 0x80+ bytes everywhere"""
    torch = gpu
    n = 8 << 20
    for pattern in (b"\n", b"G\n", b"\r\n", b"\n\nN", bytes([0xC7, 0x8A, 10, 0xCE])):
        a = np.frombuffer((pattern * (n // len(pattern) + 1))[:n], dtype=np.uint8)
        t, ptr = to_dev(torch, a, 3)
        for flags in (0, scfq.SCFQ_STRUCT_CHECK):
            p = scfq.partial_device(ptr, n, -1, flags=flags)
            ow = oracle.partial(a, -1)
            want = [int(x) for x in ow[:25]] if flags else [int(x) for x in ow[:13]] + [0] * 12
            assert p.words()[:25] == want, (pattern, flags)


def test_multi_device_option_paths(gpu, scfq, oracle, tmp_path):
    """scfq_opts.n_devices > 1: byte-range shards at arbitrary cut points, one ingest thread per listed device, host
    fold in shard order. A 1-GPU box lists device 0 several times (sessions on one device are serialised)."""
    rng = np.random.default_rng(21)
    a = random_fastq_like(rng, 9_000_001, "crlf")
    oc = oracle.count(a)
    for devs in ([0, 0], [0, 0, 0]):
        c = scfq.count_host(a, devices=devs, flags=scfq.SCFQ_QUAL_HIST | scfq.SCFQ_STRUCT_CHECK, chunk_bytes=1 << 20)
        for f in REF_FIELDS + ("bad_at", "bad_plus"):
            assert getattr(c, f) == getattr(oc, f), (devs, f)
        assert list(c.qual_hist) == list(oc.qual_hist)
    p = tmp_path / "big.fq"
    a.tofile(p)
    c = scfq.count_file(str(p), devices=[0, 0, 0, 0], chunk_bytes=1 << 20)
    for f in REF_FIELDS:
        assert getattr(c, f) == getattr(oc, f), f
