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
    rng = np.random.default_rng(hash(kind) % 2**32)
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
                p2, h2 = scfq.partial_device(ptr, n, prev, flags=scfq.SCFQ_QUAL_HIST | scfq.SCFQ_STRUCT_CHECK, want_hist=True)
                assert_partial_equal(p2, ow, ("hist+struct", kind, n, offset, prev))
                assert list(h2) == oh, ("hist", kind, n, offset, prev)


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
