"""Property tests (hypothesis) on the CPU: (a) the fq-dedup oracle against an independent, line-by-line Python restatement of
src/fq_dedup.nim:42-73 (the oracle is what the GPU pipeline is compared with, so it gets a second opinion); (b) the library's
gzip readers against zlib over random payloads, levels, strategies and member layouts."""
import gzip
import os
import zlib

import numpy as np
from hypothesis import HealthCheck, given, settings, strategies as st

from test_inflate_host import gz_member, raw_deflate


def nim_lines(data: bytes):
    """Nim 1.0.6 `lines(stream)`: '\\n' ends a line, a '\\r' directly before it is dropped, a final line without '\\n' counts"""
    out, pos = [], 0
    while pos < len(data):
        k = data.find(b"\n", pos)
        if k < 0:
            out.append(data[pos:])
            break
        line = data[pos:k]
        if line.endswith(b"\r"):
            line = line[:-1]
        out.append(line)
        pos = k + 1
    return out


def dedup_reference(data: bytes):
    """fq_dedup.nim, transliterated: pass 1 finds the IDs seen more than once (the Bloom filter is exact here), pass 2 echoes"""
    lines = nim_lines(data)
    seen, check = set(), {}
    for i, rec in enumerate(lines):                       # :42-47
        if i % 4 == 0:
            if rec in seen:
                check[rec] = check.get(rec, 0) + 1
            seen.add(rec)
    n_reads = len(lines) // 4                             # :49
    out, putative, write_ln, n_dups = [], {}, True, 0
    for i0, rec in enumerate(lines):                      # :57-73 (i is incremented first: (i-1) mod 4 == 0)
        if i0 % 4 == 0:
            if rec not in check:
                out.append(rec); write_ln = True
                continue
            putative[rec] = putative.get(rec, 0) + 1
            if putative[rec] > 1:
                write_ln = False; n_dups += 1
                continue
            out.append(rec); write_ln = True
        elif write_ln:
            out.append(rec)
    return b"".join(l + b"\n" for l in out), n_reads, n_dups


line_st = st.binary(max_size=12).map(lambda b: b.replace(b"\n", b"N"))
id_st = st.sampled_from([b"@a", b"@b", b"@a ", b"@", b"", b"@c\r", b"@a\r"])
eol_st = st.sampled_from([b"\n", b"\r\n"])


@st.composite
def fastq_like(draw):
    n = draw(st.integers(0, 12))
    parts = []
    for _ in range(n):
        eol = draw(eol_st)
        parts += [draw(id_st), eol, draw(line_st), eol, b"+", eol, draw(line_st), eol]
    blob = b"".join(parts)
    cut = draw(st.integers(0, len(blob)))
    return blob[: len(blob) - cut] if draw(st.booleans()) else blob


@settings(max_examples=300, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(fastq_like())
def test_dedup_oracle_equals_the_transliteration(oracle, data):
    want, n_reads, n_dups = dedup_reference(data)
    got, stats = oracle.dedup(data)
    assert got == want
    assert (stats.total_reads, stats.duplicates) == (n_reads, n_dups)


@settings(max_examples=40, deadline=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture])
@given(seed=st.integers(0, 2**31), level=st.sampled_from([0, 1, 4, 6, 9]), strategy=st.sampled_from([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_RLE, zlib.Z_HUFFMAN_ONLY, zlib.Z_FILTERED]),
       members=st.integers(1, 4), alphabet=st.sampled_from([b"ACGTN\n", b"F:,#\n", bytes(range(256)), b"A"]), size=st.integers(0, 300_000))
def test_gzip_readers_equal_zlib(scfq, tmp_path_factory, seed, level, strategy, members, alphabet, size):
    rng = np.random.default_rng(seed)
    data = bytes(rng.choice(np.frombuffer(alphabet, dtype=np.uint8), size))
    cuts = sorted(int(x) for x in rng.integers(0, size + 1, members - 1))
    pieces = [data[a:b] for a, b in zip([0] + cuts, cuts + [size])]
    blob = b"".join(gz_member(p, raw_deflate(p, level, strategy)) for p in pieces)
    assert gzip.decompress(blob) == data
    f = tmp_path_factory.mktemp("gz") / "x.fq.gz"
    f.write_bytes(blob)
    old = {k: os.environ.get(k) for k in ("SCFQ_PGZ_MIN_MB", "SCFQ_PGZ_SEGMENT_MB")}
    try:
        for env in ({"SCFQ_PGZ_MIN_MB": "1000"}, {"SCFQ_PGZ_MIN_MB": "0", "SCFQ_PGZ_SEGMENT_MB": "1"}):     # serial reader, parallel reader
            os.environ.update(env)
            assert scfq.debug_read_file(str(f), size + 16, 1 << 16) == data
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def count_reference(data: bytes):
    """src/fq_count.nim:38-45, transliterated (1-based line counter, `count` of the single letters G, C and N)"""
    reads = gc = n = bases = 0
    for i, line in enumerate(nim_lines(data), start=1):
        if i % 4 == 1:
            reads += 1
        if i % 4 == 2:
            gc += line.count(b"G") + line.count(b"C")
            n += line.count(b"N")
            bases += len(line)
    return reads, gc, n, bases


seq_st = st.text(alphabet="ACGTNacgtn\r @+", max_size=20).map(str.encode)


@st.composite
def ragged_text(draw):
    lines = draw(st.lists(seq_st, max_size=24))
    blob = b"".join(l + draw(eol_st) for l in lines)
    return blob[:-1] if blob and draw(st.booleans()) else blob


@settings(max_examples=400, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(ragged_text())
def test_count_oracle_equals_the_transliteration(oracle, data):
    want = count_reference(data)
    for which in ("bytes", "lines"):
        c = oracle.count(data, which)
        assert (c.reads, c.gc_bases, c.n_bases, c.bases) == want, which
