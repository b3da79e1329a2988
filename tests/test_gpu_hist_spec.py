"""K3 speculative form (the default with SCFQ_QUAL_HIST): guesses are only a fast path. Whatever the input —
well-formed, shifted by stray lines, quality lines that start with '@' / '+', CRLF, long reads, random bytes — the
quality histogram must equal the oracle's, and on well-formed input the fast form must actually be the one that ran."""
import ctypes

import numpy as np
import pytest

from test_gpu_parity import to_dev, random_fastq_like

pytestmark = pytest.mark.gpu


def make_fastq(rng, n_records, read_len=(30, 200), crlf=False, nasty_quals=False, header_len=(10, 60)):
    """well-formed 4-line records; nasty_quals: quality lines drawn from an alphabet rich in '@', '+' and newline-adjacent bytes"""
    eol = b"\r\n" if crlf else b"\n"
    out = []
    qual_alpha = np.frombuffer(b"@+@+FFFF:,#I" if nasty_quals else b"FFFFFFFF:,#I5?", dtype=np.uint8)
    for i in range(n_records):
        L = int(rng.integers(read_len[0], read_len[1] + 1))
        hl = int(rng.integers(header_len[0], header_len[1] + 1))
        head = b"@" + bytes(rng.choice(np.frombuffer(b"abcXYZ0123456789:/ ", dtype=np.uint8), hl))
        seq = bytes(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), L, p=[.29, .2, .2, .29, .02]))
        qual = bytes(rng.choice(qual_alpha, L))
        plus = b"+" if i % 3 else b"+" + head[1:]
        out += [head, eol, seq, eol, plus, eol, qual, eol]
    return np.frombuffer(b"".join(out), dtype=np.uint8)


def check_counts(scfq, oracle, torch, a, ctx, expect_fast=None, offset=0):
    t, ptr = to_dev(torch, a, offset)
    oc = oracle.count(a)
    for flags in (scfq.SCFQ_QUAL_HIST, scfq.SCFQ_QUAL_HIST | scfq.SCFQ_STRUCT_CHECK):
        c = scfq.count_device(ptr, a.size, flags=flags)
        fast, redone = scfq.hist_stats()
        assert list(c.qual_hist) == list(oc.qual_hist), (ctx, flags, fast, redone)
        for f in ("reads", "gc_bases", "n_bases", "bases", "lines", "newlines"):
            assert getattr(c, f) == getattr(oc, f), (ctx, f)
        if expect_fast is True:
            assert fast > 0 and redone * 20 <= fast + redone, (ctx, fast, redone)
        if expect_fast is False:
            assert fast == 0, (ctx, fast, redone)
    ce = scfq.count_device(ptr, a.size, flags=scfq.SCFQ_QUAL_HIST | scfq.SCFQ_HIST_EXACT)
    assert list(ce.qual_hist) == list(oc.qual_hist), (ctx, "exact")
    assert scfq.hist_stats() == (0, 0)
    return t


@pytest.mark.parametrize("crlf", [False, True])
@pytest.mark.parametrize("nasty", [False, True])
def test_well_formed_fastq_takes_the_fast_form(gpu, scfq, oracle, crlf, nasty):
    rng = np.random.default_rng(100 + 2 * crlf + nasty)
    for n_rec, offset in ((1, 0), (3, 7), (50, 0), (3000, 1), (40000, 4095), (150000, 64)):
        a = make_fastq(rng, n_rec, crlf=crlf, nasty_quals=nasty)
        # tiny inputs may not show a header/separator pair inside the look-ahead window of a range: no expectation there
        check_counts(scfq, oracle, gpu, a, ("wellformed", crlf, nasty, n_rec), expect_fast=True if n_rec >= 3000 else None,
                     offset=offset)


def test_long_reads_take_the_fast_form(gpu, scfq, oracle):
    rng = np.random.default_rng(5)
    a = make_fastq(rng, 300, read_len=(500, 50000), header_len=(40, 120))
    check_counts(scfq, oracle, gpu, a, "long reads", expect_fast=True)


def test_generator_workloads(gpu, scfq, oracle):
    for kind, seed in ((0, 20260101), (1, 20260103)):
        a, info = scfq.synth_host(kind, seed, scfq.synth_plan(kind, seed, 48 << 20).records)
        check_counts(scfq, oracle, gpu, a, ("synth", kind), expect_fast=True)


def test_phase_shifts_inside_the_input_are_redone_exactly(gpu, scfq, oracle):
    """stray lines shift the line phase mid-file: guesses after the shift name the wrong class relative to the start of the
    input; verification must catch every one of them"""
    rng = np.random.default_rng(9)
    base = make_fastq(rng, 30000)
    nl = np.flatnonzero(base == 10)
    for n_extra in (1, 2, 3, 5):
        pieces, last = [], 0
        for cut in sorted(rng.choice(nl, n_extra, replace=False)):
            pieces += [base[last:cut + 1], np.frombuffer(b"stray line\n", dtype=np.uint8)]
            last = cut + 1
        pieces.append(base[last:])
        a = np.concatenate(pieces)
        check_counts(scfq, oracle, gpu, a, ("shifted", n_extra))


def test_inputs_that_fool_the_guess(gpu, scfq, oracle):
    rng = np.random.default_rng(13)
    # (a) header and separator roles swapped relative to the start of the input (file starts mid-record)
    a = make_fastq(rng, 20000)
    first_nl = int(np.flatnonzero(a == 10)[0])
    for drop_lines in (1, 2, 3):
        nls = np.flatnonzero(a == 10)
        check_counts(scfq, oracle, gpu, a[nls[drop_lines - 1] + 1:], ("starts mid-record", drop_lines))
    # (b) every line starts with '@' or '+' at random
    lines = []
    for i in range(60000):
        L = int(rng.integers(0, 90))
        lines.append(bytes(rng.choice(np.frombuffer(b"@+", dtype=np.uint8), 1)) + bytes(rng.choice(np.frombuffer(b"ACGTNFI#@+", dtype=np.uint8), L)))
    check_counts(scfq, oracle, gpu, np.frombuffer(b"\n".join(lines), dtype=np.uint8), "random line starts")
    # (c) two-line records "@h\n+\n": '@' in classes 0 and 2, '+' in classes 1 and 3 -> ambiguous
    check_counts(scfq, oracle, gpu, np.frombuffer(b"@hdr\n+\n" * 50000, dtype=np.uint8), "two-line records")
    # (d) very short records: many quality segments per 64-byte lane
    check_counts(scfq, oracle, gpu, np.frombuffer(b"@r\nA\n+\nI\n@s\nGC\n+\n#@\n" * 40000, dtype=np.uint8), "tiny records", expect_fast=True)
    # (e) random bytes of several distributions
    for kind in ("uniform", "ascii", "dense_nl", "sparse_nl", "crlf"):
        check_counts(scfq, oracle, gpu, random_fastq_like(rng, 1_500_000, kind), ("random", kind))
    # (f) no newline at all / only newlines
    check_counts(scfq, oracle, gpu, np.full(300000, ord("F"), dtype=np.uint8), "no newline", expect_fast=False)
    check_counts(scfq, oracle, gpu, np.full(300000, 10, dtype=np.uint8), "only newlines", expect_fast=False)


def test_shards_with_unknown_start_phase(gpu, scfq, oracle):
    """scfq_partial_buffer does not know the line phase at the first byte of a shard: the fast form completes one class
    (hist_class) relative to the shard start; combine rotates it, finalize checks it"""
    rng = np.random.default_rng(21)
    a = make_fastq(rng, 60000, crlf=True, nasty_quals=True)
    t, ptr = to_dev(gpu, a, 3)
    oc = oracle.count(a)
    for n_shards in (2, 3, 5):
        cuts = [0] + sorted(int(x) for x in rng.integers(1, a.size - 1, n_shards - 1)) + [a.size]
        acc = scfq.identity()
        acc_h = (ctypes.c_uint64 * scfq.HIST_WORDS)()
        classes = []
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            p, h = scfq.partial_device(ptr + lo, hi - lo, int(a[lo - 1]) if lo else -1, flags=scfq.SCFQ_QUAL_HIST, want_hist=True)
            classes.append(p.hist_class)
            scfq.combine(acc, p, acc_h, h)
        assert all(c in (1, 2, 3, 4) for c in classes), classes      # every shard was served by the fast form
        assert acc.hist_class == 4
        c = scfq.finalize(acc, acc_h)
        assert list(c.qual_hist) == list(oc.qual_hist), (n_shards, cuts)
    # a shard whose own content is consistent but shifted against the others: combine flags it, finalize refuses
    stray = np.frombuffer(b"stray\n", dtype=np.uint8)
    b = np.concatenate([a, stray, a])
    tb, pb = to_dev(gpu, b, 0)
    cut = a.size + stray.size
    p1, h1 = scfq.partial_device(pb, cut, -1, flags=scfq.SCFQ_QUAL_HIST, want_hist=True)
    p2, h2 = scfq.partial_device(pb + cut, b.size - cut, 10, flags=scfq.SCFQ_QUAL_HIST, want_hist=True)
    acc = scfq.identity()
    acc_h = (ctypes.c_uint64 * scfq.HIST_WORDS)()
    scfq.combine(acc, p1, acc_h, h1)
    scfq.combine(acc, p2, acc_h, h2)
    assert acc.hist_class == 5
    with pytest.raises(scfq.ScfqError) as e:
        scfq.finalize(acc, acc_h)
    assert e.value.rc == scfq.SCFQ_ESPEC
    # the documented recovery: the same shards with the exact kernel
    acc = scfq.identity()
    acc_h = (ctypes.c_uint64 * scfq.HIST_WORDS)()
    for lo, hi in ((0, cut), (cut, b.size)):
        p, h = scfq.partial_device(pb + lo, hi - lo, int(b[lo - 1]) if lo else -1, flags=scfq.SCFQ_QUAL_HIST | scfq.SCFQ_HIST_EXACT, want_hist=True)
        scfq.combine(acc, p, acc_h, h)
    assert list(scfq.finalize(acc, acc_h).qual_hist) == list(oracle.count(b).qual_hist)


def test_streaming_chunks_and_files(gpu, scfq, oracle, tmp_path):
    rng = np.random.default_rng(33)
    a = make_fastq(rng, 50000)
    oc = oracle.count(a)
    for chunk in (4096, 1 << 16, 1 << 20, 0):
        c = scfq.count_host(a, chunk_bytes=chunk, flags=scfq.SCFQ_QUAL_HIST)
        assert list(c.qual_hist) == list(oc.qual_hist), chunk
    path = tmp_path / "x.fq"
    path.write_bytes(a.tobytes())
    c = scfq.count_file(str(path), flags=scfq.SCFQ_QUAL_HIST)
    assert list(c.qual_hist) == list(oc.qual_hist)
    if gpu.cuda.device_count() >= 1:
        c = scfq.count_host(a, devices=[0, 0], flags=scfq.SCFQ_QUAL_HIST, chunk_bytes=1 << 18)    # two shards on one device
        assert list(c.qual_hist) == list(oc.qual_hist)


def test_fq_meta_whole_file_quality_range(gpu, scfq):
    """SCFQ_META_WHOLE_FILE: min_qual / max_qual from K3's histogram of every quality line == the sampled loop over all records"""
    import os
    from conftest import GOLDEN
    for name in ("novaseq.fq", "illumina_3.fq", "sra.fq", "dup.fq.gz", "illumina_8.fq"):
        path = os.path.join(GOLDEN, name)
        full = scfq.meta_file_tsv(path, sample_n=1000).split("\t")
        first = scfq.meta_file_tsv(path, sample_n=1).split("\t")
        whole = scfq.meta_file_tsv(path, sample_n=1, flags=scfq.SCFQ_META_WHOLE_FILE).split("\t")
        assert whole[10:15] == full[10:15], name              # quality columns of the whole file
        assert whole[:10] == first[:10] and whole[15] == "1", name   # everything else still from the first record
