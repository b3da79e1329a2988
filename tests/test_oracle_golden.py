"""The oracle (oracle/: CPU restatement, test infrastructure) pinned against the reference's own known answers:
the 15 rows of the reference's docs/fq-count.md:27-43 over its tests/fastq inputs, plus the build-authored edge
fixtures (parity unpinned by the reference; expected values from tests/golden/make_golden.py's independent
pure-Python restatement)."""
import gzip
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden_rows

ROWS = golden_rows()


def _data(row):
    raw = open(os.path.join(GOLDEN, row["name"]), "rb").read()
    if row["name"] == "edge/two_member.fq.gz" or row["name"] == "dup.fq.gz":
        return raw, gzip.decompress(raw)
    return raw, raw


def test_reference_table_is_complete():
    ref = [r for r in ROWS if r["source"].startswith("reference:")]
    assert len(ref) == 15
    assert sum(1 for r in ROWS if r["source"] == "build:unpinned") >= 20


@pytest.mark.parametrize("row", ROWS, ids=[r["name"] for r in ROWS])
def test_fixture_integrity_and_oracle_counts(oracle, row):
    raw, data = _data(row)
    assert hashlib.sha256(raw).hexdigest() == row["sha256"], "fixture bytes changed"
    for which in ("lines", "bytes"):
        c = oracle.count(np.frombuffer(data, dtype=np.uint8), which)
        assert (c.reads, c.gc_bases, c.n_bases, c.bases) == (row["reads"], row["gc_bases"], row["n_bases"], row["bases"]), which
        assert oracle.tsv(c).split("\t")[1] == row["gc_content"]
        assert oracle.tsv(c) == "%d\t%s\t%d\t%d\t%d" % (row["reads"], row["gc_content"], row["gc_bases"], row["n_bases"], row["bases"])


@pytest.mark.parametrize("row", ROWS, ids=[r["name"] for r in ROWS])
def test_oracle_file_entry(oracle, row):
    """plain and .gz dispatch on the last three bytes of the name (src/fq_count.nim:31), gzread semantics"""
    rc, c = oracle.count_file(os.path.join(GOLDEN, row["name"]))
    assert rc == 0
    assert (c.reads, c.gc_bases, c.n_bases, c.bases) == (row["reads"], row["gc_bases"], row["n_bases"], row["bases"])


def test_oracle_unopenable(oracle, tmp_path):
    rc, _ = oracle.count_file(str(tmp_path / "missing.fq"))
    assert rc == -1
    rc, _ = oracle.count_file(str(tmp_path / "missing.fq.gz"))
    assert rc == -1


def test_quality_range_matches_reference_fq_meta_doc(oracle):
    """docs/fq-meta.md:34-37 (reference) gives min_qual/max_qual (byte-33) for four fixtures: the lowest / highest
    non-empty bin of the K3 quality histogram must agree."""
    expect = {"illumina_2000_2500.fq": (14, 14), "illumina_3000_4000.fq": (14, 14),
              "illumina_6.fq": (0, 37), "illumina_7.fq": (0, 37)}
    for name, (lo, hi) in expect.items():
        rc, c = oracle.count_file(os.path.join(GOLDEN, name))
        bins = [v for v in range(256) if c.qual_hist[v]]
        assert (min(bins) - 33, max(bins) - 33) == (lo, hi), name


def _combine(acc, b):
    """pure-Python statement of the shard monoid on 27-word oracle partials"""
    k = acc[0] & 3
    out = list(acc)
    for arr in (1, 5, 9, 13, 17, 21):
        for r in range(4):
            out[arr + r] = (acc[arr + r] + b[arr + ((r - k) & 3)]) & (2**64 - 1)
    out[0] = acc[0] + b[0]
    out[25] = acc[25] + b[25]
    out[26] = b[26] if b[25] else acc[26]
    return out


def test_partial_monoid_on_random_cuts(oracle):
    rng = np.random.default_rng(5)
    alphabet = np.frombuffer(b"ACGTN@+I\r\n\n", dtype=np.uint8)
    for trial in range(200):
        n = int(rng.integers(0, 400))
        data = rng.choice(alphabet, n).astype(np.uint8)
        whole = oracle.partial(data, -1)
        cuts = sorted(set([0, n] + [int(x) for x in rng.integers(0, n + 1, 4)]))
        acc = [0] * 27
        for a, b in zip(cuts[:-1], cuts[1:]):
            part = oracle.partial(data[a:b], int(data[a - 1]) if a else -1)
            acc = _combine(acc, part)
        assert acc[:26] == whole[:26], (trial, cuts)
        # and the folded partial reproduces the line-loop restatement
        c = oracle.count(data, "lines")
        lines = whole[0] + (1 if n and data[-1] != 10 else 0)
        assert (c.gc_bases, c.n_bases, c.bases, c.lines, c.reads) == (whole[2], whole[6], whole[10], lines, (lines + 3) // 4)
        cb = oracle.count(data, "bytes")
        assert cb.bad_at == whole[13] - whole[17] and cb.bad_plus == whole[15] - whole[23]
