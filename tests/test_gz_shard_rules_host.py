"""The two host-side rules of the gzip-member shards (scfq_count_file_sharded on an ordinary .gz of several members; the shards
themselves run under -m gpu in tests/test_gpu_comm.py): where a rank cuts the file, and how a shard that was scanned as if it began
the input is put right once the byte in front of it is known.  Checked against zlib and the oracle's partials; no device."""
import ctypes
import gzip
import zlib

import numpy as np
import pytest

from test_ingest_sources import fastq_bytes


def _member(b, level=6):
    co = zlib.compressobj(level, zlib.DEFLATED, 31)
    return co.compress(b) + co.flush()


def _boundary(scfq, path, at):
    L = scfq.lib()
    L.scfq_debug_gz_member_boundary.restype = ctypes.c_int64
    L.scfq_debug_gz_member_boundary.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.POINTER(ctypes.c_int)]
    fb = ctypes.c_int(-7)
    r = L.scfq_debug_gz_member_boundary(str(path).encode(), at, ctypes.byref(fb))
    return r, fb.value


def test_member_boundaries_are_the_true_member_starts(scfq, tmp_path):
    data = fastq_bytes(3_000_000, seed=4)
    rng = np.random.default_rng(1)
    cuts = sorted(set(int(x) for x in rng.integers(1, len(data) - 1, 12)))
    parts = [data[a:b] for a, b in zip([0] + cuts, cuts + [len(data)])]
    members = [_member(p) for p in parts]
    members.insert(5, _member(b""))                       # an empty member: the first byte of a cut there is the next member's
    parts.insert(5, b"")
    blob = b"".join(members)
    f = tmp_path / "m.fq.gz"
    f.write_bytes(blob + b"\0garbage")
    assert gzip.decompress(blob) == data
    starts = np.cumsum([0] + [len(m) for m in members]).tolist()
    # from every byte offset of a coarse grid, and around every true start: the answer is the next true start, never a byte
    # pattern inside the deflate data (the magic 1f 8b 08 turns up about once per 16 MB of compressed bytes; the verification
    # by header + inflate is what the grid exercises on whatever candidates this file holds)
    probes = set(range(0, len(blob), 7919)) | {s + d for s in starts[:-1] for d in (-1, 0, 1) if 0 <= s + d < len(blob)}
    for at in sorted(probes):
        want = next((s for s in starts[:-1] if s >= at), len(blob) + len(b"\0garbage"))
        got, fb = _boundary(scfq, f, at)
        assert got == want, (at, got, want)
        if want < len(blob):
            k = starts.index(want)
            first = next((p[0] for p in parts[k:] if p), -1)
            assert fb == first, (at, fb, first)
    # bytes that spell a member header but are not one: a stored block that carries "1f 8b 08 00 ..." followed by text
    fake = b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03" + b"@not deflate data at all, just text that follows a header-shaped run of bytes\n" * 4
    co = zlib.compressobj(0, zlib.DEFLATED, 31)
    stored = co.compress(b"@r\nAC\n+\n!!\n" + fake + b"ACGT\n" * 50) + co.flush()
    g = tmp_path / "fake.fq.gz"
    g.write_bytes(stored + _member(b"@x\nA\n+\n!\n"))
    assert fake in stored
    got, fb = _boundary(scfq, g, stored.index(fake))
    assert got == len(stored) and fb == ord("@")          # skipped: the next TRUE member is the answer


@pytest.mark.parametrize("flags_name", ["plain", "struct", "hist", "hist_struct"])
def test_shard_fix_equals_a_scan_that_knew_the_byte_before(scfq, oracle, flags_name):
    """oracle partial of a shard scanned with prev = -1, fixed with the true byte == oracle partial scanned with the true byte;
    over every combination of the byte before (\\r, \\n, a letter) and the shard's first byte (\\n, @, +, a letter)"""
    flags = {"plain": 0, "struct": scfq.SCFQ_STRUCT_CHECK, "hist": scfq.SCFQ_QUAL_HIST,
             "hist_struct": scfq.SCFQ_QUAL_HIST | scfq.SCFQ_STRUCT_CHECK}[flags_name]
    want_hist = bool(flags & scfq.SCFQ_QUAL_HIST)
    L = scfq.lib()
    body = b"ACGTN\r\n+\r\nFF:#,\r\n@r2 x\nGGCC\n+\n@+FF\n@r3\nAC"
    for prev in (13, 10, 65):
        for first in (b"\n", b"@", b"+", b"G", b"\r"):
            shard = np.frombuffer(first + body, dtype=np.uint8)
            r_true = oracle.partial(shard, prev, want_hist=want_hist)
            r_assumed = oracle.partial(shard, -1, want_hist=want_hist)
            w_true, h_true = r_true if want_hist else (r_true, None)
            w_as, h_as = r_assumed if want_hist else (r_assumed, None)
            p = scfq.Partial.from_words(list(w_as) + [0] * 5)
            if not (flags & scfq.SCFQ_STRUCT_CHECK):      # a scan without the structure check leaves the line-start words at zero
                for k in range(4):
                    p.starts[k] = p.first_at[k] = p.first_plus[k] = 0
            h = (ctypes.c_uint64 * scfq.HIST_WORDS)(*h_as) if want_hist else None
            assert L.scfq_debug_gz_shard_fix(ctypes.byref(p), ctypes.byref(h) if want_hist else None, prev, shard[0].item(), flags) == 0
            got = p.words()[:27]
            exp = list(w_true)
            if not (flags & scfq.SCFQ_STRUCT_CHECK):
                exp[13:25] = [0] * 12
            assert got == exp, (flags_name, prev, first, got, exp)
            if want_hist:
                assert list(h) == list(h_true), (flags_name, prev, first)
    # nothing in front (the shard does begin the input), or an empty shard: untouched
    shard = np.frombuffer(b"\n" + body, dtype=np.uint8)
    w = oracle.partial(shard, -1)
    p = scfq.Partial.from_words(list(w) + [0] * 5)
    L.scfq_debug_gz_shard_fix(ctypes.byref(p), None, -1, 10, scfq.SCFQ_STRUCT_CHECK)
    assert p.words()[:27] == list(w)
