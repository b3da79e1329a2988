"""Device-side BGZF inflate (csrc/bgzf_inflate_kernel.hpp) against zlib: same bytes for valid images, SCFQ_EGZ for corrupt ones."""
import gzip
import struct
import zlib

import numpy as np
import pytest

from test_ingest_sources import bgzf_block, bgzf_file, fastq_bytes

pytestmark = pytest.mark.gpu


def raw_block(data, payload):
    bsize = 18 + len(payload) + 8
    hdr = b"\x1f\x8b\x08\x04" + b"\x00\x00\x00\x00" + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1)
    return hdr + payload + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data))


def deflate(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY):
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
    return co.compress(data) + co.flush()


def test_small_images_first(gpu, scfq):
    """tiny inputs first: a kernel fault shows up here before anything large runs"""
    for data in (b"", b"a", b"hello hello hello hello\n", b"@r\nACGT\n+\nIIII\n" * 10, bytes(1000), bytes(range(256)) * 4):
        for level in (0, 1, 6, 9):
            img = bgzf_block(data, level)
            assert gzip.decompress(img) == data
            assert scfq.debug_bgzf_inflate(img, len(data) + 16) == data, (data[:10], level)


def test_corpora_levels_strategies(gpu, scfq):
    rng = np.random.default_rng(3)
    corpora = {
        "fastq": fastq_bytes(3_000_000, seed=5),
        "random": rng.integers(0, 256, 300_000, dtype=np.uint8).tobytes(),
        "zeros": bytes(400_000),
        "short_period": (b"ACGTN" * 7 + b"\n") * 9_000,
        "two_symbols": bytes(rng.choice(np.frombuffer(b"AB", dtype=np.uint8), 200_000)),
        "skewed": bytes(rng.choice(np.arange(256, dtype=np.uint8), 300_000, p=np.r_[np.full(16, 0.05), np.full(240, 0.2 / 240)])),
        # matches at the far end of the window (zlib reaches back 32506 bytes), long ones (258) and runs (distance 1) side by side
        "far_repeats": rng.integers(0, 256, 32_000, dtype=np.uint8).tobytes() * 6 + bytes(3000) + b"Q" * 700,
    }
    for name, data in corpora.items():
        for level in (1, 6, 9):
            img = b"".join(bgzf_block(data[i:i + 0xff00], level) for i in range(0, len(data), 0xff00)) + bgzf_block(b"")
            assert scfq.debug_bgzf_inflate(img, len(data) + 16) == data, (name, level)
        for strategy in (zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE):
            img = b"".join(raw_block(data[i:i + 60000], deflate(data[i:i + 60000], 6, strategy)) for i in range(0, len(data), 60000))
            assert scfq.debug_bgzf_inflate(img, len(data) + 16) == data, (name, strategy)
        img = bgzf_file(data, block=997)                    # thousands of tiny members
        assert scfq.debug_bgzf_inflate(img, len(data) + 16) == data, (name, "tiny blocks")


def test_corrupt_members_are_rejected(gpu, scfq):
    data = fastq_bytes(500_000, seed=9)
    img = bytearray(bgzf_file(data))
    rng = np.random.default_rng(4)
    rejected = 0
    for trial in range(40):
        bad = bytearray(img)
        pos = int(rng.integers(18, len(bad) - 28))
        bad[pos] ^= 1 << int(rng.integers(0, 8))
        try:
            want = gzip.decompress(bytes(bad))
        except Exception:
            want = None
        try:
            got = scfq.debug_bgzf_inflate(bytes(bad), len(data) + 65536)
        except scfq.ScfqError as e:
            assert e.rc in (scfq.SCFQ_EGZ, scfq.SCFQ_EARG), e.rc      # EARG: the flip hit a header field (no longer BGZF)
            got = None
            rejected += 1
        assert got == want, (trial, pos)
    assert rejected >= 30
    with pytest.raises(scfq.ScfqError):
        scfq.debug_bgzf_inflate(bytes(img[: len(img) // 2]), len(data) + 16)


def test_count_file_with_device_inflate(gpu, scfq, oracle, tmp_path):
    """scfq_count_file on BGZF files: device-side inflate (default) == host inflate == oracle, incl. quality histogram,
    small chunks (members spread over many chunks, look-behind byte carried on the device) and CRLF line ends"""
    import os
    import subprocess
    import sys
    data = fastq_bytes(6_000_000, seed=21).replace(b"\n", b"\r\n")
    a = np.frombuffer(data, dtype=np.uint8)
    oc = oracle.count(a)
    for name, blob in (("x.fq.gz", bgzf_file(data)), ("tiny.fq.gz", bgzf_file(data, block=3001)), ("noeof.fq.gz", bgzf_file(data, eof_marker=False))):
        f = tmp_path / name
        f.write_bytes(blob)
        for chunk in (0, 1 << 20, 1 << 16):
            c = scfq.count_file(str(f), flags=scfq.SCFQ_QUAL_HIST | scfq.SCFQ_STRUCT_CHECK, chunk_bytes=chunk)
            for fld in ("reads", "gc_bases", "n_bases", "bases", "lines", "newlines", "input_bytes", "bad_at", "bad_plus"):
                assert getattr(c, fld) == getattr(oc, fld), (name, chunk, fld)
            assert list(c.qual_hist) == list(oc.qual_hist), (name, chunk)
    # the host path on the same file, in a fresh process (the switch is read once)
    code = ("import sys; sys.path.insert(0, %r); import scfq; c = scfq.count_file(%r); print(c.reads, c.gc_bases, c.n_bases, c.bases)"
            % (os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "seq-collection_amd", "pyhost"), str(tmp_path / "x.fq.gz")))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SCFQ_BGZF_DEVICE="0"), capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.split() == [str(oc.reads), str(oc.gc_bases), str(oc.n_bases), str(oc.bases)], r.stderr
    # a corrupt member is an error, not a wrong count
    bad = bytearray(bgzf_file(data))
    bad[len(bad) // 2] ^= 0x10
    g = tmp_path / "bad.fq.gz"
    g.write_bytes(bytes(bad))
    with pytest.raises(scfq.ScfqError) as e:
        scfq.count_file(str(g))
    assert e.value.rc == scfq.SCFQ_EGZ


def test_dedup_of_a_bgzf_file_is_staged_by_the_device_inflate(gpu, scfq, oracle, tmp_path):
    data = fastq_bytes(3_000_000, seed=31)
    data = data + data[: len(data) // 3]                 # the first third again: duplicates (cut mid-record on purpose)
    want, ost = oracle.dedup(np.frombuffer(data, dtype=np.uint8))
    for name, blob in (("d.fq.gz", bgzf_file(data)), ("d_tiny.fq.gz", bgzf_file(data, block=2000))):
        f = tmp_path / name
        f.write_bytes(blob)
        out = tmp_path / "out.fq"
        with open(out, "wb") as fh:
            st = scfq.dedup_file(str(f), fh.fileno())
        assert out.read_bytes() == want and (st.duplicates, st.total_reads) == (ost.duplicates, ost.total_reads), name
    bad = bytearray(bgzf_file(data))
    bad[len(bad) // 2] ^= 4
    g = tmp_path / "bad.fq.gz"
    g.write_bytes(bytes(bad))
    with pytest.raises(scfq.ScfqError) as e:
        scfq.dedup_file(str(g), -1)
    assert e.value.rc == scfq.SCFQ_EGZ


def test_file_that_stops_being_bgzf_restarts_on_the_host_path(gpu, scfq, oracle, tmp_path):
    """The device path no longer walks every member header before it starts: a member it cannot take (plain gzip, a
    member larger than 64 KiB, a truncated tail) is found on the way, after chunks were already counted on the device;
    the count must restart on the host path and come out as if the device path had never run."""
    d1, d2 = fastq_bytes(4_000_000, seed=31), fastq_bytes(700_000, seed=32)
    big = zlib.compressobj(6, zlib.DEFLATED, 31)
    big_member = big.compress(d2) + big.flush()                                   # one plain gzip member of 700 KB
    cases = {
        "gzip_member_in_the_middle.fq.gz": (bgzf_file(d1, eof_marker=False) + gzip.compress(d2) + bgzf_file(d1[:300_000]), d1 + d2 + d1[:300_000]),
        "gzip_member_at_the_end.fq.gz": (bgzf_file(d1, eof_marker=False) + big_member, d1 + d2),
    }
    for name, (blob, data) in cases.items():
        f = tmp_path / name
        f.write_bytes(blob)
        oc = oracle.count(np.frombuffer(data, dtype=np.uint8))
        c = scfq.count_file(str(f), flags=scfq.SCFQ_QUAL_HIST)
        for fld in ("reads", "gc_bases", "n_bases", "bases", "lines", "newlines", "input_bytes"):
            assert getattr(c, fld) == getattr(oc, fld), (name, fld)
        assert list(c.qual_hist) == list(oc.qual_hist), name
    # a truncated last member: an error on either path, never a count
    g = tmp_path / "cut.fq.gz"
    g.write_bytes(bgzf_file(d1)[:-40])
    with pytest.raises(scfq.ScfqError):
        scfq.count_file(str(g))


def test_the_other_symbol_loops(gpu, tmp_path):
    """the default is the boundary-first loop (symbol_loop_dense); the lane-parallel loop of r2 and the serial one of r1 stay selectable
    (SCFQ_INFLATE_LOOP, read once per process) and must put out the same bytes: BGZF members and an ordinary gzip member, each in a
    process of its own"""
    import os, subprocess, sys
    data = (fastq_bytes(6_000_000, seed=21) + b"@long\n" + b"A" * 70_000 + b"\n+\n" + b"F" * 70_000 + b"\n"
            + b"".join(b"@p%d\n" % i + b"ACGTN" * 9 + b"\n+\n" + b"FF:,#" * 9 + b"\n" for i in range(3000)))
    (tmp_path / "b.fq.gz").write_bytes(bgzf_file(data))
    (tmp_path / "g.fq.gz").write_bytes(gzip.compress(data, 6))
    code = """
import sys, hashlib
sys.path.insert(0, sys.argv[1])
import scfq
img = open(sys.argv[2], "rb").read()
out = scfq.debug_bgzf_inflate(img, int(sys.argv[4]) + 16)
c = scfq.count_file(sys.argv[3])
print(hashlib.sha256(bytes(out)).hexdigest(), c.reads, c.gc_bases, c.n_bases, c.bases, c.input_bytes)
"""
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "seq-collection_amd", "pyhost")
    import hashlib
    seen = {}
    for loop in ("dense", "lanes", "serial"):
        r = subprocess.run([sys.executable, "-c", code, here, str(tmp_path / "b.fq.gz"), str(tmp_path / "g.fq.gz"), str(len(data))],
                           env=dict(os.environ, SCFQ_INFLATE_LOOP=loop), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (loop, r.stderr[-2000:])
        seen[loop] = r.stdout.split()
    assert seen["dense"][0] == hashlib.sha256(data).hexdigest()
    assert int(seen["dense"][5]) == len(data)
    assert seen["lanes"] == seen["dense"] and seen["serial"] == seen["dense"]
