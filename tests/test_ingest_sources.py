"""Host-side byte sources of scfq_count_file (no device): plain pread (multi-threaded), serial gzread, and the
block-parallel BGZF inflate must all yield exactly what zlib's gzread yields (gzip_stream.nim:16-17 semantics:
concatenated members, trailing garbage ignored, non-gzip bytes passed through)."""
import gzip
import os
import struct
import zlib

import numpy as np
import pytest


def bgzf_block(data: bytes, level=6) -> bytes:
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    payload = co.compress(data) + co.flush()
    bsize = 18 + len(payload) + 8
    assert bsize <= 65536
    hdr = b"\x1f\x8b\x08\x04" + b"\x00\x00\x00\x00" + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1)
    return hdr + payload + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data))


def bgzf_file(data: bytes, block=0xff00, eof_marker=True) -> bytes:
    out = b"".join(bgzf_block(data[i:i + block]) for i in range(0, len(data), block))
    return out + (bgzf_block(b"") if eof_marker else b"")


def fastq_bytes(n, seed=1):
    rng = np.random.default_rng(seed)
    rec = []
    size = 0
    i = 0
    while size < n:
        L = int(rng.integers(20, 300))
        r = b"@r%d\n" % i + bytes(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), L)) + b"\n+\n" + bytes(rng.choice(np.frombuffer(b"FFFF:,#", dtype=np.uint8), L)) + b"\n"
        rec.append(r)
        size += len(r)
        i += 1
    return b"".join(rec)


@pytest.mark.parametrize("chunk", [4096, 100_000, 1 << 20, 0])
def test_bgzf_parallel_inflate_equals_gzread(scfq, tmp_path, chunk):
    data = fastq_bytes(3_000_000)
    f = tmp_path / "x.fq.gz"
    f.write_bytes(bgzf_file(data))
    assert gzip.decompress(f.read_bytes()) == data           # python's zlib agrees it is a valid multi-member gzip
    assert scfq.debug_read_file(str(f), len(data) + 10, chunk) == data
    os.environ["SCFQ_NO_BGZF"] = "1"                         # serial gzread path on the same file
    try:
        assert scfq.debug_read_file(str(f), len(data) + 10, chunk) == data
    finally:
        del os.environ["SCFQ_NO_BGZF"]


def test_bgzf_edge_layouts(scfq, tmp_path):
    data = fastq_bytes(400_000, seed=2)
    a, b = data[:150_000], data[150_000:]
    cases = {
        "no_eof_marker.fq.gz": (bgzf_file(data, eof_marker=False), data),
        "tiny_blocks.fq.gz": (bgzf_file(data, block=997), data),
        "bgzf_then_plain_member.fq.gz": (bgzf_file(a, eof_marker=False) + gzip.compress(b, mtime=0), data),
        "plain_then_bgzf.fq.gz": (gzip.compress(a, mtime=0) + bgzf_file(b), data),
        "trailing_garbage.fq.gz": (bgzf_file(data) + b"\x00\x00garbage", data),
        "empty_bgzf.fq.gz": (bgzf_block(b""), b""),
        "stored_blocks.fq.gz": (b"".join(bgzf_block(data[i:i + 60000], level=0) for i in range(0, len(data), 60000)), data),
    }
    for name, (blob, expect) in cases.items():
        f = tmp_path / name
        f.write_bytes(blob)
        for chunk in (8192, 0):
            assert scfq.debug_read_file(str(f), len(expect) + 10, chunk) == expect, (name, chunk)


def test_bgzf_corruption_is_reported(scfq, tmp_path):
    data = fastq_bytes(300_000, seed=3)
    blob = bytearray(bgzf_file(data))
    f = tmp_path / "trunc.fq.gz"
    f.write_bytes(bytes(blob[: len(blob) // 2]))
    with pytest.raises(scfq.ScfqError) as e:
        scfq.debug_read_file(str(f), len(data) + 10)
    assert e.value.rc == scfq.SCFQ_EGZ
    blob[len(blob) // 3] ^= 0x55                             # flip a payload byte: CRC / inflate must notice
    f2 = tmp_path / "flip.fq.gz"
    f2.write_bytes(bytes(blob))
    with pytest.raises(scfq.ScfqError) as e:
        scfq.debug_read_file(str(f2), len(data) + 10)
    assert e.value.rc == scfq.SCFQ_EGZ


def test_plain_and_serial_gz_sources(scfq, tmp_path):
    data = fastq_bytes(9_000_000, seed=4)
    p = tmp_path / "plain.fq"
    p.write_bytes(data)
    assert scfq.debug_read_file(str(p), len(data), 1 << 20) == data           # multi-threaded pread pieces
    assert scfq.debug_read_file(str(p), len(data), 0) == data
    g = tmp_path / "two.fq.gz"
    g.write_bytes(gzip.compress(data[:4_000_000], mtime=0) + gzip.compress(data[4_000_000:], mtime=0))
    assert scfq.debug_read_file(str(g), len(data), 1 << 20) == data
    ng = tmp_path / "not_gzip.fq.gz"
    ng.write_bytes(data[:100_000])
    assert scfq.debug_read_file(str(ng), 100_000) == data[:100_000]           # gzread passes non-gzip bytes through
    with pytest.raises(scfq.ScfqError) as e:
        scfq.debug_read_file(str(tmp_path / "missing.fq"), 10)
    assert e.value.rc == scfq.SCFQ_EOPEN


@pytest.mark.gpu
def test_bgzf_file_counts_on_gpu(gpu, scfq, oracle, tmp_path):
    data = fastq_bytes(5_000_000, seed=5)
    f = tmp_path / "x.fq.gz"
    f.write_bytes(bgzf_file(data))
    c = scfq.count_file(str(f))
    oc = oracle.count(np.frombuffer(data, dtype=np.uint8), "lines")
    assert (c.reads, c.gc_bases, c.n_bases, c.bases, c.lines) == (oc.reads, oc.gc_bases, oc.n_bases, oc.bases, oc.lines)
