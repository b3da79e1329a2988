"""The library's own DEFLATE decoder (csrc/scfq_inflate.hpp, used for regular gzip files and BGZF blocks) against zlib,
which is what the reference reads .gz with (src/fq_count.nim:32; gzip_stream.nim:16-17): same bytes for every valid stream,
an error for every stream zlib rejects. Host only (scfq_debug_read_file), no device."""
import gzip
import os
import struct
import subprocess
import sys
import zlib

import numpy as np
import pytest

from test_ingest_sources import fastq_bytes

HERE = os.path.dirname(os.path.abspath(__file__))


def read(scfq, tmp_path, blob, cap, chunk=0, name="x.fq.gz"):
    f = tmp_path / name
    f.write_bytes(blob)
    return scfq.debug_read_file(str(f), cap, chunk)


def raw_deflate(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, wbits=-15):
    co = zlib.compressobj(level, zlib.DEFLATED, wbits, 9, strategy)
    return co.compress(data) + co.flush()


def gz_member(data, payload=None, flg=0, extra=b"", name=b"", comment=b""):
    payload = raw_deflate(data) if payload is None else payload
    h = b"\x1f\x8b\x08" + bytes([flg]) + b"\x00\x00\x00\x00\x00\x03"
    if flg & 4:
        h += struct.pack("<H", len(extra)) + extra
    if flg & 8:
        h += name + b"\x00"
    if flg & 16:
        h += comment + b"\x00"
    if flg & 2:
        h += struct.pack("<H", zlib.crc32(h) & 0xFFFF)
    return h + payload + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data) & 0xFFFFFFFF)


def corpora():
    rng = np.random.default_rng(7)
    fq = fastq_bytes(2_500_000, seed=11)
    return {
        "fastq": fq,
        "random": rng.integers(0, 256, 700_000, dtype=np.uint8).tobytes(),            # incompressible: stored blocks
        "zeros": bytes(900_000),                                                      # distance-1 matches of length 258
        "short_period": (b"ACGTN" * 7 + b"\n") * 30_000,                              # distances 2..7 and long matches
        "two_symbols": bytes(rng.choice(np.frombuffer(b"AB", dtype=np.uint8), 400_000)),
        "many_symbols": bytes(rng.choice(np.arange(256, dtype=np.uint8), 600_000, p=np.r_[np.full(16, 0.05), np.full(240, 0.2 / 240)])),
        "tiny": b"@r\nACGT\n+\nIIII\n",
        "one_byte": b"x",
        "empty": b"",
    }


@pytest.mark.parametrize("level", [1, 6, 9])
def test_levels_and_chunks(scfq, tmp_path, level):
    for name, data in corpora().items():
        blob = gzip.compress(data, compresslevel=level, mtime=0)
        for chunk in (1 << 16, 1 << 20, 0):
            assert read(scfq, tmp_path, blob, len(data) + 16, chunk) == data, (name, level, chunk)


def test_strategies_fixed_huffman_and_stored(scfq, tmp_path):
    for name, data in corpora().items():
        for strategy in (zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED):
            blob = gz_member(data, raw_deflate(data, 6, strategy))
            assert read(scfq, tmp_path, blob, len(data) + 16, 1 << 16) == data, (name, strategy)
        blob = gz_member(data, raw_deflate(data, 0))                                  # stored blocks only
        assert read(scfq, tmp_path, blob, len(data) + 16, 1 << 16) == data, (name, "stored")
        small_window = gz_member(data, raw_deflate(data, 6, wbits=-9))                # 512-byte window: short distances only
        assert read(scfq, tmp_path, small_window, len(data) + 16, 1 << 16) == data, (name, "wbits 9")


def test_member_framing(scfq, tmp_path):
    c = corpora()
    a, b, d = c["fastq"][:700_000], c["short_period"][:300_000], c["random"][:100_000]
    # header flags: FEXTRA, FNAME, FCOMMENT, FHCRC, all at once
    for flg in (4, 8, 16, 2, 4 | 8 | 16 | 2):
        blob = gz_member(a, flg=flg, extra=b"AB\x02\x00xy", name=b"reads.fq", comment=b"a comment")
        assert gzip.decompress(blob) == a
        assert read(scfq, tmp_path, blob, len(a) + 16, 1 << 16) == a, flg
    # concatenated members, empty members in between, trailing garbage after the last one (zlib ignores it)
    blob = gz_member(a) + gz_member(b"") + gz_member(b, flg=8, name=b"n") + gz_member(d) + gz_member(b"")
    assert read(scfq, tmp_path, blob, len(a + b + d) + 16, 1 << 16) == a + b + d
    assert read(scfq, tmp_path, blob + b"\x00\x00\x00garbage that is not a member", len(a + b + d) + 16, 1 << 17) == a + b + d
    assert read(scfq, tmp_path, blob + b"\x1f", len(a + b + d) + 16, 1 << 17) == a + b + d
    # a member boundary exactly at / near a chunk boundary
    for cut in (65536 - 300, 65536 - 274, 65536 - 1, 65536, 65537, 131072):
        x = c["fastq"][:cut]
        blob = gz_member(x) + gz_member(a[:50_000])
        assert read(scfq, tmp_path, blob, cut + 50_016, 1 << 16) == x + a[:50_000], cut


def test_corrupt_streams_are_rejected_like_zlib(scfq, tmp_path):
    data = corpora()["fastq"][:600_000]
    good = gzip.compress(data, mtime=0)

    def expect_error(blob, what):
        with pytest.raises((zlib.error, EOFError, OSError, gzip.BadGzipFile)):
            gzip.decompress(blob)                                  # python's zlib rejects it too
        with pytest.raises(scfq.ScfqError) as e:
            read(scfq, tmp_path, blob, len(data) + 16, 1 << 16)
        assert e.value.rc == scfq.SCFQ_EGZ, what

    expect_error(good[: len(good) // 2], "truncated in the deflate stream")
    expect_error(good[:-3], "truncated trailer")
    bad_crc = bytearray(good); bad_crc[-8] ^= 1
    expect_error(bytes(bad_crc), "CRC-32")
    bad_len = bytearray(good); bad_len[-1] ^= 0x40
    expect_error(bytes(bad_len), "ISIZE")
    expect_error(good + b"\x1f\x8b\x07\x00" + bytes(20), "second member with an unknown method")
    expect_error(good + b"\x1f\x8b\x08\xe0" + bytes(20), "second member with reserved flags")
    expect_error(b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03" + b"\x07" + bytes(30), "reserved block type 3")
    expect_error(gz_member(data, b"\x01\x05\x00\xfa\xfe" + b"hello"), "stored block with LEN != ~NLEN")
    # distance beyond the start of the stream: fixed-Huffman block, <length 3, distance 1> with no byte produced yet
    expect_error(gz_member(b"", bytes([0b00000011, 0b00000010, 0b00000000, 0]) + bytes(8)), "distance too far back")


def test_bit_flips_never_crash_and_never_pass_silently(scfq, tmp_path):
    """every corruption either fails (as it does in zlib: CRC-32 is the backstop) or leaves the output unchanged"""
    rng = np.random.default_rng(99)
    data = corpora()["fastq"][:300_000]
    for level in (1, 9):
        good = bytearray(gzip.compress(data, compresslevel=level, mtime=0))
        for trial in range(60):
            blob = bytearray(good)
            pos = int(rng.integers(10, len(blob)))
            blob[pos] ^= 1 << int(rng.integers(0, 8))
            try:
                want = gzip.decompress(bytes(blob))
            except Exception:
                want = None
            try:
                got = read(scfq, tmp_path, bytes(blob), len(data) + 65536, 1 << 16)
            except scfq.ScfqError as e:
                assert e.rc == scfq.SCFQ_EGZ
                got = None
            assert got == want, (level, trial, pos)


def test_zlib_switch_gives_the_same_bytes(scfq, tmp_path):
    data = corpora()["fastq"]
    f = tmp_path / "x.fq.gz"
    f.write_bytes(gzip.compress(data, mtime=0))
    code = ("import sys; sys.path.insert(0, %r); import scfq, hashlib; "
            "print(hashlib.sha256(scfq.debug_read_file(%r, %d, 1 << 20)).hexdigest())") % (
        os.path.join(os.path.dirname(HERE), "seq-collection_amd", "pyhost"), str(f), len(data) + 16)
    import hashlib
    for mode in ("zlib", "own"):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SCFQ_INFLATE=mode), capture_output=True, text=True)
        assert r.returncode == 0 and r.stdout.strip() == hashlib.sha256(data).hexdigest(), (mode, r.stderr)


def pgz_env(**kw):
    """the parallel single-member reader (csrc/scfq_pgz.hpp) on test-sized files: tiny segments, no minimum file size"""
    env = {"SCFQ_PGZ_MIN_MB": "0", "SCFQ_PGZ_SEGMENT_MB": "1"}
    env.update({k: str(v) for k, v in kw.items()})
    return env


class EnvPatch:
    def __init__(self, env):
        self.env = env

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.env}
        os.environ.update(self.env)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_parallel_single_member_equals_zlib(scfq, tmp_path):
    """many threads on ONE member: sync search, marker symbols, window resolution, CRC by pieces"""
    rng = np.random.default_rng(17)
    big = fastq_bytes(40_000_000, seed=3)
    cases = {
        "fastq level 6": gzip.compress(big, compresslevel=6, mtime=0),
        "fastq level 1": gzip.compress(big, compresslevel=1, mtime=0),
        "fastq level 9": gzip.compress(big[:12_000_000], compresslevel=9, mtime=0),
        "long matches": gzip.compress((b"ACGTN" * 7 + b"\n") * 400_000, mtime=0),
        "zeros": gzip.compress(bytes(30_000_000), mtime=0),
        "stored (incompressible)": gzip.compress(rng.integers(0, 256, 6_000_000, dtype=np.uint8).tobytes(), mtime=0),
        "fixed huffman": gz_member(big[:5_000_000], raw_deflate(big[:5_000_000], 6, zlib.Z_FIXED)),
        "members": gzip.compress(big[:9_000_000], mtime=0) + gzip.compress(big[9_000_000:20_000_000], compresslevel=1, mtime=0)
                   + gz_member(b"") + gzip.compress(big[20_000_000:], mtime=0) + b"trailing garbage",
    }
    with EnvPatch(pgz_env()):
        for name, blob in cases.items():
            want = gzip.decompress(blob[:-16] if name == "members" else blob) if name != "members" else big
            for chunk in (1 << 16, 1 << 22, 0):
                assert read(scfq, tmp_path, blob, len(want) + 16, chunk) == want, (name, chunk)
    with EnvPatch(pgz_env(SCFQ_PGZ_SEGMENT_MB=3)):
        assert read(scfq, tmp_path, cases["fastq level 6"], len(big) + 16, 1 << 20) == big


def test_parallel_reader_rejects_what_zlib_rejects(scfq, tmp_path):
    data = fastq_bytes(12_000_000, seed=8)
    good = bytearray(gzip.compress(data, mtime=0))
    rng = np.random.default_rng(5)
    with EnvPatch(pgz_env()):
        for trial in range(25):
            blob = bytearray(good)
            pos = int(rng.integers(10, len(blob)))
            blob[pos] ^= 1 << int(rng.integers(0, 8))
            try:
                want = gzip.decompress(bytes(blob))
            except Exception:
                want = None
            try:
                got = read(scfq, tmp_path, bytes(blob), len(data) + 65536, 1 << 20)
            except scfq.ScfqError as e:
                assert e.rc == scfq.SCFQ_EGZ
                got = None
            assert got == want, (trial, pos)
        for cut in (len(good) // 3, len(good) - 5):
            with pytest.raises(scfq.ScfqError):
                read(scfq, tmp_path, bytes(good[:cut]), len(data) + 16, 1 << 20)


def test_parallel_reader_false_syncs_and_thread_counts(scfq, tmp_path):
    """(a) a stored (level 0) member whose payload is itself a deflate stream: every sync search lands on a REAL block header
    that belongs to the inner stream, not to the member being read — the chain check (the segment before must end exactly
    there) has to throw those away; (b) odd thread counts and segment sizes, in fresh processes (the thread count is read once)"""
    inner_plain = fastq_bytes(14_000_000, seed=13)
    inner = gzip.compress(inner_plain, compresslevel=6, mtime=0)           # ~3.5 MB of dynamic blocks
    payload = inner * 4 + inner_plain[:3_000_000]
    outer = gz_member(payload, raw_deflate(payload, 0))                     # stored blocks only
    assert gzip.decompress(outer) == payload
    with EnvPatch(pgz_env()):
        assert read(scfq, tmp_path, outer, len(payload) + 16, 1 << 20) == payload
    mixed = gzip.compress(inner_plain[:6_000_000], mtime=0)[:-8]            # a member cut before its trailer ...
    data = fastq_bytes(20_000_000, seed=14)
    f = tmp_path / "t.fq.gz"
    f.write_bytes(gzip.compress(data, compresslevel=6, mtime=0))
    code = ("import sys; sys.path.insert(0, %r); import scfq, hashlib; "
            "print(hashlib.sha256(scfq.debug_read_file(%r, %d, 1 << 21)).hexdigest())") % (
        os.path.join(os.path.dirname(HERE), "seq-collection_amd", "pyhost"), str(f), len(data) + 16)
    import hashlib
    want = hashlib.sha256(data).hexdigest()
    for threads, seg in ((2, 1), (7, 1), (5, 2), (16, 1)):
        env = dict(os.environ, **pgz_env(SCFQ_INFLATE_THREADS=threads, SCFQ_PGZ_SEGMENT_MB=seg))
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        assert r.returncode == 0 and r.stdout.strip() == want, (threads, seg, r.stderr[-300:])


def test_parallel_reader_hands_many_small_members_to_the_serial_reader(scfq, tmp_path):
    """a BGZF-like file read as plain gzip (thousands of 64 KiB members) must not start a thread batch per member"""
    import time
    from test_ingest_sources import bgzf_file
    data = fastq_bytes(24_000_000, seed=23)
    blob = bgzf_file(data)                                      # ~370 members
    big_then_small = gzip.compress(data[:16_000_000], mtime=0) + bgzf_file(data[16_000_000:])
    with EnvPatch(dict(pgz_env(), SCFQ_NO_BGZF="1")):
        t = time.time()
        assert read(scfq, tmp_path, blob, len(data) + 16, 1 << 20) == data
        assert read(scfq, tmp_path, big_then_small, len(data) + 16, 1 << 20) == data
        assert time.time() - t < 20


def _gzm(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY):
    co = zlib.compressobj(level, zlib.DEFLATED, 31, 8, strategy)
    return co.compress(data) + co.flush()


def test_take_over_in_the_middle_of_a_member(scfq, tmp_path):
    """scfq_gzfast::Resume — the decoder the device gzip path hands the REST of a file to when a batch has no room: it starts at an
    exact bit inside a member with the 32 KiB in front of it and the member's CRC-32 / length so far.  Here without a device: the
    serial decoder reads the first member up to a block boundary, Resume the rest; the bytes must be zlib's whatever the cut, the
    layout behind it and the state of the trailer."""
    data = fastq_bytes(2_500_000, seed=91)
    rng = np.random.default_rng(92)
    cases = {
        "one_member": _gzm(data, 6),
        "level1": _gzm(data, 1),
        "stored_and_fixed": _gzm(data[:900_000], 0) + _gzm(data[900_000:], 6, zlib.Z_FIXED),
        "four_members_garbage": b"".join(_gzm(data[a:b], lvl) for (a, b), lvl in zip(((0, 700_001), (700_001, 700_040), (700_040, 1_900_000), (1_900_000, len(data))), (6, 9, 1, 6))) + b"\x00junk",
        "empty_members": _gzm(data[:1_000_000]) + _gzm(b"") + _gzm(data[1_000_000:]) + _gzm(b""),
        "random_bytes": _gzm(rng.integers(0, 256, 600_000, dtype=np.uint8).tobytes() + data[:400_000]),
    }
    for name, img in cases.items():
        f = tmp_path / (name + ".fq.gz")
        f.write_bytes(img)
        want = gzip.decompress(img[:-5]) if name == "four_members_garbage" else gzip.decompress(img)
        for after in (0, 1, 40_000, 333_333, len(want) // 2, len(want) - 10, 10 * len(want)):
            for chunk in (1 << 16, 1 << 20):
                got = scfq.debug_gz_resume(str(f), after, len(want) + 16, chunk)
                assert got == want, (name, after, chunk, len(got), len(want))
    # damage behind the cut: the take-over rejects what zlib rejects (trailer bits, a flipped bit in the data, a cut file)
    img = bytearray(cases["one_member"])
    bad = {"crc": bytes(img[:-6]) + bytes([img[-6] ^ 0x40]) + bytes(img[-5:]), "isize": bytes(img[:-2]) + bytes([img[-2] ^ 1]) + bytes(img[-1:]),
           "cut": bytes(img[: len(img) * 3 // 4]), "no_trailer": bytes(img[:-8])}
    flip = bytearray(img)
    flip[len(flip) * 2 // 3] ^= 0x10
    bad["flip"] = bytes(flip)
    for name, raw in bad.items():
        f = tmp_path / (name + ".fq.gz")
        f.write_bytes(raw)
        try:
            gzip.decompress(raw)
            zlib_ok = True
        except Exception:      # noqa: BLE001
            zlib_ok = False
        for after in (0, 100_000):
            try:
                scfq.debug_gz_resume(str(f), after, len(data) + 16)
                ok = True
            except scfq.ScfqError as e:
                assert e.rc == scfq.SCFQ_EGZ, (name, e)
                ok = False
            assert ok == zlib_ok, (name, after)
