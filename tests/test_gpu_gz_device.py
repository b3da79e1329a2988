"""Device-side inflate of ordinary gzip members (csrc/gz_inflate_kernels.hpp, csrc/scfq_gzdev.hpp): the row `sc fq-count`
prints must be the oracle's on the bytes zlib inflates, for every stream layout; damaged files must behave exactly as on the
host path (which is gzread, byte for byte).  Reference: src/fq_count.nim:30-34, gzip_stream.nim:16-17.
Runs the CLI in a subprocess so that the knobs (small segments: many of them even for a few MB) apply."""
import gzip
import os
import subprocess
import zlib

import numpy as np
import pytest

from conftest import PKG
from test_ingest_sources import fastq_bytes

pytestmark = pytest.mark.gpu

SC = os.path.join(PKG, "sc")
DEV_ENV = {"SCFQ_GZ_DEVICE_MIN_MB": "0", "SCFQ_GZ_DEVICE_SEGMENT_KB": "32", "SCFQ_VERBOSE": "1"}
# the same with 8 segments per batch: a few MB cross a dozen batch borders (window, look-behind byte, CRC parts and the
# chain position are carried from batch to batch; three batches are in flight at once), window chains in groups of 3 entries
BATCH_ENV = dict(DEV_ENV, SCFQ_GZ_DEVICE_BATCH_SEGMENTS="8", SCFQ_GZ_DEVICE_CHAIN_GROUP="3")
# one batch, its chain of ~50-200 entries walked in groups of 5 (maps, group windows, entry windows)
GROUP_ENV = dict(DEV_ENV, SCFQ_GZ_DEVICE_CHAIN_GROUP="5")


def run(path, **env):
    return subprocess.run([SC, "fq-count", str(path)], capture_output=True, text=True, env=dict(os.environ, **env), timeout=600)


def member(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY):
    co = zlib.compressobj(level, zlib.DEFLATED, 31, 9, strategy)
    return co.compress(data) + co.flush()


def check(oracle, path, data, expect_device=True):
    want = oracle.tsv(oracle.count(np.frombuffer(data, dtype=np.uint8))) + "\n"
    for env in (BATCH_ENV, GROUP_ENV, DEV_ENV):
        r = run(path, **env)
        assert r.returncode == 0, r.stderr[-2000:]
        assert r.stdout == want, r.stderr[-2000:]
        on_device = "on the chain" in r.stderr
        assert on_device == expect_device, r.stderr[-3000:]
        if on_device and env is BATCH_ENV and len(data) > 2_000_000:
            assert " 1 batch(es)" not in r.stderr and "batch(es)" in r.stderr, r.stderr[-1000:]
    return r


def test_levels_and_strategies(gpu, oracle, tmp_path):
    data = fastq_bytes(6_000_000, seed=31)
    for level in (1, 6, 9):
        f = tmp_path / ("l%d.fq.gz" % level)
        f.write_bytes(member(data, level))
        check(oracle, f, data)
    for name, strategy in (("fixed", zlib.Z_FIXED), ("huffman", zlib.Z_HUFFMAN_ONLY), ("rle", zlib.Z_RLE)):
        f = tmp_path / (name + ".fq.gz")
        f.write_bytes(member(data, 6, strategy))
        # fixed-Huffman streams have no dynamic block headers to find: one segment, still on the device
        check(oracle, f, data)
    f = tmp_path / "stored.fq.gz"
    f.write_bytes(member(data, 0))
    check(oracle, f, data)


def test_schedule_shapes(gpu, oracle, tmp_path):
    """the event-driven schedule (r5) at every setting of its rings: 2 .. 6 sets of symbols (compressed-byte buffers: sets + 2, at most 8),
    1 .. 3 decode streams, the first batch split — a file of several members across ~40 batches of 8 segments, so that every ring wraps many
    times, with a truncation that must still behave as on the host path; rows == oracle, on the device, several batches"""
    data = fastq_bytes(9_000_000, seed=57)
    f = tmp_path / "shapes.fq.gz"
    f.write_bytes(member(data[:5_000_000]) + member(data[5_000_000:]) + b"\0 trailing")
    want = oracle.tsv(oracle.count(np.frombuffer(data, dtype=np.uint8))) + "\n"
    # (the last two: the feed through the pinned ring and the copy engine, r2 - r4's and still what SCFQ_GZ_DEVICE_HOST_WRITES=0 selects, instead
    # of the host's threads writing fine-grained device memory)
    for slots, streams, first_div, host_writes in ((2, 1, 1, 1), (3, 2, 1, 1), (4, 3, 1, 1), (5, 1, 4, 1), (6, 3, 1, 1), (3, 1, 4, 1), (3, 1, 1, 0), (5, 2, 4, 0)):
        r = run(f, **dict(BATCH_ENV, SCFQ_GZ_DEVICE_SLOTS=str(slots), SCFQ_GZ_DEVICE_DECODE_STREAMS=str(streams), SCFQ_GZ_DEVICE_FIRST_BATCH_DIV=str(first_div),
                          SCFQ_GZ_DEVICE_HOST_WRITES=str(host_writes)))
        assert r.returncode == 0 and r.stdout == want, (slots, streams, first_div, host_writes, r.stderr[-2000:])
        assert "on the chain" in r.stderr and " 1 batch(es)" not in r.stderr and "the rest on the host" not in r.stderr, (slots, streams, host_writes, r.stderr[-1500:])
        assert ("written to the device" in r.stderr) == bool(host_writes), (host_writes, r.stderr[-1500:])


def test_other_corpora(gpu, oracle, tmp_path):
    rng = np.random.default_rng(5)
    corpora = {
        "crlf": fastq_bytes(4_000_000, seed=17).replace(b"\n", b"\r\n"),
        "long_reads": b"".join(b"@r%d\n" % i + bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 40_000)) + b"\n+\n" +
                               bytes(rng.integers(34, 74, 40_000, dtype=np.uint8)) + b"\n" for i in range(60)),
        "random_bytes": rng.integers(0, 256, 3_000_000, dtype=np.uint8).tobytes(),
        # matches at the far end of the window next to fresh bytes (ratio ~2: inside a segment's output room)
        "far_repeats": b"".join(blk + rng.integers(0, 256, 10_000, dtype=np.uint8).tobytes()
                                for blk in [rng.integers(0, 256, 20_000, dtype=np.uint8).tobytes()] * 120),
    }
    for name, data in corpora.items():
        f = tmp_path / (name + ".fq.gz")
        f.write_bytes(member(data))
        check(oracle, f, data)
    # highly compressible input overflows a segment's output room: the host path takes the file, same row
    for name, data in (("zeros", bytes(8_000_000)), ("one_record", b"@r x\r\nACGTNNGCGC\r\n+\r\nIIII#III@+\r\n" * 150_000)):
        f = tmp_path / (name + ".fq.gz")
        f.write_bytes(member(data))
        check(oracle, f, data, expect_device=False)


def test_members_and_trailing_garbage(gpu, oracle, tmp_path):
    data = fastq_bytes(7_000_000, seed=41)
    cuts = [0, 1_000_003, 1_000_003 + 37, 4_500_000, len(data)]       # members end at arbitrary bytes; one is tiny
    img = b"".join(member(data[a:b], lvl) for (a, b), lvl in zip(zip(cuts[:-1], cuts[1:]), (6, 9, 1, 6)))
    assert gzip.decompress(img) == data
    f = tmp_path / "four_members.fq.gz"
    f.write_bytes(img)
    r = check(oracle, f, data)
    assert "4 member(s)" in r.stderr
    g = tmp_path / "garbage_after.fq.gz"
    g.write_bytes(img + b"\x00" * 100 + b"not a gzip header")
    check(oracle, g, data)
    # an empty member in the middle, and an empty last member
    e = tmp_path / "empty_members.fq.gz"
    e.write_bytes(member(data[:3_000_000]) + member(b"") + member(data[3_000_000:]) + member(b""))
    check(oracle, e, data)


def test_damaged_files_behave_as_on_the_host_path(gpu, oracle, tmp_path):
    data = fastq_bytes(5_000_000, seed=43)
    img = bytearray(member(data))
    rng = np.random.default_rng(9)
    cases = {"cut_middle": bytes(img[: len(img) // 2]), "cut_trailer": bytes(img[:-5]), "cut_last_block": bytes(img[:-40])}
    for t in range(6):
        bad = bytearray(img)
        bad[int(rng.integers(20, len(bad) - 9))] ^= 1 << int(rng.integers(0, 8))
        cases["flip%d" % t] = bytes(bad)
    bad = bytearray(img)
    bad[-6] ^= 0x40                                       # CRC-32 trailer
    cases["crc"] = bytes(bad)
    bad = bytearray(img)
    bad[-2] ^= 0x01                                       # ISIZE trailer
    cases["isize"] = bytes(bad)
    for name, raw in cases.items():
        f = tmp_path / (name + ".fq.gz")
        f.write_bytes(raw)
        host = run(f, SCFQ_GZ_DEVICE="0")
        strip = lambda s: "\n".join(l for l in s.splitlines() if not l.startswith("scfq"))     # noqa: E731
        for env in (DEV_ENV, BATCH_ENV, GROUP_ENV):
            dev = run(f, **env)
            assert (dev.returncode, dev.stdout) == (host.returncode, host.stdout), (name, dev.stderr[-1500:], host.stderr[-500:])
            assert strip(dev.stderr) == strip(host.stderr), name
        try:
            zlib_ok = gzip.decompress(raw) is not None
        except Exception:      # noqa: BLE001
            zlib_ok = False
        assert (dev.returncode == 0) == zlib_ok, name
