"""Device gzip path (csrc/scfq_gzdev.hpp), the memory side of it: a segment's output room comes from an estimate, so a segment that
compresses better is decoded again with more room (never the whole file on the host); a batch that has no room at all hands the
REST of the file to the host's decoder and keeps what the batches before it folded; any number of members; members and files
beyond 4 GiB of inflated bytes (ISIZE is a length modulo 2^32).  Rows must be the oracle's / the generator's tallies.
Reference: src/fq_count.nim:30-34, gzip_stream.nim:16-17 (zlib gzread)."""
import os
import struct
import subprocess
import sys
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from conftest import PKG
from test_ingest_sources import fastq_bytes

pytestmark = pytest.mark.gpu

SC = os.path.join(PKG, "sc")
DEV_ENV = {"SCFQ_GZ_DEVICE_MIN_MB": "0", "SCFQ_GZ_DEVICE_SEGMENT_KB": "32", "SCFQ_VERBOSE": "1"}
BATCH_ENV = dict(DEV_ENV, SCFQ_GZ_DEVICE_BATCH_SEGMENTS="8", SCFQ_GZ_DEVICE_CHAIN_GROUP="3")


def run(path, **env):
    return subprocess.run([SC, "fq-count", str(path)], capture_output=True, text=True, env=dict(os.environ, **env), timeout=900)


def member(data, level=6):
    co = zlib.compressobj(level, zlib.DEFLATED, 31)
    return co.compress(data) + co.flush()


def test_segment_that_needs_more_room_is_decoded_again(gpu, oracle, tmp_path):
    data = fastq_bytes(6_000_000, seed=77)
    f = tmp_path / "tight.fq.gz"
    f.write_bytes(member(data))
    want = oracle.tsv(oracle.count(np.frombuffer(data, dtype=np.uint8))) + "\n"
    # room for 2 symbols per compressed byte (+ 128 Ki) where the data needs 4.4: every segment of 128 KiB overflows once and is decoded
    # again with 4 x the room; with 0.4 per byte twice (4 x, 16 x)
    big = dict(SCFQ_GZ_DEVICE_SEGMENT_KB="128")
    for env in (dict(DEV_ENV, SCFQ_GZ_DEVICE_RATIO="2", **big), dict(BATCH_ENV, SCFQ_GZ_DEVICE_RATIO="2", **big), dict(DEV_ENV, SCFQ_GZ_DEVICE_RATIO="0.4", **big)):
        r = run(f, **env)
        assert r.returncode == 0 and r.stdout == want, r.stderr[-3000:]
        assert "on the chain" in r.stderr and "the rest on the host" not in r.stderr, r.stderr[-3000:]
        given = [l for l in r.stderr.splitlines() if "segments given more room" in l]
        assert given and " 0 segments given more room" not in given[-1], r.stderr[-2000:]
    # the estimate taken from the file itself leaves every segment room enough
    r = run(f, **DEV_ENV)
    assert r.returncode == 0 and r.stdout == want and " 0 segments given more room" in r.stderr, r.stderr[-2000:]


def test_bytes_the_sample_did_not_show(gpu, oracle, tmp_path):
    """the block-start search rules out blocks whose literal code covers bytes >= 128 when the file's first 192 KiB (compressed) hold
    none: a file that turns binary later only loses the search there (those blocks are decoded as gaps, or the file goes to the
    host) — the row is the oracle's either way, with and without the rule"""
    rng = np.random.default_rng(81)
    text = fastq_bytes(5_000_000, seed=80)
    data = text + rng.integers(128, 256, 400_000, dtype=np.uint8).tobytes() + b"\n" + text[:2_000_000]
    f = tmp_path / "turns_binary.fq.gz"
    f.write_bytes(member(data))
    want = oracle.tsv(oracle.count(np.frombuffer(data, dtype=np.uint8))) + "\n"
    for env in (DEV_ENV, BATCH_ENV, dict(DEV_ENV, SCFQ_GZ_DEVICE_LITERAL_MASK="0")):
        r = run(f, **env)
        assert r.returncode == 0 and r.stdout == want, r.stderr[-3000:]
    r = run(f, **DEV_ENV)
    assert "rules out 0xf0" in r.stderr, r.stderr[-2000:]
    r = run(f, **dict(DEV_ENV, SCFQ_GZ_DEVICE_LITERAL_MASK="0"))
    assert "rules out 0x00" in r.stderr, r.stderr[-2000:]


def test_batch_without_room_hands_the_rest_to_the_host(gpu, oracle, tmp_path):
    data = fastq_bytes(7_000_000, seed=78)
    cuts = [0, 2_000_003, 2_000_003 + 41, 5_100_000, len(data)]
    one = tmp_path / "one_member.fq.gz"
    one.write_bytes(member(data))
    four = tmp_path / "four_members.fq.gz"
    four.write_bytes(b"".join(member(data[a:b]) for a, b in zip(cuts[:-1], cuts[1:])) + b"trailing garbage")
    want = oracle.tsv(oracle.count(np.frombuffer(data, dtype=np.uint8))) + "\n"
    for f in (one, four):
        for at in (1, 3, 5):
            r = run(f, **dict(BATCH_ENV, SCFQ_GZ_DEVICE_TEST_REST_AT=str(at)))
            assert r.returncode == 0 and r.stdout == want, (f.name, at, r.stderr[-3000:])
            assert "the rest on the host" in r.stderr, (f.name, at, r.stderr[-3000:])
        # batch 0 without room: nothing is through yet, the whole file is the host's
        r = run(f, **dict(BATCH_ENV, SCFQ_GZ_DEVICE_TEST_REST_AT="0"))
        assert r.returncode == 0 and r.stdout == want and "on the chain" not in r.stderr, r.stderr[-2000:]
    # a damaged tail behind the hand-over is still gzread's error, same text as the host path
    bad = bytearray(one.read_bytes())
    bad[-3] ^= 0x10
    g = tmp_path / "bad_isize.fq.gz"
    g.write_bytes(bytes(bad))
    host = run(g, SCFQ_GZ_DEVICE="0")
    dev = run(g, **dict(BATCH_ENV, SCFQ_GZ_DEVICE_TEST_REST_AT="5"))
    strip = lambda s: "\n".join(l for l in s.splitlines() if not l.startswith("scfq"))     # noqa: E731
    assert host.returncode != 0 and (dev.returncode, dev.stdout, strip(dev.stderr)) == (host.returncode, host.stdout, strip(host.stderr)), dev.stderr[-2000:]


def test_a_stretch_without_block_starts_hands_the_rest_to_the_host(gpu, oracle, tmp_path):
    """a member of nothing but STORED blocks (6 MiB of them: the block-start search looks for dynamic-Huffman headers and finds none)
    between two ordinary members: the walk reaches its first block with the next known start more than 64 segments away — too long
    for one wave — and, r4, hands the REST of the file to the host's decoder from there instead of starting the whole file again:
    the batches in front of it stay folded (`the rest on the host`), the row is the oracle's"""
    a = fastq_bytes(30_000_000, seed=91)
    b = fastq_bytes(6_400_000, seed=92)
    c = fastq_bytes(6_000_000, seed=93)
    f = tmp_path / "stored_middle.fq.gz"
    f.write_bytes(member(a) + member(b, level=0) + member(c))
    want = oracle.tsv(oracle.count(np.frombuffer(a + b + c, dtype=np.uint8))) + "\n"
    r = run(f, **dict(DEV_ENV, SCFQ_GZ_DEVICE_BATCH_SEGMENTS="256"))
    assert r.returncode == 0 and r.stdout == want, r.stderr[-3000:]
    assert "without a block start the search accepts" in r.stderr and "the rest on the host" in r.stderr, r.stderr[-3000:]
    host = run(f, SCFQ_GZ_DEVICE="0")
    assert host.returncode == 0 and host.stdout == want


def test_more_than_1024_members(gpu, scfq, tmp_path):
    plan = scfq.synth_plan(0, 20260105, 170_000_000)
    data, info = scfq.synth_host(0, 20260105, plan.records)
    n = 1300
    cuts = [data.size * k // n for k in range(n + 1)]         # members end at arbitrary bytes
    with ThreadPoolExecutor(8) as ex:
        blobs = list(ex.map(lambda ab: member(data[ab[0]:ab[1]].tobytes(), 1), zip(cuts[:-1], cuts[1:])))
    f = tmp_path / "members.fq.gz"
    f.write_bytes(b"".join(blobs))
    r = run(f, SCFQ_GZ_DEVICE_MIN_MB="0", SCFQ_VERBOSE="1", SCFQ_GZ_DEVICE_SEGMENT_KB="32")
    assert r.returncode == 0 and r.stdout == "%d\t%s\t%d\t%d\t%d\n" % (plan.records, r.stdout.split("\t")[1], info.gc_bases, info.n_bases, info.bases), r.stderr[-3000:]
    assert "%d member(s)" % n in r.stderr and "on the chain" in r.stderr, r.stderr[-3000:]


def _pigz_like(path, raw, level=1, step=64 << 20, threads=16):
    """one gzip member written the way pigz does: raw deflate of pieces joined by sync flushes, CRC-32 and ISIZE (mod 2^32) of the whole"""
    cuts = list(range(0, raw.size, step))

    def piece(i):
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        chunk = raw[cuts[i]:cuts[i] + step].tobytes()
        return co.compress(chunk) + co.flush(zlib.Z_FINISH if i == len(cuts) - 1 else zlib.Z_SYNC_FLUSH), zlib.crc32(chunk), len(chunk)
    with ThreadPoolExecutor(threads) as ex:
        parts = list(ex.map(piece, range(len(cuts))))
    crc = 0
    for _, c, ln in parts:
        crc = _crc32_combine(crc, c, ln)
    with open(path, "wb") as f:
        f.write(b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03")
        for b, _, _ in parts:
            f.write(b)
        f.write(struct.pack("<II", crc & 0xFFFFFFFF, raw.size & 0xFFFFFFFF))


def _crc32_combine(crc1, crc2, len2):
    """zlib's crc32_combine (not exported by the Python module): crc of A || B from crc(A), crc(B), |B|"""
    def times(mat, vec):
        s, i = 0, 0
        while vec:
            if vec & 1:
                s ^= mat[i]
            vec >>= 1
            i += 1
        return s

    def square(mat):
        return [times(mat, mat[n]) for n in range(32)]
    if len2 <= 0:
        return crc1
    odd = [0xEDB88320] + [1 << n for n in range(31)]
    even = square(odd)
    odd = square(even)
    while True:
        even = square(odd)
        if len2 & 1:
            crc1 = times(even, crc1)
        len2 >>= 1
        if not len2:
            break
        odd = square(even)
        if len2 & 1:
            crc1 = times(odd, crc1)
        len2 >>= 1
        if not len2:
            break
    return crc1 ^ crc2


def _synth_on_device(scfq, torch, seed, nbytes):
    """the synthetic stream generated on the DEVICE and copied to the host (the host generator is the same pure function, checked against
    it in test_gpu_parity.py, and takes 9 s per GB on one core): (plan, info, bytes as a numpy array)"""
    plan = scfq.synth_plan(0, seed, int(nbytes))
    buf = torch.empty(plan.bytes + 4096, dtype=torch.uint8, device="cuda:0")
    info = scfq.synth_device(0, seed, plan.records, buf.data_ptr(), plan.bytes)
    assert info.bytes == plan.bytes
    data = buf[:plan.bytes].cpu().numpy()
    del buf
    torch.cuda.empty_cache()
    return plan, info, data


def test_crc32_combine_helper():
    a, b = os.urandom(1000), os.urandom(77777)
    assert _crc32_combine(zlib.crc32(a), zlib.crc32(b), len(b)) == zlib.crc32(a + b)


def test_member_beyond_4_gib(gpu, scfq, tmp_path):
    """4.6 GB in ONE member (ISIZE wraps) through the device gzip path, and the same bytes as BGZF through the device BGZF path:
    counters == the generator's tallies (src/fq_count.nim:38-45 on bytes gzread would yield)"""
    plan, info, data = _synth_on_device(scfq, gpu, 20260104, 4.6e9)
    assert plan.bytes > (1 << 32) + (200 << 20)
    want = "%d\t" % plan.records, "\t%d\t%d\t%d\n" % (info.gc_bases, info.n_bases, info.bases)
    f = tmp_path / "big_member.fq.gz"
    _pigz_like(str(f), data)
    r = subprocess.run([SC, "fq-count", str(f)], capture_output=True, text=True, env=dict(os.environ, SCFQ_VERBOSE="1"), timeout=900)
    assert r.returncode == 0 and r.stdout.startswith(want[0]) and r.stdout.endswith(want[1]), r.stderr[-3000:]
    assert "on the chain" in r.stderr and "the rest on the host" not in r.stderr and "%d bytes inflated" % plan.bytes in r.stderr, r.stderr[-2000:]
    os.remove(f)

    def block(b):
        co = zlib.compressobj(1, zlib.DEFLATED, -15)
        payload = co.compress(b) + co.flush()
        return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 18 + len(payload) + 8 - 1) + payload +
                struct.pack("<II", zlib.crc32(b) & 0xFFFFFFFF, len(b)))

    def span(i):
        a = data[i:i + (32 << 20)]
        return b"".join(block(a[o:o + 65280].tobytes()) for o in range(0, a.size, 65280))
    g = tmp_path / "big_bgzf.fq.gz"
    with ThreadPoolExecutor(16) as ex, open(g, "wb") as out:
        for s in ex.map(span, range(0, data.size, 32 << 20)):
            out.write(s)
        out.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
    r = subprocess.run([SC, "fq-count", "--stats", str(g)], capture_output=True, text=True, env=dict(os.environ, SCFQ_VERBOSE="1"), timeout=900)
    assert r.returncode == 0 and r.stdout.startswith(want[0]) and r.stdout.endswith(want[1]), r.stderr[-3000:]
    assert '"input_bytes": %d' % plan.bytes in r.stderr and int(r.stderr.split('"h2d_bytes": ')[1].split(",")[0].rstrip("}")) < plan.bytes // 2, r.stderr[-2000:]


def test_wide_segments_on_a_level_1_member(gpu, scfq, tmp_path):
    """A member of more than 1 GiB of deflate data counted INSIDE a running process: a context's later sessions cut such a file into 128 KiB
    segments (scfq_gzdev.hpp "wide_segments"; a fresh process keeps 64 KiB) — here 4 GB of the synthetic stream at zlib level 1, whose blocks
    and matches are not level 6's (the configs[3] member below): three calls (the first may still be this context's first session), every
    row == the generator's tallies == the row of a fresh process.  src/fq_count.nim:30-45 over gzread's bytes (gzip_stream.nim:16-17)."""
    plan, info, data = _synth_on_device(scfq, gpu, 20260202, 4e9)
    want = (plan.records, info.gc_bases, info.n_bases, info.bases)
    f = tmp_path / "level1.fq.gz"
    _pigz_like(str(f), data, level=1)
    del data
    assert os.path.getsize(f) > (1 << 30)
    for _ in range(3):
        c = scfq.count_file(str(f))
        assert (c.reads, c.gc_bases, c.n_bases, c.bases) == want
    r = subprocess.run([SC, "fq-count", str(f)], capture_output=True, text=True, env=dict(os.environ, SCFQ_VERBOSE="1"), timeout=600)
    c = r.stdout.strip().split("\t")
    assert r.returncode == 0 and (int(c[0]), int(c[2]), int(c[3]), int(c[4])) == want, r.stderr[-2000:]
    assert "segments of 64 KiB" in r.stderr and "the rest on the host" not in r.stderr, r.stderr[-2000:]
    os.remove(f)


@pytest.fixture(scope="module")
def configs3_member(gpu, scfq, tmp_path_factory):
    """BASELINE configs[3]'s file, written once per module: 10 GB of the seed-20260101 stream as ONE gzip member (zlib level 6, written the
    way pigz does it by 16 threads).  (path, (records, gc, n, bases), inflated bytes)"""
    plan, info, data = _synth_on_device(scfq, gpu, 20260101, 10e9)
    f = tmp_path_factory.mktemp("configs3") / "configs3.fq.gz"
    _pigz_like(str(f), data, level=6)
    del data
    yield f, (plan.records, info.gc_bases, info.n_bases, info.bases), plan.bytes
    if os.path.exists(f):
        os.remove(f)


def test_configs3_ten_gb_member(gpu, scfq, configs3_member):
    """BASELINE configs[3] at its own size: 10 GB of the seed-20260101 stream as ONE gzip member (zlib level 6, written the way pigz
    does it by 16 threads), counted (a) through the device inflate by a fresh `sc fq-count` process and by a second call in this
    process, (b) through the path north_star names — the HOST inflates (SCFQ_GZ_DEVICE=0: the library's parallel reader) into pinned
    buffers while the copy stream moves the chunk before to HBM and the compute stream scans the one before that.  Counters == the
    generator's tallies; the scan kernels stay hidden under the host's fill.  (src/fq_count.nim:30-45, gzip_stream.nim:16-17)"""
    import json
    import time
    f, want, n_inflated = configs3_member

    class _Plan:
        bytes = n_inflated
    plan = _Plan()

    def row(r):
        c = r.stdout.strip().split("\t")
        return int(c[0]), int(c[2]), int(c[3]), int(c[4])
    t = time.time()
    r = subprocess.run([SC, "fq-count", "--stats", str(f)], capture_output=True, text=True, env=dict(os.environ, SCFQ_VERBOSE="1"), timeout=900)
    cold_s = time.time() - t
    assert r.returncode == 0 and row(r) == want, r.stderr[-3000:]
    assert "on the chain" in r.stderr and "the rest on the host" not in r.stderr and "%d bytes inflated" % plan.bytes in r.stderr, r.stderr[-2000:]
    # (the fresh process above decodes in 64 KiB segments; a context's later sessions — these: the module's earlier tests have used it —
    # take 128 KiB ones for a file of this size, the first of them growing the buffers: scfq_gzdev.hpp "wide_segments")
    walls = []
    for _ in range(3):
        t = time.time()
        c = scfq.count_file(str(f))
        walls.append(time.time() - t)
        assert (c.reads, c.gc_bases, c.n_bases, c.bases) == want
    walls = [walls[0], min(walls[1:])]
    r = subprocess.run([SC, "fq-count", "--stats", str(f)], capture_output=True, text=True, env=dict(os.environ, SCFQ_GZ_DEVICE="0"), timeout=900)
    assert r.returncode == 0 and row(r) == want, r.stderr[-3000:]
    st = [json.loads(ln) for ln in r.stderr.splitlines() if ln.startswith("{")][-1]
    assert st["h2d_bytes"] == plan.bytes and st["scan_kernel_ms"] < 0.25 * st["host_fill_ms"], st
    print("configs[3] 10 GB member: device inflate %.3f s as a fresh process, %.3f s second call in a process (%.1f GB/s); host inflate overlapped: "
          "ingest wall %.0f ms, host fill %.0f ms, scan kernels %.1f ms"
          % (cold_s, walls[1], plan.bytes / walls[1] / 1e9, st["ingest_wall_ms"], st["host_fill_ms"], st["scan_kernel_ms"]))
    assert walls[1] < 1.0          # (r3: 0.129 s; a regression to the host path's 1.8 s must not pass for the device path)


def test_configs3_member_across_ranks(gpu, scfq, configs3_member):
    """The 10 GB configs[3] member over 2 and 4 `sc fq-count --shard-rank` processes on this one device (TCP transport): a rank's share of
    the ONE deflate stream is 0.6 - 1.2 GB — several batches of 4096 segments each, the proven symbols of every batch kept across batch
    pools until the other ranks' window maps arrive (SCFQ_SHARD_GZ_KEEP=1, the default) or decoded a second time (=0), and once with the
    device reporting too little free memory for the kept symbols, which must send that rank over its stretch twice of its own accord.
    Row == the generator's tallies, every rank moved its share of the compressed bytes, the member's CRC-32 joined from the stretches
    (a mismatch would send the file to rank 0: `h2d_bytes` of the others would be 0).  src/fq_count.nim:30-34 over gzread's bytes."""
    import json
    import socket
    import time
    f, want, n_inflated = configs3_member
    gz_bytes = os.path.getsize(f)

    def free_port():
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            return s_.getsockname()[1]

    def ranks(world, **env):
        port = free_port()
        t0 = time.time()
        procs = [subprocess.Popen([SC, "fq-count", "--shard-rank=%d" % r, "--shard-world=%d" % world, "--rendezvous=127.0.0.1:%d" % port, "--transport=tcp",
                                   "--devices=0", "--stats", str(f)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                                  env=dict(os.environ, SCFQ_VERBOSE="1", **env)) for r in reversed(range(world))]
        outs = [p.communicate(timeout=600) for p in procs][::-1]          # rank order
        wall = time.time() - t0
        assert [p.returncode for p in procs] == [0] * world, [o[1][-1500:] for o in outs]
        c = outs[0][0].strip().split("\t")
        assert (int(c[0]), int(c[2]), int(c[3]), int(c[4])) == want, (world, env, outs[0][0])
        stats = [[json.loads(ln) for ln in se.splitlines() if ln.startswith("{")][-1] for _, se in outs]
        shares = [st["h2d_bytes"] for st in stats]
        # balanced: every rank moved about a world-th of the compressed bytes (a batch's margin and the cut's block on top)
        assert all(0.7 * gz_bytes / world < s_ < 1.3 * gz_bytes / world + (16 << 20) for s_ in shares), (world, env, shares, gz_bytes)
        for _, se in outs:
            assert "did not join up" not in se and "declined at" not in se, se[-2500:]
            nb = [int(ln.split(" batch(es)")[0].split()[-1]) for ln in se.splitlines() if " batch(es), " in ln and "segments planned" in ln]
            assert nb and min(nb) >= 2, (world, env, nb)                 # several batches per stretch (and per pass)
        return wall, stats, outs

    rows = []
    for world, keep in ((2, "1"), (4, "1"), (4, "0")):
        wall, stats, outs = ranks(world, SCFQ_SHARD_GZ_KEEP=keep)
        passes = [sum(1 for ln in se.splitlines() if " batch(es), " in ln and "segments planned" in ln) for _, se in outs]
        assert passes == [1 if keep == "1" else 2] * world, (world, keep, passes)
        rows.append((world, keep, wall, [round(st["ingest_wall_ms"]) for st in stats]))
    # a device that reports 20 GB free (SCFQ_TEST_DEVICE_FREE_GB): the kept symbols of a 0.6 GB stretch + the pipeline's 16 GB do not fit,
    # so every rank goes over its stretch twice although keeping was asked for
    wall, stats, outs = ranks(4, SCFQ_SHARD_GZ_KEEP="1", SCFQ_TEST_DEVICE_FREE_GB="20")
    passes = [sum(1 for ln in se.splitlines() if " batch(es), " in ln and "segments planned" in ln) for _, se in outs]
    assert passes == [2] * 4, passes
    rows.append((4, "1, short of memory", wall, [round(st["ingest_wall_ms"]) for st in stats]))
    for world, keep, wall, ing in rows:
        print("configs[3] member across %d ranks on one device (keep=%s): %.2f s for the group of processes, ingest wall per rank %s ms" % (world, keep, wall, ing))
