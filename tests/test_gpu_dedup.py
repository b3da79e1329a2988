"""fq-dedup on the GPU (scfq_dedup_*: line index, header hashes, radix sort, exact compare, scan, gather) against the CPU
restatement of src/fq_dedup.nim: byte-identical output, identical statistics."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN
from test_gpu_hist_spec import make_fastq
from test_gpu_parity import random_fastq_like, to_dev

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def check(scfq, oracle, data, ctx):
    a = np.frombuffer(bytes(data), dtype=np.uint8) if not hasattr(data, "ctypes") else data
    want, ost = oracle.dedup(a)
    got, st = scfq.dedup_host(a)
    assert (st.total_reads, st.duplicates, st.records_out, st.bytes_out) == (ost.total_reads, ost.duplicates, ost.records_out, ost.bytes_out), ctx
    assert got == want, ctx
    return st


def with_duplicates(rng, a, frac, shuffle_payload=True):
    """re-insert copies of random records (same header, possibly different sequence/quality) at random record boundaries"""
    nl = np.flatnonzero(a == 10)
    starts = np.concatenate([[0], nl[3::4] + 1])[:-1] if nl.size % 4 == 0 else np.concatenate([[0], nl[3::4] + 1])
    ends = np.concatenate([starts[1:], [a.size]])
    recs = [a[s:e].tobytes() for s, e in zip(starts, ends)]
    n_dup = max(1, int(len(recs) * frac))
    out = list(recs)
    for _ in range(n_dup):
        r = recs[int(rng.integers(0, len(recs)))]
        lines = r.split(b"\n")
        if shuffle_payload and len(lines) >= 4 and rng.integers(0, 2):
            lines[1] = lines[1][::-1]          # same ID, different read: still a duplicate by ID
        out.insert(int(rng.integers(0, len(out) + 1)), b"\n".join(lines))
    return np.frombuffer(b"".join(out), dtype=np.uint8)


def test_reference_fixture(gpu, scfq, oracle, tmp_path):
    raw = open(os.path.join(GOLDEN, "dup.fq"), "rb").read()
    st = check(scfq, oracle, raw, "dup.fq")
    assert (st.total_reads, st.duplicates) == (8, 4)
    for name in ("dup.fq", "dup.fq.gz", "nodup.fq"):       # scripts/functional-tests.sh:86-92 through the file entry + CLI
        out = tmp_path / "o.fq"
        with open(out, "wb") as f:
            st = scfq.dedup_file(os.path.join(GOLDEN, name), f.fileno())
        got = out.read_bytes()
        want, ost = oracle.dedup(open(os.path.join(GOLDEN, "dup.fq" if name.startswith("dup") else name), "rb").read())
        assert got == want and st.duplicates == ost.duplicates, name
        if name.startswith("dup"):
            assert sum(1 for line in got.split(b"\n") if b"@" in line) == 4
    sc = os.path.join(os.path.dirname(HERE), "seq-collection_amd", "sc")
    r = subprocess.run([sc, "fq-dedup", os.path.join(GOLDEN, "dup.fq.gz")], capture_output=True)
    assert r.returncode == 0 and r.stdout == oracle.dedup(raw)[0]
    assert r.stderr.decode().splitlines() == ["total_reads: 8", "duplicates 4", "false-positive: 0", "false-positive-rate: 0.0"]
    r = subprocess.run([sc, "fq-dedup", os.path.join(GOLDEN, "nodup.fq")], capture_output=True)
    assert r.stderr.decode().splitlines() == ["No Duplicates Found", "Copying fq to stdout", "total_reads: 4", "duplicates 0",
                                              "false-positive: 0", "false-positive-rate: nan"]
    r = subprocess.run([sc, "fq-dedup", "/nonexistent.fq"], capture_output=True)
    assert r.returncode == 1 and b"does not exist or is not readable" in r.stderr


@pytest.mark.parametrize("crlf", [False, True])
def test_wellformed_with_duplicates(gpu, scfq, oracle, crlf):
    rng = np.random.default_rng(3 + crlf)
    for n_rec, frac in ((1, 1.0), (5, 0.5), (200, 0.3), (20000, 0.05), (60000, 0.6)):
        a = with_duplicates(rng, make_fastq(rng, n_rec, crlf=crlf), frac)
        st = check(scfq, oracle, a, ("dups", crlf, n_rec, frac))
        assert st.duplicates >= 1
        check(scfq, oracle, a[:-1], ("no final newline", crlf, n_rec))
        check(scfq, oracle, a[: a.size * 2 // 3], ("truncated", crlf, n_rec))


def test_degenerate_inputs(gpu, scfq, oracle):
    rng = np.random.default_rng(17)
    for data in (b"", b"\n", b"x", b"@a", b"@a\n", b"\n\n\n\n\n\n\n\n\n", b"@a\nA\n+\nI\n" * 3000, b"@a\r", b"\r\n" * 1000):
        check(scfq, oracle, data, data[:20])
    for kind in ("uniform", "ascii", "dense_nl", "sparse_nl", "crlf"):
        for n in (1, 100, 5000, 1_000_000):
            check(scfq, oracle, random_fastq_like(rng, n, kind), (kind, n))
    # one ID a million times; every ID twice, far apart
    check(scfq, oracle, b"@same id\nACGT\n+\nIIII\n" * 300000, "one id")
    a = make_fastq(rng, 50000)
    check(scfq, oracle, np.concatenate([a, a]), "doubled")


def test_device_resident_and_generator(gpu, scfq, oracle):
    torch = gpu
    plan = scfq.synth_plan(0, 20260101, 64 << 20)
    t = torch.empty(2 * plan.bytes + 4096, dtype=torch.uint8, device="cuda")
    scfq.synth_device(0, 20260101, plan.records, t.data_ptr(), plan.bytes)
    t[plan.bytes:2 * plan.bytes] = t[:plan.bytes]            # the whole file twice: the second half is all duplicates
    n = 2 * plan.bytes
    nb, st = scfq.dedup_device(t.data_ptr(), n)
    # generator IDs are unique per record (lane:tile:x:y drawn per record may repeat: compare with the oracle, not a formula)
    a = t[:n].cpu().numpy()
    want, ost = oracle.dedup(a)
    assert (nb, st.duplicates, st.total_reads) == (len(want), ost.duplicates, 2 * plan.records)
    out = torch.empty(nb + 64, dtype=torch.uint8, device="cuda")
    nb2, st2 = scfq.dedup_device(t.data_ptr(), n, out.data_ptr(), nb)
    assert nb2 == nb and out[:nb].cpu().numpy().tobytes() == want


def test_hash_collisions_are_resolved_exactly(gpu, scfq, oracle):
    """SCFQ_DEDUP_HASH_BITS truncates the header hash (test hook): every equal-hash run then mixes many different IDs and
    only the exact compare separates them.  4 bits and 32 (the default) take the 32-bit key path, 40 and 64 the 64-bit one: same
    bytes, same counts."""
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import scfq, ctypes, conftest\n"
        "from test_gpu_hist_spec import make_fastq\n"
        "from test_gpu_dedup import with_duplicates\n"
        "rng = np.random.default_rng(5)\n"
        "a = with_duplicates(rng, make_fastq(rng, 3000), 0.4)\n"
        "got, st = scfq.dedup_host(a)\n"
        "L = ctypes.CDLL(conftest._build_oracle())\n"
        "out = (ctypes.c_uint8 * (2 * a.size))(); ost = conftest.OracleDedupStats()\n"
        "L.oracle_dedup.restype = ctypes.c_int64\n"
        "L.oracle_dedup.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]\n"
        "n = L.oracle_dedup(a.ctypes.data, a.size, out, 2 * a.size, ctypes.byref(ost))\n"
        "assert got == bytes(out[:n]) and st.duplicates == ost.duplicates, (st.duplicates, ost.duplicates)\n"
        "import os\n"
        "assert (st.hash_collisions > 1000) == (os.environ['SCFQ_DEDUP_HASH_BITS'] == '4'), st.hash_collisions\n"
        "print('collisions ok', st.hash_collisions)\n"
    ) % (os.path.join(os.path.dirname(HERE), "seq-collection_amd", "pyhost"), HERE)
    for bits in ("4", "32", "40", "64"):
        env = dict(os.environ, SCFQ_DEDUP_HASH_BITS=bits)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        assert r.returncode == 0 and "collisions ok" in r.stdout, (bits, r.stdout + r.stderr)


def test_large_input_properties(gpu, scfq):
    """size-independent properties at a few GB (no oracle needed): doubling the input changes nothing but the statistics,
    de-duplication is idempotent, and every record is either echoed or counted as a duplicate"""
    torch = gpu
    plan = scfq.synth_plan(0, 20260101, 1 << 30)
    n1 = plan.bytes
    buf = torch.empty(3 * n1 + 4096, dtype=torch.uint8, device="cuda")
    scfq.synth_device(0, 20260101, plan.records, buf.data_ptr(), n1)
    buf[n1:2 * n1] = buf[:n1]
    buf[2 * n1:3 * n1] = buf[:n1]
    out1 = torch.empty(n1 + 4096, dtype=torch.uint8, device="cuda")
    out3 = torch.empty(n1 + 4096, dtype=torch.uint8, device="cuda")
    nb1, st1 = scfq.dedup_device(buf.data_ptr(), n1, out1.data_ptr(), n1)
    nb3, st3 = scfq.dedup_device(buf.data_ptr(), 3 * n1, out3.data_ptr(), n1)
    assert st1.total_reads == plan.records and st3.total_reads == 3 * plan.records
    assert st1.records_out + st1.duplicates == plan.records and st3.records_out + st3.duplicates == 3 * plan.records
    assert nb3 == nb1 and st3.records_out == st1.records_out and st3.duplicates == st1.duplicates + 2 * plan.records
    assert torch.equal(out1[:nb1], out3[:nb3])                       # the copies add nothing
    again = torch.empty(nb1 + 4096, dtype=torch.uint8, device="cuda")
    nb2, st2 = scfq.dedup_device(out1.data_ptr(), nb1, again.data_ptr(), nb1)
    assert nb2 == nb1 and st2.duplicates == 0 and torch.equal(again[:nb2], out1[:nb1])     # idempotent
    # the generator's IDs are nearly unique: what is dropped from a single copy is a handful of chance repeats
    assert st1.duplicates < plan.records // 1000


def _records(rng, n, hdr_len, read_len, at_quality=False, crlf=False):
    """n records with headers of hdr_len bytes (unique: a counter inside), reads of read_len, optionally quality lines that start with '@'"""
    eol = b"\r\n" if crlf else b"\n"
    out = []
    for i in range(n):
        tag = b"@r%d/" % i
        hdr = tag + bytes(rng.integers(97, 123, max(0, hdr_len - len(tag)), dtype=np.uint8))      # lowercase filler
        seq = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), read_len))
        q = bytes(rng.integers(35, 75, read_len, dtype=np.uint8))
        if at_quality and read_len:
            q = b"@" + q[1:]
        out.append(hdr + eol + seq + eol + b"+" + eol + q + eol)
    return out


@pytest.mark.parametrize("fused", ["1", "0"])
def test_header_hashes_from_the_index_pass(gpu, fused):
    """The line index hashes the headers it has in LDS (scfq_hdrhash.hpp); what it cannot do goes to the hash kernel: headers that cross
    a 4 KiB tile, headers longer than 255 bytes, files of short lines (64+ newlines in a tile: its list of left-over records is not
    complete then), the input's first line, the "line" behind the last newline.  Every shape against the oracle, with the fused path
    (default) and without it (SCFQ_DEDUP_FUSED_HASH=0): same bytes, same statistics."""
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import scfq, conftest\n"
        "from test_gpu_dedup import _records, check\n"
        "oracle = conftest._oracle_for_subprocess()\n"
        "rng = np.random.default_rng(11)\n"
        "shapes = [dict(hdr_len=57, read_len=150), dict(hdr_len=57, read_len=150, crlf=True), dict(hdr_len=300, read_len=100),\n"
        "          dict(hdr_len=255, read_len=40), dict(hdr_len=256, read_len=40), dict(hdr_len=20, read_len=25), dict(hdr_len=12, read_len=8),\n"
        "          dict(hdr_len=57, read_len=150, at_quality=True), dict(hdr_len=64, read_len=64), dict(hdr_len=65, read_len=1)]\n"
        "for sh in shapes:\n"
        "    recs = _records(rng, 6000, **sh)\n"
        "    dup = [recs[int(k)] for k in rng.integers(0, len(recs), 1500)]\n"
        "    allr = recs + dup\n"
        "    order = rng.permutation(len(allr))\n"
        "    data = b''.join(allr[int(k)] for k in order)\n"
        "    st = check(scfq, oracle, data, sh)\n"
        "    assert st.duplicates == 1500, (sh, st.duplicates)\n"
        "    for cut in (1, 2, 7):\n"
        "        check(scfq, oracle, data[:-cut], (sh, 'cut', cut))\n"
        "    check(scfq, oracle, b'x' * 4090 + data, (sh, 'shifted'))\n"
        "print('fused cases ok')\n"
    ) % (os.path.join(os.path.dirname(HERE), "seq-collection_amd", "pyhost"), HERE)
    env = dict(os.environ, SCFQ_DEDUP_FUSED_HASH=fused)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    assert r.returncode == 0 and "fused cases ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
