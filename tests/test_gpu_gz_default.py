"""BASELINE configs[3] on the GPU: gzip-compressed FASTQ through scfq_count_file's DEFAULT path — device-side inflate of the
compressed bytes (csrc/gz_inflate_kernels.hpp) — and through the host path behind it (SCFQ_GZ_DEVICE=0: the library's own
DEFLATE decoder, one member on many threads, pinned ring, copy stream, scans overlapped with the inflate), both against the
oracle on the inflated bytes.  Reference: src/fq_count.nim:30-34 (".gz" -> newGZFileStream), gzip_stream.nim:16-17 (gzread)."""
import json
import ctypes
import gzip
import os
import subprocess
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

REF_FIELDS = ("reads", "gc_bases", "n_bases", "bases", "lines")


@pytest.fixture(scope="module")
def illumina(scfq):
    """~270 MB of the synthetic Illumina stream of SURVEY.md §8d (host generator: the same bytes as the device generator)"""
    plan = scfq.synth_plan(0, 20260101, 270_000_000)
    data, info = scfq.synth_host(0, 20260101, plan.records)
    return data, info


def _member(piece, level=6):
    co = zlib.compressobj(level, zlib.DEFLATED, 31)
    return co.compress(piece) + co.flush()


def _check(scfq, oracle, path, data, info, flags=0):
    c = scfq.count_file(str(path), flags=flags | scfq.SCFQ_TIMING)
    t = scfq.last_timing()
    oc = oracle.count(data, "lines")          # the reference-shaped line loop (src/fq_count.nim:38-45)
    if flags & (scfq.SCFQ_QUAL_HIST | scfq.SCFQ_STRUCT_CHECK):
        oc = oracle.count(data, "bytes")      # the byte-serial restatement also carries the K3 / K4 additions
    for f in REF_FIELDS:
        assert getattr(c, f) == getattr(oc, f), (f, getattr(c, f), getattr(oc, f))
    assert (c.reads, c.gc_bases, c.n_bases, c.bases) == (info.records, info.gc_bases, info.n_bases, info.bases)
    assert c.input_bytes == data.size
    assert scfq.format_tsv(c) == oracle.tsv(oc)
    if flags & scfq.SCFQ_QUAL_HIST:
        assert list(c.qual_hist) == list(oc.qual_hist) and sum(c.qual_hist) == c.bases
    if flags & scfq.SCFQ_STRUCT_CHECK:
        assert (c.bad_at, c.bad_plus) == (oc.bad_at, oc.bad_plus) == (0, 0)
    return c, t


def test_single_member_gzip6_default_path(gpu, scfq, oracle, illumina, tmp_path):
    data, info = illumina
    f = tmp_path / "one_member.fq.gz"
    f.write_bytes(_member(data.tobytes()))
    assert os.path.getsize(f) >= 60 << 20          # a >= 64 MiB-class member: the parallel single-member reader's case
    c, t = _check(scfq, oracle, f, data, info)
    # default: the COMPRESSED bytes cross PCIe and the device inflates them; one scan over the proven stream
    assert t.h2d_bytes == os.path.getsize(f) and t.scan_bytes == data.size and t.scan_launches == 1
    # the host path behind it (parallel single-member reader): the inflate is the critical path, copy and scan hide under it
    r = subprocess.run([os.path.join(ROOT, "seq-collection_amd", "sc"), "fq-count", "--stats", str(f)], capture_output=True, text=True,
                       env=dict(os.environ, SCFQ_GZ_DEVICE="0"), timeout=600)
    assert r.returncode == 0 and r.stdout == scfq.format_tsv(c) + "\n", r.stderr
    st = json.loads([ln for ln in r.stderr.splitlines() if ln.startswith("{")][-1])
    assert st["h2d_bytes"] == data.size and st["scan_launches"] >= 4
    assert st["scan_kernel_ms"] < 0.5 * st["host_fill_ms"], st
    assert st["ingest_wall_ms"] < 1.25 * st["host_fill_ms"] + 25.0, st      # (wall includes thread start-up: loose on a shared box)
    # the same file through the serial own decoder and through zlib give the same row
    for env in ({"SCFQ_PGZ": "0"}, {"SCFQ_INFLATE": "zlib"}):
        r = subprocess.run([os.path.join(ROOT, "seq-collection_amd", "sc"), "fq-count", str(f)], capture_output=True, text=True,
                           env=dict(os.environ, SCFQ_GZ_DEVICE="0", **env), timeout=600)
        assert r.returncode == 0 and r.stdout == scfq.format_tsv(c) + "\n", (env, r.stderr)
    _check(scfq, oracle, f, data, info, flags=scfq.SCFQ_QUAL_HIST | scfq.SCFQ_STRUCT_CHECK)


def test_thirty_member_concatenation_default_path(gpu, scfq, oracle, illumina, tmp_path):
    data, info = illumina
    raw = data.tobytes()
    cuts = [len(raw) * k // 30 for k in range(31)]          # members end at arbitrary bytes, not at record boundaries
    with ThreadPoolExecutor(8) as ex:
        members = list(ex.map(lambda k: _member(raw[cuts[k]:cuts[k + 1]]), range(30)))
    f = tmp_path / "thirty_members.fq.gz"
    f.write_bytes(b"".join(members))
    c, t = _check(scfq, oracle, f, data, info)
    r = subprocess.run([os.path.join(ROOT, "seq-collection_amd", "sc"), "fq-count", "--stats", str(f)], capture_output=True, text=True,
                       env=dict(os.environ, SCFQ_GZ_DEVICE="0"), timeout=600)
    assert r.returncode == 0 and r.stdout == scfq.format_tsv(c) + "\n", r.stderr
    st = json.loads([ln for ln in r.stderr.splitlines() if ln.startswith("{")][-1])
    assert st["scan_kernel_ms"] < 0.5 * st["host_fill_ms"], st
    # trailing garbage after the last member is ignored, as gzread does
    g = tmp_path / "thirty_members_garbage.fq.gz"
    g.write_bytes(b"".join(members) + b"\x00" * 100)
    _check(scfq, oracle, g, data, info)


def test_corrupt_and_truncated_big_members_are_errors(gpu, scfq, illumina, tmp_path):
    data, _ = illumina
    img = bytearray(_member(data.tobytes()[:40_000_000]))
    bad = bytearray(img)
    bad[len(bad) // 2] ^= 0x10
    f = tmp_path / "flipped.fq.gz"
    f.write_bytes(bad)
    with pytest.raises(scfq.ScfqError) as e:
        scfq.count_file(str(f))
    assert e.value.rc == scfq.SCFQ_EGZ
    try:
        gzip.decompress(bytes(bad))
        raise AssertionError("zlib accepted the flipped stream")
    except (OSError, EOFError, zlib.error):
        pass
    t = tmp_path / "cut.fq.gz"
    t.write_bytes(img[: len(img) * 2 // 3])
    with pytest.raises(scfq.ScfqError) as e:
        scfq.count_file(str(t))
    assert e.value.rc == scfq.SCFQ_EGZ
