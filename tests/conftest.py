import ctypes
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "seq-collection_amd")
sys.path.insert(0, os.path.join(PKG, "pyhost"))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _ensure_built():
    """hipcc cross-compiles gfx950 without a GPU: (re)build the product library + CLI when missing or stale."""
    lib = os.path.join(PKG, "libsc_fqcount_hip.so")
    srcs = [os.path.join(PKG, "csrc", f) for f in os.listdir(os.path.join(PKG, "csrc"))] + \
           [os.path.join(PKG, "cli", "sc_main.cpp"), os.path.join(PKG, "Makefile"), os.path.join(ROOT, "include", "sc_fqcount.h"),
            os.path.join(ROOT, "include", "sc_fqcount_debug.h")]
    newest = max(os.path.getmtime(s) for s in srcs)
    for target in (lib, os.path.join(PKG, "sc")):
        if not os.path.exists(target) or os.path.getmtime(target) < newest:
            subprocess.check_call(["make", "-C", PKG], stdout=subprocess.DEVNULL)
            break


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    _ensure_built()


class OracleCounts(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint64) for n in
                "reads gc_bases n_bases bases lines newlines input_bytes bad_at bad_plus".split()] + \
               [("qual_hist", ctypes.c_uint64 * 256)]


class OracleDedupStats(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint64) for n in "total_reads duplicates false_positive records_out bytes_out".split()]


def _build_oracle():
    so = os.path.join(ROOT, "oracle", "libfqcount_oracle.so")
    srcs = [os.path.join(ROOT, "oracle", f) for f in ("fqcount_oracle.c", "fqdedup_oracle.c", "fqcount_oracle.h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    return so


def make_oracle():
    """The CPU restatement (oracle/): test infrastructure, the checker the HIP path is compared with."""
    L = ctypes.CDLL(_build_oracle())
    L.oracle_count_lines.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(OracleCounts)]
    L.oracle_count_bytes.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(OracleCounts)]
    L.oracle_count_file.argtypes = [ctypes.c_char_p, ctypes.POINTER(OracleCounts)]
    L.oracle_partial.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64),
                                 ctypes.c_void_p]
    L.oracle_format_tsv.argtypes = [ctypes.POINTER(OracleCounts), ctypes.c_char_p, ctypes.c_size_t]
    L.oracle_dedup.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(OracleDedupStats)]
    L.oracle_dedup.restype = ctypes.c_int64

    class O:
        lib = L

        @staticmethod
        def _buf(data):
            import numpy as np
            a = np.frombuffer(bytes(data), dtype=np.uint8) if not hasattr(data, "ctypes") else data
            return a

        @classmethod
        def count(cls, data, which="bytes"):
            a = cls._buf(data)
            c = OracleCounts()
            fn = L.oracle_count_bytes if which == "bytes" else L.oracle_count_lines
            fn(a.ctypes.data if a.size else None, a.size, ctypes.byref(c))
            return c

        @classmethod
        def count_file(cls, path):
            c = OracleCounts()
            rc = L.oracle_count_file(os.fsencode(path), ctypes.byref(c))
            return rc, c

        @classmethod
        def partial(cls, data, prev_byte=-1, want_hist=False):
            a = cls._buf(data)
            w = (ctypes.c_uint64 * 27)()
            h = (ctypes.c_uint64 * 1024)() if want_hist else None
            L.oracle_partial(a.ctypes.data if a.size else None, a.size, prev_byte, w,
                             ctypes.byref(h) if want_hist else None)
            return (list(w), list(h)) if want_hist else list(w)

        @classmethod
        def dedup(cls, data):
            """(output bytes, stats) of the fq-dedup restatement"""
            a = cls._buf(data)
            out = (ctypes.c_uint8 * (2 * a.size + 16))()
            st = OracleDedupStats()
            n = L.oracle_dedup(a.ctypes.data if a.size else None, a.size, out, 2 * a.size + 16, ctypes.byref(st))
            assert n >= 0, n
            return bytes(out[:n]), st

        @staticmethod
        def tsv(c):
            b = ctypes.create_string_buffer(256)
            L.oracle_format_tsv(ctypes.byref(c), b, 256)
            return b.value.decode()

    return O


@pytest.fixture(scope="session")
def oracle():
    return make_oracle()


_oracle_for_subprocess = make_oracle      # (tests that run their cases in a child process with another environment)


def golden_rows():
    rows = []
    with open(os.path.join(GOLDEN, "golden.tsv")) as f:
        next(f)
        for line in f:
            name, sha, source, reads, gcc, gc, n, bases = line.rstrip("\n").split("\t")
            rows.append(dict(name=name, sha256=sha, source=source, reads=int(reads), gc_content=gcc,
                             gc_bases=int(gc), n_bases=int(n), bases=int(bases)))
    return rows


@pytest.fixture(scope="session")
def scfq():
    import scfq as m
    return m


@pytest.fixture(scope="session")
def gpu(scfq):
    """Fails (does not skip) when the HIP library or the GPU is missing: -m gpu tests must run native code."""
    import torch
    assert torch.cuda.is_available(), "no GPU visible to torch"
    scfq.lib()
    assert scfq.lib().scfq_device_count() >= 1
    # the tests fill device buffers with torch kernels and hand them straight to the library: order every device-pointer
    # call of this thread after torch's stream (include/sc_fqcount.h: scfq_set_wait_stream) instead of synchronising by hand
    scfq.set_wait_stream(torch.cuda.current_stream().cuda_stream)
    return torch
