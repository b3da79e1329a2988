"""ctypes host binding of libsc_fqcount_hip.so and a Python mirror of the reference's fq-count operator.

Mirrors (reference tree):
  src/fq_count.nim:7-11    fq_count_header
  src/fq_count.nim:14      proc fq_count*(fastq: string, basename: bool, absolute: bool)
  src/utils/helpers.nim:200-224   output_header / get_absolute / output_w_fnames
  src/utils/helpers.nim:29-34     error_msg / quit_error

The counting itself always goes through the C ABI (include/sc_fqcount.h) into the HIP kernels; there is
no Python or CPU fallback here: if the shared library is missing or no GPU is visible the calls raise.

A process that also uses PyTorch: `import torch` BEFORE the first call of this module.  torch ships a libamdhip64 of its own; whichever
copy is loaded first serves both, and with two copies in one process a device pointer of one runtime means nothing to the other
(`scfq_synth_device` / `count_device` on a torch tensor then fail with SCFQ_EHIP).
"""
import ctypes
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SCFQ_LIB_OVERRIDE") or os.path.join(os.path.dirname(_HERE), "libsc_fqcount_hip.so")

SCFQ_QUAL_HIST = 0x1
SCFQ_STRUCT_CHECK = 0x2
SCFQ_TIMING = 0x4
SCFQ_PREV_IN_MEMORY = 0x8
SCFQ_HIST_EXACT = 0x10
SCFQ_WAIT_STREAM = 0x20
SCFQ_EOPEN, SCFQ_EGZ, SCFQ_EHIP, SCFQ_ERCCL, SCFQ_EARG, SCFQ_EIO, SCFQ_ENOMEM, SCFQ_ESPEC, SCFQ_EPIPE = -1, -2, -3, -4, -5, -6, -7, -8, -9
SCFQ_COMM_RCCL, SCFQ_COMM_TCP = 0, 1
COMM_ID_BYTES = 128
PARTIAL_WORDS = 32
HIST_WORDS = 4 * 256

EXPORTS = [
    "scfq_count_file", "scfq_count_buffer", "scfq_partial_buffer", "scfq_partial_identity",
    "scfq_partial_combine", "scfq_partial_finalize", "scfq_format_tsv", "scfq_strerror",
    "scfq_last_error_detail", "scfq_last_timing", "scfq_device_bytes_now", "scfq_device_bytes_high_water", "scfq_device_count", "scfq_shutdown",
    "scfq_debug_partial_simple", "scfq_synth_plan", "scfq_synth_host", "scfq_synth_device", "scfq_synth_locate",
    "scfq_debug_read_file", "scfq_debug_gz_resume", "scfq_debug_stream_ms", "scfq_debug_hist_stats",
    "scfq_index_lines", "scfq_dedup_buffer", "scfq_dedup_file", "scfq_dedup_error_detail", "scfq_stage_file",
    "scfq_device_free", "scfq_meta_header", "scfq_meta_file_tsv", "scfq_debug_bgzf_inflate",
    "scfq_set_wait_stream", "scfq_get_wait_stream", "scfq_count_file_sharded",
    "scfq_comm_unique_id", "scfq_comm_init_rank", "scfq_comm_init_rendezvous", "scfq_comm_init_all", "scfq_comm_world",
    "scfq_comm_rank", "scfq_comm_is_broken", "scfq_prepare", "scfq_comm_transport", "scfq_comm_exchange", "scfq_comm_exchange_start", "scfq_comm_exchange_finish",
    "scfq_comm_allgather_u64", "scfq_comm_destroy", "scfq_comm_error_detail", "scfq_debug_stages", "scfq_debug_stage_mark",
    "scfq_debug_gz_member_boundary", "scfq_debug_gz_shard_fix", "scfq_debug_last_scan_kernel",
]


class Counts(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint64) for n in (
        "struct_size", "abi_version", "reads", "gc_bases", "n_bases", "bases", "lines", "newlines",
        "input_bytes", "bad_at", "bad_plus")] + [("qual_hist", ctypes.c_uint64 * 256)]


class Opts(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_uint64), ("n_devices", ctypes.c_int32),
                ("device_ids", ctypes.POINTER(ctypes.c_int32)), ("flags", ctypes.c_uint32),
                ("reserved", ctypes.c_uint32), ("chunk_bytes", ctypes.c_uint64), ("wait_stream", ctypes.c_void_p)]


class Partial(ctypes.Structure):
    _fields_ = [("nl", ctypes.c_uint64), ("gc", ctypes.c_uint64 * 4), ("n", ctypes.c_uint64 * 4),
                ("len", ctypes.c_uint64 * 4), ("starts", ctypes.c_uint64 * 4),
                ("first_at", ctypes.c_uint64 * 4), ("first_plus", ctypes.c_uint64 * 4),
                ("bytes", ctypes.c_uint64), ("last_byte", ctypes.c_uint64), ("hist_class", ctypes.c_uint64),
                ("reserved", ctypes.c_uint64 * 4)]

    def words(self):
        return list((ctypes.c_uint64 * PARTIAL_WORDS).from_buffer_copy(self))

    @classmethod
    def from_words(cls, w):
        arr = (ctypes.c_uint64 * PARTIAL_WORDS)(*[int(x) & 0xFFFFFFFFFFFFFFFF for x in w])
        return cls.from_buffer_copy(arr)


class Timing(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_uint64), ("scan_kernel_ms", ctypes.c_double),
                ("fold_kernel_ms", ctypes.c_double), ("scan_bytes", ctypes.c_uint64),
                ("scan_launches", ctypes.c_uint64), ("host_fill_ms", ctypes.c_double),
                ("ingest_wall_ms", ctypes.c_double), ("h2d_bytes", ctypes.c_uint64), ("h2d_ms", ctypes.c_double)]


class DedupStats(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint64) for n in ("struct_size", "total_reads", "duplicates", "false_positive", "records_out",
                                               "bytes_out", "hash_collisions")]


class SynthInfo(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint64) for n in ("struct_size", "records", "bytes", "gc_bases", "n_bases", "bases")]


class ScfqError(RuntimeError):
    def __init__(self, rc, what, detail=""):
        self.rc = rc
        super().__init__("%s failed: rc=%d (%s)%s" % (what, rc, strerror(rc), (" — " + detail) if detail else ""))


_lib = None


def lib():
    """Loads the HIP library; fails loudly when it has not been built (no fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libsc_fqcount_hip.so not built: run `make -C %s` (hipcc --offload-arch=gfx950)"
                              % os.path.dirname(LIB_PATH))
        L = ctypes.CDLL(LIB_PATH)
        L.scfq_count_file.argtypes = [ctypes.c_char_p, ctypes.POINTER(Opts), ctypes.POINTER(Counts)]
        L.scfq_count_buffer.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.POINTER(Opts),
                                        ctypes.POINTER(Counts)]
        L.scfq_partial_buffer.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_int,
                                          ctypes.POINTER(Opts), ctypes.POINTER(Partial), ctypes.c_void_p]
        L.scfq_partial_identity.argtypes = [ctypes.POINTER(Partial), ctypes.c_void_p]
        L.scfq_partial_identity.restype = None
        L.scfq_partial_combine.argtypes = [ctypes.POINTER(Partial), ctypes.POINTER(Partial), ctypes.c_void_p,
                                           ctypes.c_void_p]
        L.scfq_partial_finalize.argtypes = [ctypes.POINTER(Partial), ctypes.c_void_p, ctypes.POINTER(Counts)]
        L.scfq_format_tsv.argtypes = [ctypes.POINTER(Counts), ctypes.c_char_p, ctypes.c_uint64]
        L.scfq_strerror.argtypes = [ctypes.c_int]
        L.scfq_strerror.restype = ctypes.c_char_p
        L.scfq_last_error_detail.restype = ctypes.c_char_p
        L.scfq_last_timing.argtypes = [ctypes.POINTER(Timing)]
        L.scfq_debug_partial_simple.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int,
                                                ctypes.POINTER(Partial)]
        for name in ("scfq_synth_plan",):
            getattr(L, name).argtypes = [ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64,
                                         ctypes.POINTER(SynthInfo)]
        for name in ("scfq_synth_host", "scfq_synth_device"):
            getattr(L, name).argtypes = [ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64,
                                         ctypes.c_void_p, ctypes.c_uint64, ctypes.POINTER(SynthInfo)]
        L.scfq_debug_read_file.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64]
        L.scfq_debug_read_file.restype = ctypes.c_int64
        L.scfq_debug_gz_resume.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64]
        L.scfq_debug_gz_resume.restype = ctypes.c_int64
        L.scfq_dedup_buffer.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64,
                                        ctypes.c_int, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(DedupStats)]
        L.scfq_dedup_file.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(DedupStats)]
        L.scfq_dedup_error_detail.restype = ctypes.c_char_p
        L.scfq_stage_file.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p),
                                      ctypes.POINTER(ctypes.c_uint64)]
        L.scfq_device_free.argtypes = [ctypes.c_void_p]
        L.scfq_meta_header.restype = ctypes.c_char_p
        L.scfq_meta_file_tsv.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_uint64]
        L.scfq_debug_bgzf_inflate.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64]
        L.scfq_debug_bgzf_inflate.restype = ctypes.c_int64
        L.scfq_index_lines.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64,
                                       ctypes.POINTER(ctypes.c_uint64)]
        L.scfq_debug_stream_ms.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int]
        L.scfq_debug_stream_ms.restype = ctypes.c_double
        L.scfq_synth_locate.argtypes = [ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64,
                                        ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
        L.scfq_set_wait_stream.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.scfq_get_wait_stream.argtypes = [ctypes.POINTER(ctypes.c_int)]
        L.scfq_get_wait_stream.restype = ctypes.c_void_p
        vp, pvp = ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)
        L.scfq_comm_unique_id.argtypes = [vp, ctypes.c_uint64]
        L.scfq_comm_init_rank.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, pvp]
        L.scfq_comm_init_rendezvous.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                ctypes.c_int, ctypes.c_int, pvp]
        L.scfq_comm_init_all.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_int32), ctypes.c_int, pvp]
        L.scfq_comm_world.argtypes = [vp]
        L.scfq_comm_rank.argtypes = [vp]
        L.scfq_device_bytes_now.restype = ctypes.c_uint64
        L.scfq_device_bytes_high_water.restype = ctypes.c_uint64
        L.scfq_comm_is_broken.argtypes = [vp]
        L.scfq_prepare.argtypes = [ctypes.POINTER(Opts)]
        L.scfq_comm_transport.argtypes = [vp]
        L.scfq_comm_transport.restype = ctypes.c_char_p
        L.scfq_comm_exchange.argtypes = [vp, ctypes.POINTER(Partial), vp, ctypes.POINTER(Partial), vp, ctypes.c_int]
        L.scfq_comm_exchange_start.argtypes = [vp, ctypes.POINTER(Partial), vp, ctypes.c_int]
        L.scfq_comm_exchange_finish.argtypes = [vp, ctypes.POINTER(Partial), vp, ctypes.c_int]
        L.scfq_comm_allgather_u64.argtypes = [vp, vp, ctypes.c_uint32, vp, ctypes.c_int]
        L.scfq_comm_destroy.argtypes = [vp]
        L.scfq_comm_error_detail.restype = ctypes.c_char_p
        L.scfq_count_file_sharded.argtypes = [ctypes.c_char_p, ctypes.POINTER(Opts), vp, ctypes.POINTER(Counts)]
        _lib = L
    return _lib


def strerror(rc):
    return lib().scfq_strerror(rc).decode()


def _check(rc, what):
    if rc != 0:
        raise ScfqError(rc, what, lib().scfq_last_error_detail().decode())


def make_opts(flags=0, devices=None, chunk_bytes=0, wait_stream=None):
    o = Opts()
    o.struct_size = ctypes.sizeof(Opts)
    o.flags = flags
    o.chunk_bytes = chunk_bytes
    if wait_stream is not None:          # 0 names the legacy default stream (what torch uses unless told otherwise)
        o.flags |= SCFQ_WAIT_STREAM
        o.wait_stream = wait_stream or None
    if devices:
        arr = (ctypes.c_int32 * len(devices))(*devices)
        o._keep = arr
        o.n_devices = len(devices)
        o.device_ids = ctypes.cast(arr, ctypes.POINTER(ctypes.c_int32))
    return o


def _new_counts():
    c = Counts()
    c.struct_size = ctypes.sizeof(Counts)
    return c


def _host_ptr(data):
    """bytes / bytearray / numpy uint8 array -> (address, length, keepalive)"""
    if isinstance(data, (bytes, bytearray)):
        buf = (ctypes.c_char * len(data)).from_buffer_copy(data) if len(data) else (ctypes.c_char * 1)()
        return ctypes.addressof(buf), len(data), buf
    import numpy as np
    a = np.ascontiguousarray(data, dtype=np.uint8)
    return a.ctypes.data, a.size, a


def count_file(path, flags=0, devices=None, chunk_bytes=0):
    c = _new_counts()
    o = make_opts(flags, devices, chunk_bytes)
    _check(lib().scfq_count_file(os.fsencode(path), ctypes.byref(o), ctypes.byref(c)), "scfq_count_file")
    return c


def count_host(data, flags=0, devices=None, chunk_bytes=0):
    addr, n, keep = _host_ptr(data)
    c = _new_counts()
    o = make_opts(flags, devices, chunk_bytes)
    _check(lib().scfq_count_buffer(addr, n, 0, ctypes.byref(o), ctypes.byref(c)), "scfq_count_buffer")
    return c


def count_device(dev_ptr, n, flags=0, wait_stream=None):
    """wait_stream: hipStream_t (int) of the producer of the buffer, e.g. torch.cuda.current_stream().cuda_stream; None = the
    caller has synchronised (include/sc_fqcount.h: scfq_opts.wait_stream)"""
    c = _new_counts()
    o = make_opts(flags, wait_stream=wait_stream)
    _check(lib().scfq_count_buffer(ctypes.c_void_p(dev_ptr), n, 1, ctypes.byref(o), ctypes.byref(c)),
           "scfq_count_buffer")
    return c


def partial_device(dev_ptr, n, prev_byte=-1, flags=0, want_hist=False, wait_stream=None):
    p = Partial()
    o = make_opts(flags, wait_stream=wait_stream)
    hist = (ctypes.c_uint64 * HIST_WORDS)() if want_hist else None
    _check(lib().scfq_partial_buffer(ctypes.c_void_p(dev_ptr), n, 1, prev_byte, ctypes.byref(o), ctypes.byref(p),
                                     ctypes.byref(hist) if want_hist else None), "scfq_partial_buffer")
    return (p, hist) if want_hist else p


def partial_host(data, prev_byte=-1, flags=0, want_hist=False, chunk_bytes=0):
    addr, n, keep = _host_ptr(data)
    p = Partial()
    o = make_opts(flags, None, chunk_bytes)
    hist = (ctypes.c_uint64 * HIST_WORDS)() if want_hist else None
    _check(lib().scfq_partial_buffer(addr, n, 0, prev_byte, ctypes.byref(o), ctypes.byref(p),
                                     ctypes.byref(hist) if want_hist else None), "scfq_partial_buffer")
    return (p, hist) if want_hist else p


def set_wait_stream(stream):
    """this thread's default caller stream for device-pointer calls (scfq_set_wait_stream): an int hipStream_t handle
    (0 = the legacy default stream, e.g. torch.cuda.current_stream().cuda_stream); None clears it"""
    lib().scfq_set_wait_stream(ctypes.c_void_p(stream or 0), 0 if stream is None else 1)


class Comm:
    """scfq_comm: the cross-rank exchange of shard partials inside the C library (RCCL all-gather + rank-ordered fold)"""

    def __init__(self, handle):
        self.h = ctypes.c_void_p(handle)

    @staticmethod
    def _err(rc, what):
        if rc != 0:
            raise ScfqError(rc, what, lib().scfq_comm_error_detail().decode())

    @staticmethod
    def unique_id():
        buf = ctypes.create_string_buffer(COMM_ID_BYTES)
        Comm._err(lib().scfq_comm_unique_id(buf, COMM_ID_BYTES), "scfq_comm_unique_id")
        return buf.raw

    @classmethod
    def init_rank(cls, uid, world, rank, device, timeout_ms=0):
        h = ctypes.c_void_p()
        cls._err(lib().scfq_comm_init_rank(uid, world, rank, device, timeout_ms, ctypes.byref(h)), "scfq_comm_init_rank")
        return cls(h.value)

    @classmethod
    def init_rendezvous(cls, host, port, world, rank, device=0, transport=SCFQ_COMM_RCCL, timeout_ms=0):
        h = ctypes.c_void_p()
        cls._err(lib().scfq_comm_init_rendezvous(host.encode() if host else None, port, world, rank, device, transport,
                                                 timeout_ms, ctypes.byref(h)), "scfq_comm_init_rendezvous")
        return cls(h.value)

    @classmethod
    def init_all(cls, devices, timeout_ms=0):
        n = len(devices)
        arr = (ctypes.c_int32 * n)(*devices)
        hs = (ctypes.c_void_p * n)()
        cls._err(lib().scfq_comm_init_all(n, arr, timeout_ms, hs), "scfq_comm_init_all")
        return [cls(h) for h in hs]

    world = property(lambda self: lib().scfq_comm_world(self.h))
    rank = property(lambda self: lib().scfq_comm_rank(self.h))
    transport = property(lambda self: lib().scfq_comm_transport(self.h).decode())
    broken = property(lambda self: bool(lib().scfq_comm_is_broken(self.h)))

    def exchange(self, partial, hist=None, timeout_ms=0):
        out = Partial()
        out_h = (ctypes.c_uint64 * HIST_WORDS)() if hist is not None else None
        self._err(lib().scfq_comm_exchange(self.h, ctypes.byref(partial), ctypes.byref(hist) if hist is not None else None,
                                           ctypes.byref(out), ctypes.byref(out_h) if hist is not None else None, timeout_ms),
                  "scfq_comm_exchange")
        return (out, out_h) if hist is not None else out

    def start(self, partial, hist=None, timeout_ms=0):
        self._err(lib().scfq_comm_exchange_start(self.h, ctypes.byref(partial), ctypes.byref(hist) if hist is not None else None,
                                                 timeout_ms), "scfq_comm_exchange_start")

    def finish(self, want_hist=False, timeout_ms=0):
        out = Partial()
        out_h = (ctypes.c_uint64 * HIST_WORDS)() if want_hist else None
        self._err(lib().scfq_comm_exchange_finish(self.h, ctypes.byref(out), ctypes.byref(out_h) if want_hist else None, timeout_ms),
                  "scfq_comm_exchange_finish")
        return (out, out_h) if want_hist else out

    def allgather_u64(self, words, timeout_ms=0):
        n = len(words)
        mine = (ctypes.c_uint64 * n)(*[int(w) & 0xFFFFFFFFFFFFFFFF for w in words])
        out = (ctypes.c_uint64 * (n * self.world))()
        self._err(lib().scfq_comm_allgather_u64(self.h, mine, n, out, timeout_ms), "scfq_comm_allgather_u64")
        return [list(out[r * n:(r + 1) * n]) for r in range(self.world)]

    def count_file(self, path, flags=0, devices=None, chunk_bytes=0):
        """scfq_count_file_sharded: this rank's byte range of `path`, exchange, the whole file's counters on every rank"""
        c = _new_counts()
        o = make_opts(flags, devices, chunk_bytes)
        _check(lib().scfq_count_file_sharded(os.fsencode(path), ctypes.byref(o), self.h, ctypes.byref(c)), "scfq_count_file_sharded")
        return c

    def destroy(self):
        if self.h:
            lib().scfq_comm_destroy(self.h)
            self.h = ctypes.c_void_p()


def index_lines_device(dev_ptr, n, line_off_ptr=None, cap=0):
    """K5: number of lines; with a device buffer of `cap` uint64 entries also writes the line-start offsets + sentinel"""
    lines = ctypes.c_uint64()
    _check(lib().scfq_index_lines(ctypes.c_void_p(dev_ptr), n, ctypes.c_void_p(line_off_ptr) if line_off_ptr else None, cap,
                                  ctypes.byref(lines)), "scfq_index_lines")
    return lines.value


def _new_dedup_stats():
    st = DedupStats()
    st.struct_size = ctypes.sizeof(DedupStats)
    return st


def dedup_device(dev_ptr, n, out_dev_ptr=None, out_cap=0):
    """fq-dedup of a device-resident FASTQ; result into device memory (out_dev_ptr None: size + statistics only).
    Returns (bytes_out, DedupStats)."""
    st = _new_dedup_stats()
    nb = ctypes.c_uint64()
    _check(lib().scfq_dedup_buffer(ctypes.c_void_p(dev_ptr), n, 1, ctypes.c_void_p(out_dev_ptr) if out_dev_ptr else None,
                                   out_cap, 1, ctypes.byref(nb), ctypes.byref(st)), "scfq_dedup_buffer")
    return nb.value, st


def dedup_host(data):
    """fq-dedup of a host buffer (bytes / numpy uint8); returns (bytes, DedupStats)"""
    addr, n, keep = _host_ptr(data)
    st = _new_dedup_stats()
    nb = ctypes.c_uint64()
    _check(lib().scfq_dedup_buffer(addr, n, 0, None, 0, 0, ctypes.byref(nb), ctypes.byref(st)), "scfq_dedup_buffer")
    out = (ctypes.c_uint8 * max(nb.value, 1))()
    st = _new_dedup_stats()
    _check(lib().scfq_dedup_buffer(addr, n, 0, out, nb.value, 0, ctypes.byref(nb), ctypes.byref(st)), "scfq_dedup_buffer")
    return bytes(out[:nb.value]), st


def dedup_file(path, out_fd=-1):
    st = _new_dedup_stats()
    _check(lib().scfq_dedup_file(os.fsencode(path), None, out_fd, ctypes.byref(st)), "scfq_dedup_file")
    return st


SCFQ_META_WHOLE_FILE = 0x1


def meta_header():
    return lib().scfq_meta_header().decode()


def meta_file_tsv(path, sample_n=100, flags=0):
    """the 16-column fq-meta row of `path` (reference: src/fq_meta.nim:197-278)"""
    buf = ctypes.create_string_buffer(4096)
    _check(min(0, lib().scfq_meta_file_tsv(os.fsencode(path), sample_n, flags, buf, 4096)), "scfq_meta_file_tsv")
    return buf.value.decode()


def debug_bgzf_inflate(image, cap):
    """device-side inflate of a BGZF image held in host memory (bytes) -> bytes"""
    addr, n, keep = _host_ptr(image)
    out = (ctypes.c_uint8 * max(cap, 1))()
    r = lib().scfq_debug_bgzf_inflate(addr, n, out, cap)
    if r < 0:
        raise ScfqError(int(r), "scfq_debug_bgzf_inflate", lib().scfq_last_error_detail().decode())
    return ctypes.string_at(out, r)


def hist_stats():
    """(ranges served by the speculative K3 form, ranges (re)done by the exact kernel) of this thread's last histogram session"""
    a, b = ctypes.c_uint64(), ctypes.c_uint64()
    lib().scfq_debug_hist_stats(ctypes.byref(a), ctypes.byref(b))
    return a.value, b.value


def partial_simple_device(dev_ptr, n, prev_byte=-1):
    p = Partial()
    _check(lib().scfq_debug_partial_simple(ctypes.c_void_p(dev_ptr), n, prev_byte, ctypes.byref(p)),
           "scfq_debug_partial_simple")
    return p


def debug_gz_resume(path, after_bytes, cap, chunk_bytes=0):
    """host only: a gzip file's bytes with the first member read in two halves (serial decoder, then the mid-member take-over)"""
    buf = ctypes.create_string_buffer(max(cap, 1))
    n = lib().scfq_debug_gz_resume(os.fsencode(path), after_bytes, buf, cap, chunk_bytes)
    if n < 0:
        raise ScfqError(int(n), "scfq_debug_gz_resume", lib().scfq_last_error_detail().decode())
    return buf.raw[:n]


def debug_read_file(path, cap, chunk_bytes=0):
    """host-only: the byte stream count_file would scan (plain / BGZF parallel inflate / serial gzread)"""
    buf = (ctypes.c_uint8 * max(cap, 1))()
    n = lib().scfq_debug_read_file(os.fsencode(path), buf, cap, chunk_bytes)
    if n < 0:
        raise ScfqError(int(n), "scfq_debug_read_file", lib().scfq_last_error_detail().decode())
    return bytes(buf[:n])


def debug_stream_ms(dev_ptr, n, reps=5):
    """diagnostic: ms for the scan kernel's load structure alone (4 KiB-aligned device pointer)"""
    return lib().scfq_debug_stream_ms(ctypes.c_void_p(dev_ptr), n, reps)


def debug_last_scan_kernel():
    """diagnostic: the scan kernel instance + range geometry of this thread's last launch, e.g.
    'fq_scan_tiles<false, 0, 2, true, false> tiles_per_range=100'"""
    L = lib()
    L.scfq_debug_last_scan_kernel.restype = ctypes.c_int64
    L.scfq_debug_last_scan_kernel.argtypes = [ctypes.c_char_p, ctypes.c_uint64]
    buf = ctypes.create_string_buffer(256)
    L.scfq_debug_last_scan_kernel(buf, 256)
    return buf.value.decode()


def combine(acc, b, hist_acc=None, hist_b=None):
    _check(lib().scfq_partial_combine(ctypes.byref(acc), ctypes.byref(b),
                                      ctypes.byref(hist_acc) if hist_acc is not None else None,
                                      ctypes.byref(hist_b) if hist_b is not None else None), "scfq_partial_combine")
    return acc


def identity():
    p = Partial()
    lib().scfq_partial_identity(ctypes.byref(p), None)
    return p


def finalize(p, hist=None):
    c = _new_counts()
    _check(lib().scfq_partial_finalize(ctypes.byref(p), ctypes.byref(hist) if hist is not None else None,
                                       ctypes.byref(c)), "scfq_partial_finalize")
    return c


def format_tsv(c):
    buf = ctypes.create_string_buffer(256)
    lib().scfq_format_tsv(ctypes.byref(c), buf, 256)
    return buf.value.decode()


def last_timing():
    t = Timing()
    t.struct_size = ctypes.sizeof(Timing)
    _check(lib().scfq_last_timing(ctypes.byref(t)), "scfq_last_timing")
    return t


def synth_plan(kind, seed, min_bytes, first_record=0):
    info = SynthInfo()
    info.struct_size = ctypes.sizeof(SynthInfo)
    _check(lib().scfq_synth_plan(kind, seed, first_record, min_bytes, ctypes.byref(info)), "scfq_synth_plan")
    return info


def synth_locate(kind, seed, offset):
    rec, start = ctypes.c_uint64(), ctypes.c_uint64()
    _check(lib().scfq_synth_locate(kind, seed, offset, ctypes.byref(rec), ctypes.byref(start)), "scfq_synth_locate")
    return rec.value, start.value


def synth_host(kind, seed, records, first_record=0):
    import numpy as np
    plan = SynthInfo()
    plan.struct_size = ctypes.sizeof(SynthInfo)
    # upper bound on size: plan with records via a dry call (dst NULL, cap 0 returns bytes needed)
    _check(lib().scfq_synth_host(kind, seed, first_record, records, None, 0, ctypes.byref(plan)), "scfq_synth_host")
    out = np.empty(plan.bytes, dtype=np.uint8)
    info = SynthInfo()
    info.struct_size = ctypes.sizeof(SynthInfo)
    _check(lib().scfq_synth_host(kind, seed, first_record, records, out.ctypes.data, out.size, ctypes.byref(info)),
           "scfq_synth_host")
    return out, info


def synth_device(kind, seed, records, dev_ptr, cap, first_record=0):
    info = SynthInfo()
    info.struct_size = ctypes.sizeof(SynthInfo)
    _check(lib().scfq_synth_device(kind, seed, first_record, records, ctypes.c_void_p(dev_ptr), cap,
                                   ctypes.byref(info)), "scfq_synth_device")
    return info


# ---------------------------------------------------------------------------------------------
# Python mirror of the reference operator (same names, argument meaning and error behaviour)
# ---------------------------------------------------------------------------------------------
fq_count_header = "\t".join(["reads", "gc_content", "gc_bases", "n_bases", "bases"])   # src/fq_count.nim:7-11


def output_header(header, basename, absolute):
    """src/utils/helpers.nim:200-208"""
    return "\t".join([x for x in (header, "basename" if basename else "", "absolute" if absolute else "") if x])


def get_absolute(path):
    """src/utils/helpers.nim:210-213 — symlinks: absolutePath(expandSymlink(path)) (relative targets resolve
    against the CWD, a reference quirk kept here)."""
    if os.path.islink(path):
        return os.path.abspath(os.readlink(path))
    return os.path.abspath(path)


def last_path_part(path):
    """Nim os.lastPathPart: tail of the path after stripping trailing separators."""
    return os.path.basename(path.rstrip("/")) if path.rstrip("/") else ""


def output_w_fnames(line, path, basename, absolute):
    """src/utils/helpers.nim:215-224"""
    parts = [line, last_path_part(path) if basename else "", get_absolute(path) if absolute else ""]
    return "\t".join([x for x in parts if x])


def error_msg(msg, error_code=1, stream=None):
    """src/utils/helpers.nim:29-30 (colorize fgRed)"""
    (stream or sys.stderr).write("\x1b[31mError %d: %s\x1b[0m\n" % (error_code, msg))


def quit_error(msg, error_code=1):
    """src/utils/helpers.nim:32-34"""
    error_msg(msg, error_code)
    sys.exit(error_code)


def fq_count(fastq, basename=False, absolute=False, out=None, flags=0, devices=None):
    """proc fq_count*(fastq: string, basename: bool, absolute: bool)   (src/fq_count.nim:14-53)

    Prints one TSV row. Unopenable input -> "Error 2: Unable to open file: <path>" and exit status 2
    (src/fq_count.nim:35-36). Paths shorter than 3 characters raise like the reference's
    fastq[^3 .. ^1] slice does (IndexError there; surfaces as exit 1 through sc.nim:299-305).
    """
    if len(fastq) < 3:
        raise IndexError("index out of bounds")   # fastq[^3 .. ^1], src/fq_count.nim:31
    try:
        c = count_file(fastq, flags=flags, devices=devices)
    except ScfqError as e:
        if e.rc == SCFQ_EOPEN:
            quit_error("Unable to open file: " + fastq, 2)
        raise
    (out or sys.stdout).write(output_w_fnames(format_tsv(c), fastq, basename, absolute) + "\n")
    return c
