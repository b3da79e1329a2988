"""Helper run in a subprocess by test_gpu_variants.py with tuning knobs set in the environment: random buffers
through the HIP path vs the oracle (partials, bit-exact)."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "seq-collection_amd", "pyhost"))
import torch
import scfq

O = ctypes.CDLL(os.path.join(ROOT, "oracle", "libfqcount_oracle.so"))
O.oracle_partial.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64), ctypes.c_void_p]
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
alphabets = [b"ACGTN@+FI#:,\r\n\n", b"ACGTN" * 20 + b"\n", bytes(range(256)), b"\n\nG\r"]
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
for case in range(n_cases):
    n = int(rng.integers(0, 600_000)) if case % 5 else int(rng.integers(0, 200))
    off = int(rng.integers(0, 4096))
    a = rng.choice(np.frombuffer(alphabets[case % 4], dtype=np.uint8), n).astype(np.uint8)
    t = torch.full((off + n + 8192,), 0x47, dtype=torch.uint8, device="cuda")
    base = ((t.data_ptr() + 4095) // 4096) * 4096 - t.data_ptr() + off
    t[base:base + n] = torch.from_numpy(a)
    torch.cuda.synchronize()
    prev = int(rng.choice([-1, 10, 13, 65]))
    w = (ctypes.c_uint64 * 27)()
    h = (ctypes.c_uint64 * 1024)()
    O.oracle_partial(a.ctypes.data if n else None, n, prev, w, ctypes.byref(h))
    flags = [0, scfq.SCFQ_STRUCT_CHECK, scfq.SCFQ_QUAL_HIST | scfq.SCFQ_STRUCT_CHECK][case % 3]
    if flags & scfq.SCFQ_QUAL_HIST:
        p, hh = scfq.partial_device(t.data_ptr() + base, n, prev, flags=flags | scfq.SCFQ_HIST_EXACT, want_hist=True)
        assert list(hh) == list(h), ("hist", case, n, off)
        p, hh = scfq.partial_device(t.data_ptr() + base, n, prev, flags=flags, want_hist=True)
        ks = range(4) if p.hist_class == 0 else [p.hist_class - 1]
        for k in ks:
            assert list(hh)[k * 256:(k + 1) * 256] == list(h)[k * 256:(k + 1) * 256], ("spec hist", case, n, off, k)
    else:
        p = scfq.partial_device(t.data_ptr() + base, n, prev, flags=flags)
    want = list(w)[:25] if flags & scfq.SCFQ_STRUCT_CHECK else list(w)[:13] + [0] * 12
    assert p.words()[:25] == want, (case, n, off, prev, flags)
print("variant ok", os.environ.get("SCFQ_RING"), os.environ.get("SCFQ_NT"), os.environ.get("SCFQ_TILES_PER_RANGE"))
