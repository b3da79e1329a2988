"""K5 line index vs numpy: line_off[j] = (position of the j-th '\\n') + 1, line 0 at 0, sentinel after the last line."""
import numpy as np
import pytest

from test_gpu_parity import to_dev, random_fastq_like

pytestmark = pytest.mark.gpu


def expected_index(a):
    nl = np.flatnonzero(a == 10).astype(np.uint64)
    starts = np.concatenate([np.zeros(1, np.uint64), nl + 1])
    if a.size and a[-1] != 10:
        starts = np.concatenate([starts, np.array([a.size + 1], np.uint64)])     # implied final '\n'
    lines = starts.size - 1
    return lines, starts


@pytest.mark.parametrize("kind", ["uniform", "ascii", "dense_nl", "sparse_nl", "crlf"])
def test_line_index_random(gpu, scfq, kind):
    torch = gpu
    rng = np.random.default_rng(len(kind))
    for n in (0, 1, 2, 63, 64, 65, 4095, 4096, 4097, 70001, 1 << 20, 3_000_001):
        for offset in (0, 1, 63, 4095):
            a = random_fastq_like(rng, n, kind)
            t, ptr = to_dev(torch, a, offset)
            lines, starts = expected_index(a)
            assert scfq.index_lines_device(ptr, n) == lines
            buf = torch.full((lines + 3,), 0x7777777777777777, dtype=torch.int64, device="cuda")
            assert scfq.index_lines_device(ptr, n, buf.data_ptr(), lines + 1) == lines
            got = buf.cpu().numpy().view(np.uint64)
            assert np.array_equal(got[:lines + 1], starts), (kind, n, offset)
            assert got[lines + 1] == 0x7777777777777777      # nothing written past the sentinel


def test_line_index_synthetic(gpu, scfq):
    torch = gpu
    for kind, seed in ((0, 20260101), (1, 20260103)):
        plan = scfq.synth_plan(kind, seed, 96 << 20)
        t = torch.empty(plan.bytes + 4096, dtype=torch.uint8, device="cuda")
        scfq.synth_device(kind, seed, plan.records, t.data_ptr(), plan.bytes)
        a = t[:plan.bytes].cpu().numpy()
        lines, starts = expected_index(a)
        assert lines == 4 * plan.records
        buf = torch.zeros(lines + 1, dtype=torch.int64, device="cuda")
        assert scfq.index_lines_device(t.data_ptr(), plan.bytes, buf.data_ptr(), lines + 1) == lines
        assert np.array_equal(buf.cpu().numpy().view(np.uint64), starts)
        # every 4th line is a header
        assert bool((a[starts[:-1:4].astype(np.int64)] == ord("@")).all())


def test_line_index_slot_boundary(gpu, scfq):
    """The compact form keeps up to 127 newline positions per 4 KiB tile; one tile with more sends the call to the mask form: the same
    index on either side of the limit (126 .. 129 newlines in one tile, the others sparse), at aligned and unaligned starts."""
    torch = gpu
    rng = np.random.default_rng(7)
    for k in (0, 1, 126, 127, 128, 129, 4096):
        for offset in (0, 17):
            a = rng.integers(65, 91, 3 * 4096 + 100, dtype=np.uint8)
            a[100] = 10
            # tile 1 of the buffer as the kernel sees it begins at byte 4096 - offset of the input
            lo = 4096 - offset
            where = rng.choice(4096, size=k, replace=False) if k < 4096 else np.arange(4096)
            a[lo + where] = 10
            t, ptr = to_dev(torch, a, offset)
            lines, starts = expected_index(a)
            buf = torch.full((lines + 2,), 0x5555555555555555, dtype=torch.int64, device="cuda")
            assert scfq.index_lines_device(ptr, a.size, buf.data_ptr(), lines + 1) == lines
            got = buf.cpu().numpy().view(np.uint64)
            assert np.array_equal(got[:lines + 1], starts), (k, offset)
            assert got[lines + 1] == 0x5555555555555555
            # a buffer that is too small gets the first `cap` entries and nothing behind them; the count is the whole input's
            cap = max(1, lines // 2)
            small = torch.full((cap + 1,), 0x5555555555555555, dtype=torch.int64, device="cuda")
            assert scfq.index_lines_device(ptr, a.size, small.data_ptr(), cap) == lines
            got = small.cpu().numpy().view(np.uint64)
            assert np.array_equal(got[:cap], starts[:cap]) and got[cap] == 0x5555555555555555, (k, offset, "small")
