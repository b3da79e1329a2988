"""The bit-parallel field tests of the block-start search (csrc/gz_inflate_kernels.hpp: sync_fields32) restated: for 32 consecutive
bit positions at once, "BTYPE is 10, HLIT <= 29, HDIST <= 29" is a handful of shifts and ANDs of the 64 bits at the first
position — and must agree with reading the three fields position by position."""
import random

M64 = (1 << 64) - 1


def fields32(x):
    m = ((~x & M64) >> 1) & (x >> 2)
    m &= ~((x >> 4) & (x >> 5) & (x >> 6) & (x >> 7)) & M64
    m &= ~((x >> 9) & (x >> 10) & (x >> 11) & (x >> 12)) & M64
    return m & 0xFFFFFFFF


def one_position(x, i):
    w = x >> i
    return ((w >> 1) & 3) == 2 and ((w >> 3) & 31) <= 29 and ((w >> 8) & 31) <= 29


def test_fields_of_32_positions_at_once():
    rng = random.Random(2)
    seen = 0
    for _ in range(4000):
        x = rng.getrandbits(64)
        m = fields32(x)
        for i in range(32):
            assert bool((m >> i) & 1) == one_position(x, i), (hex(x), i)
        seen += bin(m).count("1")
    assert 0.15 < seen / (4000 * 32) < 0.30        # about a fifth of the positions go on to the Kraft sum
