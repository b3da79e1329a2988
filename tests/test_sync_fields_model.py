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


# ---- the Kraft sum of the code-length code without a loop (sync_kraft, r5) ---------------------------------------------------------
def kraft_loop(w):
    """what the round-2 kernel did: w = the 64 bits at position + 13 (HCLEN, then HCLEN + 4 three-bit lengths)"""
    hclen = (w & 15) + 4
    w2 = w >> 4
    k = 0
    for j in range(hclen):
        l = (w2 >> (3 * j)) & 7
        k += (128 >> l) if l else 0
    return k == 128


def kraft_swar(w):
    hclen = (w & 15) + 4
    fm = 0x1249249249249249 & ((1 << (3 * hclen)) - 1)
    w2 = w >> 4
    b0, b1, b2 = w2 & fm, (w2 >> 1) & fm, (w2 >> 2) & fm
    n0, n1, n2 = ~b0 & M64, ~b1 & M64, ~b2 & M64
    pc = lambda v: bin(v).count("1")
    k = (64 * pc(b0 & n1 & n2) + 32 * pc(n0 & b1 & n2) + 16 * pc(b0 & b1 & n2) + 8 * pc(n0 & n1 & b2) + 4 * pc(b0 & n1 & b2) +
         2 * pc(n0 & b1 & b2) + pc(b0 & b1 & b2))
    return k == 128


def test_kraft_sum_by_population_counts():
    rng = random.Random(5)
    hits = 0
    for _ in range(200000):
        w = rng.getrandbits(64)
        assert kraft_loop(w) == kraft_swar(w), hex(w)
        hits += kraft_swar(w)
    assert hits > 100          # complete codes do turn up among chance bits (about one position in a hundred of those that get here)
    # complete codes built on purpose, every HCLEN: lengths whose weights sum to 128, padded with zeros
    for hclen in range(4, 20):
        for _ in range(300):
            lens, left = [], 128
            while left and len(lens) < hclen:
                l = rng.choice([v for v in range(1, 8) if (128 >> v) <= left])
                lens.append(l); left -= 128 >> l
            lens += [0] * (hclen - len(lens))
            rng.shuffle(lens)
            w = (hclen - 4) | sum(l << (4 + 3 * j) for j, l in enumerate(lens)) | (rng.getrandbits(64) << (4 + 3 * hclen))
            w &= M64
            assert kraft_swar(w) == (left == 0) == kraft_loop(w), (hclen, lens)


# ---- the code-length code as a 128-entry table indexed MSB-first (sync_deep_tab, r5) --------------------------------------------------
CL_ORDER = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]       # RFC 1951 3.2.7
CL_FIELD = [3, 17, 15, 13, 11, 9, 7, 5, 4, 6, 8, 10, 12, 14, 16, 18, 0, 1, 2]       # the kernel's kClField


def table_msb_first(lens19):
    """the kernel's construction: first code per length from the counts, symbols in ascending order, a symbol's entries one aligned run"""
    cnt = [0] * 8
    for l in lens19:
        cnt[l] += 1
    nc = [0] * 8
    for l in range(2, 8):
        nc[l] = (nc[l - 1] + cnt[l - 1]) << 1
    tab = [None] * 128
    for sy in range(19):
        l = lens19[sy]
        if l:
            code = nc[l]; nc[l] += 1
            run, at = 128 >> l, code << (7 - l)
            assert at % run == 0
            for o in range(run):
                assert tab[at + o] is None
                tab[at + o] = (sy << 3) | l
    return tab


def canonical_decode(lens19, bits):
    """RFC 1951 3.2.2, bit by bit: (symbol, length) of the code the stream `bits` (first bit = bit 0) starts with"""
    cnt = [0] * 8
    for l in lens19:
        cnt[l] += 1
    cnt[0] = 0
    code, first, index = 0, 0, 0
    order = sorted((l, s) for s, l in enumerate(lens19) if l)
    for ln in range(1, 8):
        code |= bits & 1
        bits >>= 1
        if code - first < cnt[ln]:
            return order[index + code - first][1], ln
        index += cnt[ln]
        first = (first + cnt[ln]) << 1
        code <<= 1
    return None


def test_code_length_table_matches_the_canonical_decoder():
    assert [CL_ORDER[f] for f in CL_FIELD] == list(range(19)) and sorted(CL_FIELD) == list(range(19))
    rng = random.Random(11)
    for _ in range(400):
        # a complete code over a random subset of the 19 symbols
        lens, left = [], 128
        while left:
            l = rng.choice([v for v in range(1, 8) if (128 >> v) <= left])
            lens.append(l); left -= 128 >> l
            if len(lens) == 19 and left:
                lens, left = [], 128
        lens += [0] * (19 - len(lens))
        rng.shuffle(lens)
        tab = table_msb_first(lens)
        assert all(e is not None for e in tab)                 # the runs tile the table exactly
        for _ in range(200):
            x = rng.getrandbits(32)
            v = int("{:032b}".format(x)[::-1], 2) >> 25          # __builtin_bitreverse32(x) >> 25: the next 7 stream bits, first bit on top
            e = tab[v]
            assert (e >> 3, e & 7) == canonical_decode(lens, x), (lens, hex(x))
