"""The ROUND algorithm of the lane-parallel symbol loop (csrc/bgzf_inflate_kernel.hpp: symbol_loop_lanes) restated step by step
in Python and run against zlib: decode a symbol at every one of 64 bit positions, follow the chain of real ones, prefix-sum
their output lengths, end the round in front of the first symbol it cannot take (a match that reads the round's own output,
reaches before the start of the output, or ends beyond 64 bytes), produce every output byte from the symbol noted at the
highest start at or below it — with the kernel's memory model: a round's bytes reach memory at the START of the next round's
output step, loads see exactly what has been stored before them.  Not a test of the kernel (tests/test_gpu_bgzf_device.py and
test_gpu_gz_device.py are, on the device): a check that the rules the kernel implements reproduce inflate, on any machine."""
import random
import zlib

LEN_BASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
LEN_EXTRA = [0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0]
DIST_BASE = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289,
             16385, 24577]
DIST_EXTRA = [0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13]
CL_ORDER = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]
LIT, MATCH, EOB, INVALID = 0, 1, 2, 3


class Bits:
    def __init__(self, data):
        self.v = int.from_bytes(data, "little")
        self.p = 0

    def peek(self, n, at=None):
        return (self.v >> (self.p if at is None else at)) & ((1 << n) - 1)

    def take(self, n):
        v = self.peek(n)
        self.p += n
        return v


def canonical(lens):
    count = [0] * 16
    for ln in lens:
        count[ln] += 1
    count[0] = 0
    code, nxt = 0, [0] * 16
    for ln in range(1, 16):
        code = (code + count[ln - 1]) << 1
        nxt[ln] = code
    table = {}
    for sym, ln in enumerate(lens):
        if ln:
            table[(ln, int(format(nxt[ln], "0%db" % ln)[::-1], 2))] = sym
            nxt[ln] += 1
    return table


def code_at(table, bits, at):
    for ln in range(1, 16):
        sym = table.get((ln, bits.peek(ln, at)))
        if sym is not None:
            return sym, ln
    return None, 0


def symbol_at(bits, at, lit, dist):
    """what a lane decodes: (kind, literal | length, distance, bits) of the symbol that would start at bit `at`"""
    s, n = code_at(lit, bits, at)
    if s is None or s > 285:
        return INVALID, 0, 0, n
    if s < 256:
        return LIT, s, 0, n
    if s == 256:
        return EOB, 0, 0, n
    lx = LEN_EXTRA[s - 257]
    length = LEN_BASE[s - 257] + bits.peek(lx, at + n)
    t = n + lx
    d, dn = code_at(dist, bits, at + t)
    if d is None or d > 29:
        return MATCH, length, 0, t                       # the distance of a code that is not assigned is 0: caught by the round's cut
    dx = DIST_EXTRA[d]
    return MATCH, length, DIST_BASE[d] + bits.peek(dx, at + t + dn), t + dn + dx


def huffman_block_in_rounds(bits, lit, dist, mem, stats):
    """one Huffman-coded block, the way the kernel does it; mem: the output so far (bytearray), extended in place"""
    pos, pending = len(mem), []

    def store_pending():
        for off, v in pending:
            assert off == len(mem) or off < len(mem)
            if off == len(mem):
                mem.append(v)
            else:
                mem[off] = v
        pending.clear()

    while True:
        lanes = [symbol_at(bits, bits.p + i, lit, dist) for i in range(64)]
        # the chain: lane 0, then wherever each symbol ends (the scalar unit's part)
        cur, chain, a = 0, [], None
        while True:
            a = lanes[cur]
            if a[0] >= EOB:
                break
            chain.append(cur)
            cur += a[3]
            if cur >= 64:
                break
        stopped = a[0] >= EOB
        # output offsets: prefix sum of the output lengths over the chain lanes
        on = [i in chain for i in range(64)]
        ln = [(lanes[i][1] if lanes[i][0] == MATCH else 1) if on[i] else 0 for i in range(64)]
        st, run = [], 0
        for i in range(64):
            st.append(run)
            run += ln[i]
        # the round ends in front of the first symbol it cannot take
        cut = [on[i] and ((lanes[i][0] == MATCH and (st[i] + ln[i] > lanes[i][2] or (lanes[i][2] - 1) % (1 << 32) >= pos + st[i]))
                          or st[i] + ln[i] > 64) for i in range(64)]
        stop, alone, c = False, False, 64
        if not any(cut):
            n_out, advance = run, (cur + a[3] if stopped else cur)
            if stopped:
                assert a[0] == EOB, "a code that is not assigned"
                stop = True
        else:
            c = cut.index(True)
            n_out, advance, alone = st[c], c, c == 0
        store_pending()                                  # the round before reaches memory now
        noted = [None] * 64                              # by output byte: the symbol that starts there
        for i in range(64):
            if on[i] and i < c:
                noted[st[i]] = lanes[i]
        for t in range(n_out):
            kind, val, off, _ = noted[max(j for j in range(t + 1) if noted[j])]
            if kind == MATCH:
                src = pos + t - off                      # a match byte is `distance` bytes back
                assert 0 <= src < len(mem), "the round's one load reads memory that is not there yet"
                pending.append((pos + t, mem[src]))
            else:
                pending.append((pos + t, val))
        pos += n_out
        stats["rounds"] += 1
        if alone:                                        # the first symbol overlaps its own output or is longer than 64: the serial way
            stats["alone"] += 1
            kind, length, off, nbits = lanes[0]
            assert off - 1 < pos, "distance too far back"
            assert not pending
            for k in range(length):
                j = k if off >= length else (0 if off == 1 else k % off)
                mem.append(mem[pos - off + j])
            pos += length
            advance = nbits
        bits.p += advance
        if stop:
            break
    store_pending()
    assert pos == len(mem)


def inflate_in_rounds(raw, stats):
    bits, out = Bits(raw), bytearray()
    while True:
        last, kind = bits.take(1), bits.take(2)
        if kind == 0:
            bits.p = (bits.p + 7) & ~7
            n = bits.take(16)
            bits.take(16)
            out += raw[bits.p >> 3:(bits.p >> 3) + n]
            bits.p += 8 * n
        else:
            if kind == 1:
                lit, dist = canonical([8] * 144 + [9] * 112 + [7] * 24 + [8] * 8), canonical([5] * 32)
            else:
                hlit, hdist, hclen = bits.take(5) + 257, bits.take(5) + 1, bits.take(4) + 4
                cl = [0] * 19
                for i in range(hclen):
                    cl[CL_ORDER[i]] = bits.take(3)
                clt, lens = canonical(cl), []
                while len(lens) < hlit + hdist:
                    s, n = code_at(clt, bits, bits.p)
                    bits.p += n
                    if s < 16:
                        lens.append(s)
                    elif s == 16:
                        lens += [lens[-1]] * (3 + bits.take(2))
                    elif s == 17:
                        lens += [0] * (3 + bits.take(3))
                    else:
                        lens += [0] * (11 + bits.take(7))
                lit, dist = canonical(lens[:hlit]), canonical(lens[hlit:])
            huffman_block_in_rounds(bits, lit, dist, out, stats)
        if last:
            return bytes(out)


def fastq(rng, n):
    recs = []
    for i in range(n):
        ln = rng.choice([50, 100, 150])
        recs.append("@r%d x/%d\n%s\n+\n%s\n" % (i, i % 3, "".join(rng.choice("ACGTN") for _ in range(ln)), "".join(rng.choice("IIIIIFFFF#@:,") for _ in range(ln))))
    return "".join(recs).encode()


def test_rounds_reproduce_inflate():
    rng = random.Random(7)
    corpora = {
        "fastq": fastq(rng, 120),
        "runs and short periods": b"A" * 3000 + fastq(rng, 20) + b"xyz" * 300 + b"ACGTN" * 200,
        "long matches": (fastq(rng, 4) * 6),
        "two symbols": bytes(rng.choice(b"AB") for _ in range(4000)),
        "empty": b"",
        "one byte": b"a",
    }
    total = {"rounds": 0, "alone": 0}
    for name, data in corpora.items():
        for level in (1, 6, 9):
            for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_RLE):
                co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
                raw = co.compress(data) + co.flush()
                stats = {"rounds": 0, "alone": 0}
                assert inflate_in_rounds(raw, stats) == data, (name, level, strategy)
                for k in total:
                    total[k] += stats[k]
    assert total["rounds"] > 1000 and 0 < total["alone"] < total["rounds"]
