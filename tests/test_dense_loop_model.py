"""The rules of the boundary-first symbol loop (csrc/bgzf_inflate_kernel.hpp: symbol_loop_dense) restated in Python and run against
zlib.  Part A looks only at the FIRST table level (10 bits of a literal/length code, 8 of a distance code) to learn how long the code
at each of 128 bit positions (two per lane) is, follows the chain and collects the positions of its symbols; a code of the second level, a length
code whose distance code is one, and the end-of-block code end the chain there ("hard": collected, decoded in part B, which says where
the chain goes on).  Part B decodes up to 64 collected symbols at once, puts them at the prefix sum of their lengths and writes the
output in SUB-GROUPS — a match that reads output of its own sub-group starts the next one, one that overlaps its own output is copied
alone — in chunks of 64 symbols whose every lane finds its owner: symbols start in order, so it is the last symbol that started in front
of the chunk plus the number of starts at or below the lane.  The memory model is the kernel's: a chunk's symbols reach memory when the
next chunk's load has been issued, or before the first load of the next sub-group; a load sees exactly what has been stored.  Not a test
of the kernel
(tests/test_gpu_bgzf_device.py and test_gpu_gz_device.py run it on the device with SCFQ_INFLATE_LOOP=dense)."""
import random
import zlib

from test_lane_loop_model import Bits, CL_ORDER, EOB, INVALID, LIT, MATCH, canonical, code_at, fastq, symbol_at, LEN_EXTRA

LIT_ROOT, DIST_ROOT = 10, 8


def length_at(bits, at, lit, dist):
    """part A, one lane: ("go", bits) for a symbol the chain steps over, ("hard",), ("eob", bits) or ("bad",)"""
    s, n = code_at(lit, bits, at)
    if s is None or s > 285:
        return ("bad",)
    if n > LIT_ROOT:
        return ("hard",)
    if s < 256:
        return ("go", n)
    if s == 256:
        return ("eob", n)
    t = n + LEN_EXTRA[s - 257]
    d, dn = code_at(dist, bits, at + t)
    if d is None or d > 29 or dn > DIST_ROOT:
        return ("hard",)                                 # part B finds the distance, or that there is none
    from test_lane_loop_model import DIST_EXTRA
    return ("go", t + dn + DIST_EXTRA[d])


def huffman_block_dense(bits, lit, dist, mem, stats):
    pos = len(mem)
    pending = []                                          # (offset, value): the chunk whose store has not been issued yet

    def flush():
        for off, v in pending:
            assert off <= len(mem)
            if off == len(mem):
                mem.append(v)
            else:
                mem[off] = v
        pending.clear()

    P, endk, rel, stop = [], 0, bits.p, False             # collected positions | 0 chain goes on, 1 ended (rel behind the end-of-block code), 2 last one is hard
    while not stop:
        # ---- A
        while len(P) < 64 and endk == 0:                  # (P holds up to 63 + 128 positions)
            lanes = [length_at(bits, rel + i, lit, dist) for i in range(128)]      # (two positions per lane: i and 64 + i)
            cur, chain = 0, []
            while cur < 128:
                a = lanes[cur]
                chain.append(cur)                         # (the walk notes a lane before it knows whether it can step over it)
                if a[0] != "go":
                    break
                cur += a[1]
            stats["rounds"] += 1
            if a[0] != "go":
                sl = chain[-1]
                if a[0] == "hard":
                    endk = 2
                    stats["hard"] += 1
                else:
                    assert a[0] == "eob", "a code that is not assigned"
                    chain.pop()
                    endk, nxt = 1, rel + sl + a[1]
            else:
                nxt = rel + cur
            P += [rel + i for i in chain]
            if endk != 2:
                rel = nxt
        # ---- B
        m = min(len(P), 64)
        if m:
            sym = [symbol_at(bits, p, lit, dist) for p in P[:m]]
            ln = [s[1] if s[0] == MATCH else (1 if s[0] == LIT else 0) for s in sym]
            st = [sum(ln[:k]) for k in range(m)]
            R = sum(ln)
            tail = endk == 2 and len(P) <= 64
            for k, s in enumerate(sym):
                assert s[0] != INVALID
                assert s[0] != EOB or (tail and k == m - 1)
                assert s[0] != MATCH or 1 <= s[2] <= pos + st[k], "a distance that is none, or too far back"
            stats["groups"] += 1
            stats["symbols"] += m
            k0, s0 = 0, 0
            while k0 < m:
                cut = [k for k in range(k0, m) if sym[k][0] == MATCH and sym[k][2] < st[k] + ln[k] - s0]
                kc = cut[0] if cut else m
                if kc == k0:                              # overlaps its own output: alone, with its period
                    stats["alone"] += 1
                    flush()
                    _, length, off, _ = sym[k0]
                    assert off < length and s0 == st[k0]
                    for k in range(length):
                        mem.append(mem[pos + s0 - off + (0 if off == 1 else k % off)])
                    k0, s0 = k0 + 1, s0 + length
                    continue
                s1 = st[kc] if kc < m else R
                stats["subgroups"] += 1
                ka, first = k0, True                      # ka: symbols that start in front of the chunk
                for base in range(s0, s1, 64):
                    stats["chunks"] += 1
                    # symbols start in order: a lane's owner is symbol ka - 1 plus the number of starts at or below the lane
                    starts = [any(ln[k] and st[k] == base + t for k in range(k0, kc)) for t in range(64)]
                    own = [ka - 1 + sum(starts[:t + 1]) for t in range(64)]
                    ka += sum(starts)
                    if first:
                        flush()
                    chunk = []
                    for lane in range(64):
                        t = base + lane
                        if t >= s1:
                            continue
                        kind, val, off, _ = sym[own[lane]]
                        assert st[own[lane]] <= t < st[own[lane]] + ln[own[lane]]
                        if kind == MATCH:
                            src = pos + t - off
                            assert 0 <= src < len(mem), "a load of memory that is not there yet"
                            chunk.append((pos + t, mem[src]))
                        else:
                            chunk.append((pos + t, val))
                    if not first:
                        flush()
                    pending.extend(chunk)
                    first = False
                k0, s0 = kc, s1
            pos += R
            if tail:
                kind, _, _, nbits = sym[m - 1]
                rel = P[m - 1] + nbits
                if kind == EOB:
                    stop = True
                endk = 0
        if len(P) > 64:
            P = P[64:]
        else:
            P = []
            if endk == 1:
                stop = True
    flush()
    bits.p = rel
    assert pos == len(mem)


def inflate_dense(raw, stats):
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
            huffman_block_dense(bits, lit, dist, out, stats)
        if last:
            return bytes(out)


def new_stats():
    return {"rounds": 0, "groups": 0, "symbols": 0, "hard": 0, "subgroups": 0, "chunks": 0, "alone": 0}


def test_groups_reproduce_inflate():
    rng = random.Random(11)
    text = bytes(rng.choice(b"abcdefghijklmnopqrstuvwxyz ABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789.,;:!?\n") for _ in range(6000))
    corpora = {
        "fastq": fastq(rng, 120),
        "runs and short periods": b"A" * 3000 + fastq(rng, 20) + b"xyz" * 300 + b"ACGTN" * 200,
        "long matches": (fastq(rng, 4) * 6),
        "two symbols": bytes(rng.choice(b"AB") for _ in range(4000)),
        "a wide alphabet (codes of the second level)": text + bytes(rng.randrange(256) for _ in range(3000)) + text[:2000],
        "empty": b"",
        "one byte": b"a",
    }
    total = new_stats()
    for name, data in corpora.items():
        for level in (1, 6, 9):
            for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_RLE):
                co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
                raw = co.compress(data) + co.flush()
                stats = new_stats()
                assert inflate_dense(raw, stats) == data, (name, level, strategy)
                for k in total:
                    total[k] += stats[k]
    assert total["groups"] > 300 and total["hard"] > 50 and total["alone"] > 0 and total["subgroups"] > total["groups"]
