"""CRC-32 by carry-less multiplication (csrc/scfq_crc32.hpp) against zlib, through the one place that exposes it without a
device: every gzip member the host readers accept has had its trailer checked with it (tests/test_inflate_host.py), and a
wrong constant would reject every valid file.  This file pins the constants themselves: they are x^n mod P, bit-reflected
and shifted left by one."""


def xpow_mod(n, poly=0x104C11DB7):
    r = 1
    for _ in range(n):
        r <<= 1
        if r >> 32:
            r ^= poly
    return r


def reflect(v, bits):
    return int("{:0{w}b}".format(v, w=bits)[::-1], 2)


def test_folding_constants_are_what_the_header_says():
    import os
    import re
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "seq-collection_amd", "csrc", "scfq_crc32.hpp")).read()
    consts = [int(x, 16) for x in re.findall(r"0x([0-9a-f]+)ll", src)]
    want = {n: reflect(xpow_mod(n), 32) << 1 for n in (544, 480, 160, 96, 64)}
    # r2r1 = (x^480, x^544), r4r3 = (x^96, x^160), r5 = x^64
    assert consts[0] == want[480] and consts[1] == want[544]
    assert consts[2] == want[96] and consts[3] == want[160]
    assert want[64] in consts
    # Barrett: u = floor(x^64 / P), and P itself, reflected over 33 bits
    poly = 0x104C11DB7
    q, r = 0, 1 << 64
    for d in range(64, 31, -1):
        if (r >> d) & 1:
            q |= 1 << (d - 32)
            r ^= poly << (d - 32)
    assert reflect(q, 33) in consts and reflect(poly, 33) in consts


# ---- the arithmetic of the DEVICE CRC kernels (bgzf_inflate's tail, gz_crc32_tiles + the host's Horner fold), restated ----------
# Reflected representation as in the kernels: bit 31 is the coefficient of x^0.
def _mulmod(a, b):
    p = 0
    for k in range(32):
        if (a >> (31 - k)) & 1:
            p ^= b
        b = (b >> 1) ^ (0xEDB88320 if b & 1 else 0)
    return p


def _xpow8(nbytes):                      # x^(8 nbytes) mod P by the table of x^(8 2^k), as crc_xp[] in bgzf_inflate
    xp, sq = [], 1 << 23                 # x^8
    for _ in range(24):
        xp.append(sq)
        sq = _mulmod(sq, sq)
    p, k = 1 << 31, 0
    while nbytes:
        if nbytes & 1:
            p = _mulmod(xp[k], p)
        nbytes >>= 1
        k += 1
    return p


def _tables():
    t0 = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ (0xEDB88320 if c & 1 else 0)
        t0.append(c)
    tabs = [t0]
    for _ in range(3):
        prev = tabs[-1]
        tabs.append([(q >> 8) ^ t0[q & 255] for q in prev])
    return tabs


def _raw_crc(data, tabs):                # zero initial value, no final inversion, four bytes per step then the tail
    c, k = 0, 0
    while k + 4 <= len(data):
        c ^= int.from_bytes(data[k:k + 4], "little")
        c = tabs[3][c & 255] ^ tabs[2][(c >> 8) & 255] ^ tabs[1][(c >> 16) & 255] ^ tabs[0][c >> 24]
        k += 4
    for b in data[k:]:
        c = tabs[0][(c ^ b) & 255] ^ (c >> 8)
    return c


def test_lane_slices_stitched_in_a_tree_give_zlib_crc32():
    """bgzf_inflate: 64 equally long slices of "pad zeros + member", raw CRC per lane, six tree steps with one shift per step,
    then crc32(M) = R(M) ^ 0xFFFFFFFF x^(8|M|) ^ 0xFFFFFFFF"""
    import random
    import zlib
    tabs, rng = _tables(), random.Random(3)
    for n in (0, 1, 3, 63, 64, 65, 255, 1000, 4097, 65280, 65536):
        data = bytes(rng.randrange(256) for _ in range(n))
        per = (((n + 63) // 64) + 3) & ~3
        pad = 64 * per - n
        c = []
        for lane in range(64):
            v0, v1 = lane * per, lane * per + per
            lo = 0 if v1 <= pad else max(v0, pad) - pad
            hi = 0 if v1 <= pad else v1 - pad
            c.append(_raw_crc(data[lo:hi], tabs))
        shift = _xpow8(per)
        for j in range(6):
            step = 1 << j
            c = [(_mulmod(c[i], shift) ^ (c[i + step] if i + step < 64 else c[i])) for i in range(64)]
            shift = _mulmod(shift, shift)
        total = c[0] ^ _mulmod(_xpow8(n), 0xFFFFFFFF) ^ 0xFFFFFFFF
        assert total == zlib.crc32(data), n


def test_tiles_and_parts_folded_by_horner_give_zlib_crc32():
    """gz_crc32_tiles + scfq_gzdev.hpp: a member's bytes arrive in parts (one per batch), every part as tiles of a virtual message
    with zeros in front; tiles fold with x^(8 tile), parts with x^(8 |part|)"""
    import random
    import zlib
    tabs, rng = _tables(), random.Random(5)
    tile = 1 << 12                       # (the kernel's tile is 1 MiB; the arithmetic does not care)
    data = bytes(rng.randrange(256) for _ in range(50_000))
    for cuts in ([0, 50_000], [0, 1, 50_000], [0, 4096, 8192, 50_000], [0, 12_345, 12_345, 40_001, 50_000]):
        raw = 0
        for a, b in zip(cuts[:-1], cuts[1:]):
            part = data[a:b]
            nt = (len(part) + tile - 1) // tile
            virt = bytes(nt * tile - len(part)) + part
            r = 0
            for t in range(nt):
                r = _mulmod(_xpow8(tile), r) ^ _raw_crc(virt[t * tile:(t + 1) * tile], tabs)
            raw = _mulmod(_xpow8(len(part)), raw) ^ r
        assert raw ^ _mulmod(_xpow8(len(data)), 0xFFFFFFFF) ^ 0xFFFFFFFF == zlib.crc32(data), cuts
