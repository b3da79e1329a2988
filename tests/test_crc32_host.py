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
