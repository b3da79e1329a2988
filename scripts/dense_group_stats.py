#!/usr/bin/env python3
"""What the boundary-first symbol loop (symbol_loop_dense) meets on the bench's FASTQ at gzip level 6: symbols per group, how often
a code needs the second table level ("hard"), sub-groups and chunks per group, rounds per group.  Host only: a serial inflate that
notes every symbol's bits and code lengths, then the grouping rules of tests/test_dense_loop_model.py applied to that sequence.
usage: dense_group_stats.py [inflated bytes, default 3e6] [profile 0|1|2]"""
import os, sys, zlib, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "seq-collection_amd", "pyhost")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import scfq
from test_lane_loop_model import LEN_BASE, LEN_EXTRA, DIST_BASE, DIST_EXTRA, CL_ORDER

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3_000_000
prof = int(sys.argv[2]) if len(sys.argv) > 2 else 0
plan = scfq.synth_plan(prof, 20260101, n)
data, _ = scfq.synth_host(prof, 20260101, plan.records)
co = zlib.compressobj(6, zlib.DEFLATED, -15)
raw = co.compress(data.tobytes()) + co.flush()

class R:
    def __init__(s, b): s.b = b; s.p = 0
    def peek(s, n):
        i = s.p >> 3; v = int.from_bytes(s.b[i:i + 8], "little") >> (s.p & 7); return v & ((1 << n) - 1)
    def take(s, n): v = s.peek(n); s.p += n; return v

def table(lens):
    count = [0] * 16
    for l in lens: count[l] += 1
    count[0] = 0; code = 0; nxt = [0] * 16
    for l in range(1, 16): code = (code + count[l - 1]) << 1; nxt[l] = code
    t = {}
    for sym, l in enumerate(lens):
        if l: t[(l, int(format(nxt[l], "0%db" % l)[::-1], 2))] = sym; nxt[l] += 1
    return t
def code(t, r):
    v = r.peek(15)
    for l in range(1, 16):
        s = t.get((l, v & ((1 << l) - 1)))
        if s is not None: r.p += l; return s, l
    raise ValueError
r = R(raw); blocks = []
while True:
    last, kind = r.take(1), r.take(2)
    assert kind == 2, kind
    hlit, hdist, hclen = r.take(5) + 257, r.take(5) + 1, r.take(4) + 4
    cl = [0] * 19
    for i in range(hclen): cl[CL_ORDER[i]] = r.take(3)
    clt = table(cl); lens = []
    while len(lens) < hlit + hdist:
        s, _ = code(clt, r)
        if s < 16: lens.append(s)
        elif s == 16: lens += [lens[-1]] * (3 + r.take(2))
        elif s == 17: lens += [0] * (3 + r.take(3))
        else: lens += [0] * (11 + r.take(7))
    lit, dist = table(lens[:hlit]), table(lens[hlit:])
    syms = []        # (bit position, total bits, out length, distance, hard)
    while True:
        p0 = r.p
        s, l = code(lit, r)
        if s < 256: syms.append((p0, l, 1, 0, l > 10)); continue
        if s == 256: syms.append((p0, l, 0, 0, True)); break          # the end-of-block code always ends part A (hard or not)
        ln = LEN_BASE[s - 257] + r.take(LEN_EXTRA[s - 257])
        d, dl = code(dist, r)
        off = DIST_BASE[d] + r.take(DIST_EXTRA[d])
        syms.append((p0, r.p - p0, ln, off, l > 10 or dl > 8))
    blocks.append(syms)
    if last: break
st = dict(blocks=len(blocks), symbols=0, out=0, hard=0, groups=0, rounds=0, subgroups=0, chunks=0, alone=0, lit=0, f_chunks=0, f_stalls=0, f_alone=0)
for syms in blocks:
    i = 0; n_s = len(syms); g = []
    while i < n_s or g:
        # part A: rounds of 128 bit positions from the first symbol not yet collected, until 64 symbols are there or a hard one is
        hard = bool(g) and g[-1][4]
        if i < n_s and not hard: rel = syms[i][0]
        while len(g) < 64 and i < n_s and not hard:
            st["rounds"] += 1
            end = rel + 128
            while i < n_s and syms[i][0] < end:
                g.append(syms[i]); i += 1
                if g[-1][4]: hard = True; break
                rel = g[-1][0] + g[-1][1]
        grp, g = g[:64], g[64:]                          # (what is beyond 64 is carried into the next group)
        st["groups"] += 1; st["symbols"] += len(grp)
        stt = 0; starts = []
        for s in grp: starts.append(stt); stt += s[2]
        st["out"] += stt; st["hard"] += sum(1 for s in grp if s[4]); st["lit"] += sum(1 for s in grp if s[2] == 1)
        # the same group in chunks of 64 output symbols that do not restart at symbol boundaries: a chunk ends in front of the first lane that
        # reads the chunk's own output or belongs to a match overlapping itself (copied alone); it waits for the chunk before ("stall") only
        # when a lane reads what that chunk puts out
        owner = [k for k, sy in enumerate(grp) for _ in range(sy[2])]
        base = 0; pend = (-1, -1)
        while base < stt:
            k = owner[base]
            if grp[k][3] and grp[k][3] < grp[k][2]:                    # overlaps its own output
                assert starts[k] == base
                st["f_alone"] += 1; base += grp[k][2]; pend = (-1, -1); continue
            c = 0; stall = False
            while c < 64 and base + c < stt:
                k = owner[base + c]; d = grp[k][3]
                if d:
                    if d < grp[k][2] or d <= c: break
                    if pend[0] <= base + c - d < pend[1]: stall = True
                c += 1
            assert c > 0
            st["f_chunks"] += 1; st["f_stalls"] += stall
            pend = (base, base + c); base += c
        k0, s0 = 0, 0
        while k0 < len(grp):
            kc = next((k for k in range(k0, len(grp)) if grp[k][3] and grp[k][3] < starts[k] + grp[k][2] - s0), len(grp))
            if kc == k0: st["alone"] += 1; s0 += grp[k0][2]; k0 += 1; continue
            s1 = starts[kc] if kc < len(grp) else stt
            st["subgroups"] += 1; st["chunks"] += (s1 - s0 + 63) // 64
            k0, s0 = kc, s1
g = st["groups"]
print(json.dumps(dict(st, profile=prof, inflated=len(data), deflated=len(raw), symbols_per_group=round(st["symbols"] / g, 1), out_per_group=round(st["out"] / g, 1),
                      rounds_per_group=round(st["rounds"] / g, 2), subgroups_per_group=round(st["subgroups"] / g, 2), chunks_per_group=round(st["chunks"] / g, 2),
                      alone_per_group=round(st["alone"] / g, 2), fixed_chunks_per_group=round(st["f_chunks"] / g, 2), fixed_stalls_per_group=round(st["f_stalls"] / g, 2), hard_share=round(st["hard"] / st["symbols"], 4), literal_share=round(st["lit"] / st["symbols"], 3))))
