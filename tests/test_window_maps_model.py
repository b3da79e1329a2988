"""The grouped window chain of the device gzip path (csrc/gz_inflate_kernels.hpp: gz_window_maps + gz_window_chain, three
launches) restated on a small window: what a run of segments does to the window in front of it is a MAP (byte i behind the run =
a literal, or byte j of the window in front), maps compose, so the windows in front of all segments follow from (1) every
group's map from the identity, (2) the maps applied group by group, (3) every group walked from its own window — and must equal
the plain sequential walk.  Not a test of the kernels (tests/test_gpu_gz_device.py is, with groups of 3, 5 and 64 entries)."""
import random

W = 64                                   # the window (32768 on the device)
MARK = 0x8000


def step(window, seg):
    """window behind a segment: the last W symbols of (window ++ seg), markers of seg resolved through `window`"""
    resolved = [window[s & 0x7FFF] if s & MARK else s for s in seg]
    return (window + resolved)[-W:]


def sequential(first, segs):
    wins, w = [], first
    for seg in segs:
        wins.append(w)
        w = step(w, seg)
    return wins, w


def grouped(first, segs, group):
    identity = [MARK | k for k in range(W)]
    cuts = list(range(0, len(segs), group)) + [len(segs)]
    # 1. every group's map, from the identity (markers stay markers: byte k of the window in front of the GROUP)
    maps = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        m = identity
        for seg in segs[a:b]:
            m = step(m, seg)
        maps.append(m)
    # 2. the window in front of every group: a map has the form of a segment of W symbols
    gwin, w = [], first
    for m in maps:
        gwin.append(w)
        w = step(w, m)
    last = w
    # 3. every group from its own window
    wins = []
    for (a, b), w0 in zip(zip(cuts[:-1], cuts[1:]), gwin):
        wins += sequential(w0, segs[a:b])[0]
    return wins, last


def test_grouped_walk_equals_sequential_walk():
    rng = random.Random(11)
    for trial in range(60):
        n = rng.randrange(1, 40)
        segs = []
        for _ in range(n):
            ln = rng.choice([0, 1, 5, W - 1, W, W + 3, 3 * W])          # shorter than, equal to and longer than the window
            segs.append([(MARK | rng.randrange(W)) if rng.random() < 0.4 else rng.randrange(256) for _ in range(ln)])
        first = [rng.randrange(256) for _ in range(W)]
        want, want_last = sequential(first, segs)
        for group in (1, 2, 3, 5, 64):
            got, got_last = grouped(first, segs, group)
            assert got == want and got_last == want_last, (trial, group)
