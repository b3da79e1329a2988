#!/usr/bin/env python3
"""Regenerates tests/golden/golden.tsv and the build-authored edge fixtures in tests/golden/edge/.

Inputs in tests/golden/*.fq, dup.fq.gz are the reference's own test data (reference tree:
tests/fastq/, MIT, (c) 2019 Daniel E Cook) copied verbatim as fixture DATA. The expected integers
are the rows of the reference's docs/fq-count.md:27-43 (typed in below, not computed); gc_content
is gc/(bases-n) under Nim 1.0.6 `$float` ("%.16g" + ".0" rule), as tabulated in BASELINE.md §2.

Edge fixtures are authored here (the reference has no fixture with N bases, CRLF, blank lines,
truncated records, empty input or multi-member gzip: "parity unpinned"); their expected values
come from `restate()` below — a pure-Python restatement of src/fq_count.nim:38-45 that is
independent of both oracle/ C restatements — and were hand-checked.
"""
import gzip, hashlib, io, os, sys

HERE = os.path.dirname(os.path.abspath(__file__))

# docs/fq-count.md:27-43  (reads, gc_content as %.16g-rule text, gc_bases, n_bases, bases)
REFERENCE_ROWS = {
    "dup.fq": (8, "0.53125", 17, 0, 32),
    "dup.fq.gz": (8, "0.53125", 17, 0, 32),
    "illumina_1.fq": (1, "0.35", 21, 0, 60),
    "illumina_2.fq": (1, "0.35", 21, 0, 60),
    "illumina_2000_2500.fq": (1, "1.0", 101, 0, 101),
    "illumina_3.fq": (6, "0.35", 126, 0, 360),
    "illumina_3000_4000.fq": (1, "1.0", 101, 0, 101),
    "illumina_4.fq": (1, "0.35", 21, 0, 60),
    "illumina_6.fq": (1, "0.35", 21, 0, 60),
    "illumina_7.fq": (1, "0.35", 21, 0, 60),
    "illumina_8.fq": (2, "0.3333333333333333", 14, 0, 42),
    "illumina_hiseq_x.fq": (1, "0.35", 21, 0, 60),
    "nodup.fq": (4, "0.5", 8, 0, 16),
    "novaseq.fq": (9, "0.0", 0, 0, 9),
    "sra.fq": (2, "0.4305555555555556", 62, 0, 144),
}


def nim_float(gc, denom):
    if denom == 0:
        return "nan"
    s = "%.16g" % (gc / denom)
    if not any(ch == "." or ch.isalpha() for ch in s):
        s += ".0"
    return s


def restate(data: bytes):
    """src/fq_count.nim:38-45 over Nim 1.0.6 readLine semantics, pure Python."""
    lines = data.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()            # no phantom line after a final '\n' (and none for empty input)
        terminated = [True] * len(lines)
    else:
        terminated = [True] * (len(lines) - 1) + [False]
    reads = gc = n = bases = 0
    for i, (ln, term) in enumerate(zip(lines, terminated), start=1):
        if term and ln.endswith(b"\r"):
            ln = ln[:-1]
        if i % 4 == 1:
            reads += 1
        if i % 4 == 2:
            gc += ln.count(b"G") + ln.count(b"C")
            n += ln.count(b"N")
            bases += len(ln)
    return reads, nim_float(gc, bases - n), gc, n, bases


REC = b"@r%d\nACGTNNGC\n+\nIIII#III\n"

EDGE = {
    # name: bytes
    "empty.fq": b"",
    "one_byte.fq": b"@",
    "only_newline.fq": b"\n",
    "n_rich.fq": b"@a\nNNNNACGTNNGGCC\n+\n!!!!!!!!!!!!!!\n@b\nNNNN\n+\n!!!!\n",
    "all_n.fq": b"@a\nNNNN\n+\n!!!!\n",
    "crlf.fq": b"@a\r\nACGTGC\r\n+\r\nIIIIII\r\n@b\r\nGGNN\r\n+\r\nIIII\r\n",
    "crlf_no_final.fq": b"@a\r\nACGTGC\r\n+\r\nIIIIII\r\n@b\r\nGGNN\r",
    "lone_cr.fq": b"@a\nAC\rGT\n+\nII\rII\n",
    "no_final_newline_seq.fq": b"@a\nACGT\n+\nIIII\n@b\nGGCCNN",
    "trunc5.fq": b"@a\nACGT\n+\nIIII\n@b\n",
    "trunc6.fq": b"@a\nACGT\n+\nIIII\n@b\nGGGN\n",
    "trunc7.fq": b"@a\nACGT\n+\nIIII\n@b\nGGGN\n+\n",
    "blank_lines.fq": b"\n\n\n\n@a\n\n+\n\n\nGC\n",
    "lowercase.fq": b"@a\nacgtnGCN\n+\nIIIIIIII\n",
    "gc_in_header_and_qual.fq": b"@GCGCNN\nAT\n+GCN\nGC\n@NGC\nTTA\n+\nNNN\n",
    "at_in_quality.fq": b"@a\nACGT\n+\n@@@@\n@b\nGGCC\n+\n+@+@\n",
    "long_line_50k.fq": b"@long\n" + (b"ACGTNGGCCA" * 5000) + b"\n+\n" + (b"5" * 50000) + b"\n",
    "many_short.fq": b"".join(REC % i for i in range(300)),
    "high_bytes.fq": b"@a\nAC\xc7\xc3GT\xce\x8a\n+\n\xff\xfe\x80\x81IIII\n",
    "nul_free_binaryish.fq": bytes(range(1, 256)) * 3 + b"\n" + bytes(range(1, 256)) + b"\n",
}


def main():
    rows = []
    for name, exp in sorted(REFERENCE_ROWS.items()):
        path = os.path.join(HERE, name)
        raw = open(path, "rb").read()
        sha = hashlib.sha256(raw).hexdigest()
        data = gzip.decompress(raw) if name.endswith(".gz") else raw
        got = restate(data)
        assert got == exp, (name, got, exp)   # the restatement reproduces the reference table
        rows.append((name, sha, "reference:docs/fq-count.md") + exp)
    os.makedirs(os.path.join(HERE, "edge"), exist_ok=True)
    edge = dict(EDGE)
    # gzip variants: two concatenated members, and plain bytes behind a .gz name (gzread passes through)
    two = gzip.compress(EDGE["many_short.fq"][:2000], mtime=0) + gzip.compress(EDGE["many_short.fq"][2000:], mtime=0)
    edge["two_member.fq.gz"] = two
    edge["not_gzip.fq.gz"] = EDGE["n_rich.fq"]
    for name, raw in sorted(edge.items()):
        with open(os.path.join(HERE, "edge", name), "wb") as f:
            f.write(raw)
        if name == "two_member.fq.gz":
            data = EDGE["many_short.fq"]
        else:
            data = raw
        exp = restate(data)
        rows.append(("edge/" + name, hashlib.sha256(raw).hexdigest(), "build:unpinned") + exp)
    with open(os.path.join(HERE, "golden.tsv"), "w") as f:
        f.write("file\tsha256\tsource\treads\tgc_content\tgc_bases\tn_bases\tbases\n")
        for r in rows:
            f.write("\t".join(str(x) for x in r) + "\n")
    print("wrote", len(rows), "rows")


if __name__ == "__main__":
    main()
