"""fq-meta (host logic, no GPU): the rows of the reference's docs/fq-meta.md:34-37 and what scripts/functional-tests.sh:115-166
asserts about sequencer / prob_sequencer for the reference's own fixtures (tests/golden == reference tests/fastq)."""
import os
import subprocess

import pytest

from conftest import GOLDEN, ROOT

SC = os.path.join(ROOT, "seq-collection_amd", "sc")

# docs/fq-meta.md:34-37 (qual_multiple is rendered TRUE there; Nim's `$true` prints "true")
DOC_ROWS = {
    "illumina_2000_2500.fq": "D00446|HiSeq2000/2500|high:machine+flowcell|C8HN4ANXX|High Output (8-lane) v4 flow cell|1|8||GCTCGGTA||Sanger;Illumina 1.8+|Phred+33|true|14|14|1",
    "illumina_3000_4000.fq": "K00100|HiSeq3000/4000|high:machine+flowcell|H300JBBXX|(8-lane) v1 flow cell|33|6||GCCAAT||Sanger;Illumina 1.8+|Phred+33|true|14|14|1",
    "illumina_6.fq": "D00209|HiSeq2000/2500|high:machine+flowcell|CACDKANXX|High Output (8-lane) v4 flow cell|258|6||CGCAGTT||Sanger;Illumina 1.8+|Phred+33|true|0|37|1",
    "illumina_7.fq": "D00209|HiSeq2000/2500|high:machine+flowcell|CACDKANXX|High Output (8-lane) v4 flow cell|258|6||GAGCAAG||Sanger;Illumina 1.8+|Phred+33|true|0|37|1",
}
# scripts/functional-tests.sh:115-166: (column 2, column 3 up to ':')
FUNCTIONAL = {
    "illumina_1.fq": ("GenomeAnalyzerIIx", "likely"), "illumina_2.fq": ("GenomeAnalyzerIIx", "likely"),
    "illumina_3.fq": ("", ""), "illumina_4.fq": ("", ""),
    "illumina_2000_2500.fq": ("HiSeq2000/2500", "high"), "illumina_3000_4000.fq": ("HiSeq3000/4000", "high"),
    "illumina_hiseq_x.fq": ("HiSeqX", "high"), "novaseq.fq": ("NovaSeq", "high"),
}


def test_docs_table_rows(scfq):
    for name, want in DOC_ROWS.items():
        assert scfq.meta_file_tsv(os.path.join(GOLDEN, name)) == want.replace("|", "\t"), name


def test_functional_test_expectations(scfq):
    for name, (sequencer, prob) in FUNCTIONAL.items():
        cols = scfq.meta_file_tsv(os.path.join(GOLDEN, name)).split("\t")
        assert len(cols) == 16
        assert (cols[1], cols[2].split(":")[0]) == (sequencer, prob), name
    assert scfq.meta_file_tsv(os.path.join(GOLDEN, "novaseq.fq")).split("\t")[2] == "high:machine+flowcell"


def test_sampling_gz_and_edge_cases(scfq, tmp_path):
    # -n limits the records looked at (fq_meta.nim:226); n_lines = records read
    assert scfq.meta_file_tsv(os.path.join(GOLDEN, "novaseq.fq"), sample_n=2).split("\t")[15] == "2"
    assert scfq.meta_file_tsv(os.path.join(GOLDEN, "novaseq.fq")).split("\t")[15] == "9"
    # .gz by (case-insensitive) suffix; same row as the plain file
    assert scfq.meta_file_tsv(os.path.join(GOLDEN, "dup.fq.gz")) == scfq.meta_file_tsv(os.path.join(GOLDEN, "dup.fq"))
    up = tmp_path / "X.FQ.GZ"
    up.write_bytes(open(os.path.join(GOLDEN, "dup.fq.gz"), "rb").read())
    assert scfq.meta_file_tsv(str(up)) == scfq.meta_file_tsv(os.path.join(GOLDEN, "dup.fq"))
    # a single-field header is the sequence id, '@' stripped on both sides; no machine / flowcell -> no sequencer guess
    cols = scfq.meta_file_tsv(os.path.join(GOLDEN, "dup.fq")).split("\t")
    assert cols[7] == "t1" and cols[0] == cols[1] == cols[2] == ""
    # empty file: every column empty except qual_multiple and n_lines
    e = tmp_path / "e.fq"
    e.write_bytes(b"")
    assert scfq.meta_file_tsv(str(e)).split("\t") == [""] * 12 + ["false", "", "", "0"]
    # most frequent index wins; CRLF line ends are stripped before parsing
    f = tmp_path / "idx.fq"
    f.write_bytes(b"".join(b"@M01234:7:000000000-A1B2C:1:1101:1:%d 1:N:0:%s\r\nACGT\r\n+\r\nIIII\r\n" % (k, bc)
                           for k, bc in enumerate([b"AAAA", b"CCCC", b"CCCC", b"TT"])))
    cols = scfq.meta_file_tsv(str(f)).split("\t")
    assert cols[0] == "M01234" and cols[1] == "MiSeq" and cols[8] == "CCCC" and cols[3] == "000000000-A1B2C"
    assert (cols[13], cols[14], cols[15]) == ("40", "40", "4")
    # a first header with two ':' fields and no '/' is an IndexError in the reference
    g = tmp_path / "bad.fq"
    g.write_bytes(b"@a:b\nA\n+\nI\n")
    with pytest.raises(scfq.ScfqError):
        scfq.meta_file_tsv(str(g))
    with pytest.raises(scfq.ScfqError) as err:
        scfq.meta_file_tsv(str(tmp_path / "missing.fq"))
    assert err.value.rc == scfq.SCFQ_EOPEN


def test_cli_fq_meta():
    r = subprocess.run([SC, "fq-meta", "-t", "-b", "-n", "5", os.path.join(GOLDEN, "illumina_6.fq")], capture_output=True, text=True)
    assert r.returncode == 0
    lines = r.stdout.splitlines()
    assert lines[0].split("\t") == ["machine", "sequencer", "prob_sequencer", "flowcell", "flowcell_description", "run", "lane",
                                    "sequence_id", "index1", "index2", "qual_format", "qual_phred", "qual_multiple", "min_qual",
                                    "max_qual", "n_lines", "basename"]
    assert lines[1] == DOC_ROWS["illumina_6.fq"].replace("|", "\t") + "\tillumina_6.fq"
    r = subprocess.run([SC, "fq-meta", "/nonexistent.fq"], capture_output=True, text=True)
    assert r.returncode == 2 and "Unable to open file: /nonexistent.fq" in r.stderr
    r = subprocess.run([SC, "fq-meta", "--header"], capture_output=True, text=True)      # header only, no files: not an error
    assert r.returncode == 0 and r.stdout.count("\n") == 1
