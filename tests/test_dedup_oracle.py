"""fq-dedup restatement (oracle/fqdedup_oracle.c) against what the reference's own tests pin
(scripts/functional-tests.sh:86-92: dup.fq and dup.fq.gz -> 4 lines containing '@') and against the behaviours of
src/fq_dedup.nim:42-73 spelled out case by case. CPU only."""
import gzip
import os

from conftest import GOLDEN


def test_reference_functional_test_dup_fq(oracle):
    raw = open(os.path.join(GOLDEN, "dup.fq"), "rb").read()
    gz = gzip.open(os.path.join(GOLDEN, "dup.fq.gz"), "rb").read()
    assert raw == gz
    out, st = oracle.dedup(raw)
    # assert_equal 4 "$(grep -c '@' "${STDOUT_FILE}")"
    assert sum(1 for line in out.split(b"\n") if b"@" in line) == 4
    assert (st.total_reads, st.duplicates, st.records_out) == (8, 4, 4)
    # first occurrence of every ID survives, in file order, with ITS OWN sequence / quality lines
    assert out == b"@t1\nAGGA\n+\nAAAA\n@t2\nAGGA\n+\nAAAA\n@t3\nAGGA\n+\nAAAA\n@t4\nAGGA\n+\nAAJA\n"


def test_no_duplicates_is_a_copy(oracle):
    raw = open(os.path.join(GOLDEN, "nodup.fq"), "rb").read()
    out, st = oracle.dedup(raw)
    assert out == (raw if raw.endswith(b"\n") else raw + b"\n")     # echo adds the final newline
    assert st.duplicates == 0 and st.total_reads == 4


def test_echo_semantics(oracle):
    # "\r\n" is stripped by readLine and echo writes "\n"; the ID compare is on the stripped line
    out, st = oracle.dedup(b"@a\r\nAC\r\n+\r\nII\r\n@a\nGG\n+\n##\n@b\r\nTT\r\n+\r\nII")
    assert out == b"@a\nAC\n+\nII\n@b\nTT\n+\nII\n"
    assert (st.total_reads, st.duplicates) == (3, 1)      # 12 lines div 4 (the last line has no '\n' and still counts)
    # a lone '\r' at the end of the input is kept (no '\n' follows it)
    out, st = oracle.dedup(b"@a\nAC\n+\nII\n@a\r")
    assert out == b"@a\nAC\n+\nII\n@a\r\n" and st.duplicates == 0
    # empty input, blank lines, IDs that differ only in trailing blanks
    assert oracle.dedup(b"")[0] == b""
    out, st = oracle.dedup(b"\n\n\n\n\nx\n\n\n")
    assert out == b"\n\n\n\n" and st.duplicates == 1 and st.total_reads == 2
    out, st = oracle.dedup(b"@a\n1\n+\n1\n@a \n2\n+\n2\n")
    assert st.duplicates == 0
    # lines after a dropped header are dropped until the next header (write_ln), also in a truncated tail
    out, st = oracle.dedup(b"@a\n1\n+\n1\n@a\n2\n+\n")
    assert out == b"@a\n1\n+\n1\n" and st.duplicates == 1 and st.total_reads == 1
