/*
 * TEST INFRASTRUCTURE — NOT PART OF THE PRODUCT.
 * CPU restatement of the reference's `sc fq-count` algorithm (danielecook/seq-collection).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The shipped path (libsc_fqcount_hip.so) never links, loads or calls anything in oracle/.
 *
 * Parity pin: checked against all 15 rows of the reference's docs/fq-count.md:27-43 over the
 * reference's tests/fastq inputs (committed as tests/golden/, expected values in
 * tests/golden/golden.tsv). The reference itself (Nim 1.0.6 + nimble zip/hts/argparse) cannot be
 * compiled in this image (no nim toolchain), so there is no oracle/_ref build; behaviours no
 * reference fixture exercises (N bases, CRLF, blank / truncated records, empty input, multi-member
 * gzip) follow the pinned Nim 1.0.6 stdlib semantics restated below and are listed as
 * "parity unpinned" in DESIGN.md.
 */
#ifndef FQCOUNT_ORACLE_H
#define FQCOUNT_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_counts {
  uint64_t reads, gc_bases, n_bases, bases;   /* src/fq_count.nim:24-28 */
  uint64_t lines, newlines, input_bytes;
  uint64_t bad_at, bad_plus;                   /* K4 additions (not in reference) */
  uint64_t qual_hist[256];                     /* K3 addition (not in reference) */
} oracle_counts;

/* Reference-shaped: split into lines exactly as Nim 1.0.6 io.readLine(File) does, then the
 * `i mod 4` classifier with three separate count passes (src/fq_count.nim:38-45). Also the timed
 * CPU baseline ("port") of bench.py. */
void oracle_count_lines(const uint8_t* buf, size_t n, oracle_counts* out);

/* Independent second restatement: one byte-serial state machine, no line buffer. Used to
 * cross-check oracle_count_lines and to derive K3/K4 expectations. Fills every field. */
void oracle_count_bytes(const uint8_t* buf, size_t n, oracle_counts* out);

/* File entry: ".gz" suffix (last three bytes, case-sensitive, src/fq_count.nim:31) -> zlib gzread
 * (gzip_stream.nim:16-17 semantics), else plain read. Returns 0, or -1 when the file cannot be
 * opened (reference: quit_error "Unable to open file", exit 2, src/fq_count.nim:35-36). */
int oracle_count_file(const char* path, oracle_counts* out);

/* The shard partial of SURVEY.md §7 computed serially (for tests of arbitrary shard boundaries).
 * Layout identical to scfq_partial's first 27 words: nl, gc[4], n[4], len[4], starts[4],
 * first_at[4], first_plus[4], bytes, last_byte. hist may be NULL (else uint64_t[4][256]). */
void oracle_partial(const uint8_t* buf, size_t n, int prev_byte, uint64_t out27[27], uint64_t* hist);

/* src/fq_count.nim:47-51 with Nim 1.0.6 `$float` ("%.16g" + ".0" rule, "nan"). Returns length. */
int oracle_format_tsv(const oracle_counts* c, char* buf, size_t cap);

/* ---- `sc fq-dedup` (reference: src/fq_dedup.nim:14-84), restated in fqdedup_oracle.c ------------------------------ */
typedef struct oracle_dedup_stats {
  uint64_t total_reads;      /* n_reads = lines div 4                 src/fq_dedup.nim:49 */
  uint64_t duplicates;       /* n_dups: records dropped               :65 */
  uint64_t false_positive;   /* Bloom diagnostics (:76-80): restated as 0, see fqdedup_oracle.c */
  uint64_t records_out, bytes_out;
} oracle_dedup_stats;
/* Writes the de-duplicated FASTQ (what the reference echoes to stdout) into out; returns its length, -2 when out_cap is
 * too small, -1 on allocation failure. */
int64_t oracle_dedup(const uint8_t* buf, size_t n, uint8_t* out, size_t out_cap, oracle_dedup_stats* st);

#ifdef __cplusplus
}
#endif
#endif
