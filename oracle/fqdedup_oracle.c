/*
 * TEST INFRASTRUCTURE — NOT PART OF THE PRODUCT (see fqcount_oracle.h).
 *
 * CPU restatement of `sc fq-dedup` (reference: src/fq_dedup.nim:14-84): de-duplicate a FASTQ by read ID, keep the
 * first record of every ID, echo everything else unchanged.
 *
 * What the reference does (and what is restated here):
 *   pass 1 (:42-47)  for every line with 0-based index i, i mod 4 == 0 (header lines): if the Bloom filter already
 *                    holds the line, remember it in `check`; insert it.  n_reads = i div 4 (:49).
 *   pass 2 (:54-73)  header lines not in `check` are echoed; header lines in `check` are echoed the first time they are
 *                    met and dropped (with the non-header lines that follow, write_ln = false) from the second time
 *                    on, n_dups.inc.  Non-header lines follow the latest header's write_ln.
 * The Bloom filter (nimble `bloom`, not in the reference tree; capacity 1e8, error rate 1e-4, :29) is only a
 * pre-filter: a false positive puts a unique ID into `check`, where it is echoed on its first (only) occurrence
 * anyway.  So stdout and `duplicates` are exactly "drop every record whose header line was seen before", which is
 * what is restated.  `false-positive` (:76-80) counts those Bloom false positives; it depends on the Bloom
 * implementation's hash functions, is documented as "for diagnostics only" (docs/fq-dedup.md:24) and is 0 for every
 * input this size: restated as 0 ("parity unpinned").
 * `echo record` writes the line (EOL stripped by Nim 1.0.6 readLine: '\n', and a '\r' directly before it) plus '\n':
 * CRLF input comes out as LF, a final line without '\n' gains one.
 *
 * Parity pin: scripts/functional-tests.sh:86-92 (dup.fq and dup.fq.gz -> 4 lines containing '@').
 */
#include "fqcount_oracle.h"

#include <stdlib.h>
#include <string.h>

typedef struct { const uint8_t* p; size_t n; } str_t;

static uint64_t fnv(const uint8_t* p, size_t n) {
  uint64_t h = 1469598103934665603ull;
  for (size_t k = 0; k < n; ++k) { h ^= p[k]; h *= 1099511628211ull; }
  return h;
}

/* returns 1 when the string was already present, else inserts it and returns 0 (open addressing, exact compare) */
static int set_test_and_insert(str_t* tab, size_t cap, const uint8_t* p, size_t n) {
  size_t k = (size_t)(fnv(p, n) & (cap - 1));
  for (;;) {
    if (!tab[k].p) { tab[k].p = p; tab[k].n = n; return 0; }
    if (tab[k].n == n && memcmp(tab[k].p, p, n) == 0) return 1;
    k = (k + 1) & (cap - 1);
  }
}

int64_t oracle_dedup(const uint8_t* buf, size_t n, uint8_t* out, size_t out_cap, oracle_dedup_stats* st) {
  memset(st, 0, sizeof(*st));
  /* number of lines -> table size */
  uint64_t lines = 0;
  for (size_t k = 0; k < n; ++k) lines += (buf[k] == '\n');
  if (n && buf[n - 1] != '\n') lines++;
  size_t cap = 16;
  while (cap < (lines / 4 + 1) * 2 + 2) cap <<= 1;
  str_t* tab = (str_t*)calloc(cap, sizeof(str_t));
  if (!tab) return -1;
  static const uint8_t empty = 0;
  uint64_t i = 0;
  size_t pos = 0, w = 0;
  int write_ln = 1;
  int64_t rc = 0;
  while (pos < n) {
    const uint8_t* nlp = (const uint8_t*)memchr(buf + pos, '\n', n - pos);
    size_t end = nlp ? (size_t)(nlp - buf) : n;
    size_t len = end - pos;
    if (nlp && len > 0 && buf[end - 1] == '\r') len--;
    if (i % 4 == 0) {                                    /* header line        fq_dedup.nim:57 */
      if (set_test_and_insert(tab, cap, len ? buf + pos : &empty, len)) { write_ln = 0; st->duplicates++; }   /* :62-66 */
      else write_ln = 1;
    }
    if (write_ln) {                                      /* echo record        :59,67,70-71 */
      if (w + len + 1 > out_cap) { rc = -2; break; }
      memcpy(out + w, buf + pos, len);
      out[w + len] = '\n';
      w += len + 1;
      if (i % 4 == 0) st->records_out++;
    }
    i++;
    pos = nlp ? end + 1 : n;
  }
  free(tab);
  st->total_reads = i / 4;                               /* n_reads = i div 4  :49 */
  st->false_positive = 0;
  st->bytes_out = w;
  return rc < 0 ? rc : (int64_t)w;
}
