/*
 * TEST INFRASTRUCTURE — NOT PART OF THE PRODUCT (see fqcount_oracle.h).
 *
 * CPU restatement of `sc fq-count` (reference: src/fq_count.nim:14-53). Third-party semantics
 * that are not in the reference tree are restated from their published behaviour:
 *   - Nim 1.0.6 (pinned by the reference CI, .github/workflows/build.yml:45) lib/system/io.nim
 *     readLine(File): a line ends at '\n'; a '\r' directly before that '\n' is dropped; a final
 *     line without '\n' is returned; no phantom empty line after a final '\n'.
 *   - Nim 1.0.6 lib/pure/strutils.nim count(s, sub: string): repeated find() from the last hit;
 *     find of a 1-char needle is memchr.
 *   - Nim 1.0.6 `$`(float): C "%.16g", ".0" appended when no '.'/letter is present, NaN -> "nan".
 *   - nimble zip >= 0.2.1 gzipfiles.newGZFileStream == zlib gzopen/gzread (same calls as the
 *     reference's own gzip_stream.nim:13-23).
 */
#include "fqcount_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

/* strutils.count(line, "<c>") : find -> memchr per hit, restart after the hit. */
static uint64_t count_char(const uint8_t* s, size_t n, int c) {
  uint64_t k = 0;
  const uint8_t* p = s;
  const uint8_t* e = s + n;
  while (p < e) {
    const uint8_t* q = (const uint8_t*)memchr(p, c, (size_t)(e - p));
    if (!q) break;
    ++k;
    p = q + 1;
  }
  return k;
}

/* src/fq_count.nim:38-45, with the line iterator of Nim 1.0.6 io.readLine restated inline. */
void oracle_count_lines(const uint8_t* buf, size_t n, oracle_counts* out) {
  memset(out, 0, sizeof(*out));
  out->input_bytes = n;
  size_t cap = 256;
  uint8_t* line = (uint8_t*)malloc(cap);   /* the reference copies each line into a string */
  uint64_t i = 0;                            /* fq_count.nim:23  i = 0 */
  size_t pos = 0;
  while (pos < n) {                          /* for line in lines(stream)   :38 */
    const uint8_t* nlp = (const uint8_t*)memchr(buf + pos, '\n', n - pos);
    size_t end = nlp ? (size_t)(nlp - buf) : n;   /* exclusive end of line text */
    size_t len = end - pos;
    if (nlp) {
      out->newlines++;
      if (len > 0 && buf[end - 1] == '\r') len--;   /* "\r\n" -> strip the '\r' too */
    }
    if (len + 1 > cap) { cap = (len + 1) * 2; line = (uint8_t*)realloc(line, cap); }
    memcpy(line, buf + pos, len);
    i++;                                     /* i.inc()                     :39 */
    if ((i % 4) == 1) out->reads++;          /* n_reads.inc()               :40-41 */
    if ((i % 4) == 2) {                      /*                              :42 */
      out->gc_bases += count_char(line, len, 'G') + count_char(line, len, 'C');  /* :43 */
      out->n_bases += count_char(line, len, 'N');                                 /* :44 */
      out->bases += len;                                                          /* :45 */
    }
    pos = nlp ? end + 1 : n;
  }
  out->lines = i;
  free(line);
}

/* Independent byte-serial restatement; also defines the K3 / K4 additions. */
void oracle_count_bytes(const uint8_t* buf, size_t n, oracle_counts* out) {
  memset(out, 0, sizeof(*out));
  out->input_bytes = n;
  uint64_t line_no = 0;     /* 1-based number of the line the current byte belongs to */
  int at_line_start = 1;
  for (size_t k = 0; k < n; ++k) {
    uint8_t b = buf[k];
    if (at_line_start) {
      line_no++;
      if ((line_no & 3) == 1) { out->reads++; if (b != '@') out->bad_at++; }
      if ((line_no & 3) == 3) { if (b != '+') out->bad_plus++; }
      at_line_start = 0;
    }
    if (b == '\n') {
      out->newlines++;
      at_line_start = 1;
      continue;
    }
    int eol_cr = (b == '\r' && k + 1 < n && buf[k + 1] == '\n');
    if (eol_cr) continue;
    if ((line_no & 3) == 2) {
      out->bases++;
      if (b == 'G' || b == 'C') out->gc_bases++;
      if (b == 'N') out->n_bases++;
    } else if ((line_no & 3) == 0) {
      out->qual_hist[b]++;
    }
  }
  out->lines = line_no;
}

void oracle_partial(const uint8_t* buf, size_t n, int prev_byte, uint64_t o[27], uint64_t* hist) {
  memset(o, 0, 27 * sizeof(uint64_t));
  if (hist) memset(hist, 0, 4 * 256 * sizeof(uint64_t));
  uint64_t* nl = &o[0];
  uint64_t* gc = &o[1];
  uint64_t* nn = &o[5];
  uint64_t* len = &o[9];
  uint64_t* starts = &o[13];
  uint64_t* fat = &o[17];
  uint64_t* fplus = &o[21];
  unsigned r = 0;
  int prev = prev_byte;
  for (size_t k = 0; k < n; ++k) {
    uint8_t b = buf[k];
    if (prev == -1 || prev == '\n') {
      starts[r]++;
      if (b == '@') fat[r]++;
      if (b == '+') fplus[r]++;
    }
    if (b == '\n') {
      if (prev == '\r') {            /* that '\r' was counted as a line byte: take it back */
        len[r]--;                    /* may wrap when the '\r' lies in the previous shard: u64 modular, exact after combine */
        if (hist) hist[r * 256 + '\r']--;
      }
      (*nl)++;
      r = (r + 1) & 3;
    } else {
      len[r]++;
      if (b == 'G' || b == 'C') gc[r]++;
      if (b == 'N') nn[r]++;
      if (hist) hist[r * 256 + b]++;
    }
    prev = b;
  }
  o[25] = n;
  o[26] = n ? buf[n - 1] : 0;
}

int oracle_count_file(const char* path, oracle_counts* out) {
  size_t plen = strlen(path);
  int is_gz = plen >= 3 && memcmp(path + plen - 3, ".gz", 3) == 0;   /* fastq[^3 .. ^1] == ".gz" */
  size_t cap = 1 << 20, n = 0;
  uint8_t* buf = (uint8_t*)malloc(cap);
  if (is_gz) {
    gzFile f = gzopen(path, "r");
    if (!f) { free(buf); return -1; }
    for (;;) {
      if (cap - n < (1 << 16)) { cap *= 2; buf = (uint8_t*)realloc(buf, cap); }
      int got = gzread(f, buf + n, (unsigned)(cap - n > (1u << 30) ? (1u << 30) : cap - n));
      if (got <= 0) break;
      n += (size_t)got;
    }
    gzclose(f);
  } else {
    FILE* f = fopen(path, "rb");
    if (!f) { free(buf); return -1; }
    for (;;) {
      if (cap - n < (1 << 16)) { cap *= 2; buf = (uint8_t*)realloc(buf, cap); }
      size_t got = fread(buf + n, 1, cap - n, f);
      if (got == 0) break;
      n += got;
    }
    fclose(f);
  }
  oracle_counts a, b;
  oracle_count_lines(buf, n, &a);
  oracle_count_bytes(buf, n, &b);
  free(buf);
  /* the two restatements must agree on every reference counter */
  if (a.reads != b.reads || a.gc_bases != b.gc_bases || a.n_bases != b.n_bases ||
      a.bases != b.bases || a.lines != b.lines || a.newlines != b.newlines) {
    fprintf(stderr, "oracle: internal restatements disagree on %s\n", path);
    abort();
  }
  *out = b;
  return 0;
}

int oracle_format_tsv(const oracle_counts* c, char* buf, size_t cap) {
  /* $(gc_cnt.float / (total_len - n_cnt).float)   src/fq_count.nim:48 */
  double denom = (double)(int64_t)(c->bases - c->n_bases);
  double v = (double)(int64_t)c->gc_bases / denom;
  char f[80];
  int m = snprintf(f, sizeof f, "%.16g", v);
  int has_dot = 0;
  for (int k = 0; k < m; ++k) {
    if (f[k] == ',') { f[k] = '.'; has_dot = 1; }
    else if ((f[k] >= 'a' && f[k] <= 'z') || (f[k] >= 'A' && f[k] <= 'Z') || f[k] == '.') has_dot = 1;
  }
  if (!has_dot) { f[m] = '.'; f[m + 1] = '0'; f[m + 2] = 0; m += 2; }
  if (m > 0 && (f[m - 1] == 'n' || f[m - 1] == 'N')) { strcpy(f, "nan"); }
  else if (m > 0 && (f[m - 1] == 'f' || f[m - 1] == 'F')) { strcpy(f, f[0] == '-' ? "-inf" : "inf"); }
  return snprintf(buf, cap, "%llu\t%s\t%llu\t%llu\t%llu", (unsigned long long)c->reads, f,
                  (unsigned long long)c->gc_bases, (unsigned long long)c->n_bases,
                  (unsigned long long)c->bases);
}
