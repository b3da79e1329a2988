/*
 * count_shards.c — plain-C host over the C ABI (include/sc_fqcount.h): counts a FASTQ image that is already in
 * host memory as K byte-range shards cut at arbitrary offsets, folds the shard partials in order and prints the
 * reference's TSV row (src/fq_count.nim:47-53).  This is the shape of a multi-GPU host: one shard per device or
 * per rank, one ordered combine.
 *
 *   gcc -std=c99 -Iinclude examples/count_shards.c -Lseq-collection_amd -lsc_fqcount_hip -Wl,-rpath,$PWD/seq-collection_amd -o count_shards
 *   ./count_shards tests/golden/sra.fq 3
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sc_fqcount.h"

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: %s file.fq [shards]\n", argv[0]); return 2; }
  const int shards = argc > 2 ? atoi(argv[2]) : 4;
  FILE* f = fopen(argv[1], "rb");
  if (!f) { fprintf(stderr, "Unable to open file: %s\n", argv[1]); return 2; }
  fseek(f, 0, SEEK_END);
  const long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  unsigned char* buf = (unsigned char*)malloc(n > 0 ? (size_t)n : 1);
  if (fread(buf, 1, (size_t)n, f) != (size_t)n) { fprintf(stderr, "read error\n"); return 1; }
  fclose(f);

  scfq_partial acc;
  scfq_partial_identity(&acc, NULL);
  for (int k = 0; k < shards; ++k) {
    const uint64_t lo = (uint64_t)n * (uint64_t)k / (uint64_t)shards, hi = (uint64_t)n * (uint64_t)(k + 1) / (uint64_t)shards;
    scfq_partial part;
    /* the byte before the shard is its 1-byte halo ("\r\n" line ends); -1 at the start of the file */
    const int rc = scfq_partial_buffer(buf + lo, hi - lo, /*is_device=*/0, lo ? buf[lo - 1] : -1, NULL, &part, NULL);
    if (rc != SCFQ_OK) { fprintf(stderr, "%s: %s\n", scfq_strerror(rc), scfq_last_error_detail()); return 1; }
    scfq_partial_combine(&acc, &part, NULL, NULL);   /* ordered: shard k after shards 0..k-1 */
  }
  scfq_counts c;
  memset(&c, 0, sizeof c);
  c.struct_size = sizeof c;
  scfq_partial_finalize(&acc, NULL, &c);
  char row[256];
  scfq_format_tsv(&c, row, sizeof row);
  printf("%s\n", row);
  free(buf);
  scfq_shutdown();
  return 0;
}
