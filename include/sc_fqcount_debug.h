/*
 * sc_fqcount_debug.h — diagnostic entry points of libsc_fqcount_hip.so. NOT part of the drop-in boundary
 * (include/sc_fqcount.h): nothing here is needed by a reference-side binding of `sc fq-count`
 * (src/fq_count.nim:14-53); the parity tests, bench.py's stream-ceiling probe and the measurement scripts use them.
 * No counting entry point calls any of these.
 */
#ifndef SC_FQCOUNT_DEBUG_H
#define SC_FQCOUNT_DEBUG_H

#include "sc_fqcount.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Diagnostic: ranges of this thread's last histogram session that the speculative K3 form served / that were (re)done
 * by the exact kernel (no guess, or a guess that did not verify). */
int  scfq_debug_hist_stats(uint64_t* fast_ranges, uint64_t* redone_ranges);
/* Diagnostic only (used by the parity tests as a second, independent device implementation):
 * byte-serial HIP kernel, one thread per 256 bytes. Never called by the counting entry points. */
int  scfq_debug_partial_simple(const void* device_ptr, uint64_t n, int prev_byte, scfq_partial* out);
/* Diagnostic only: the byte stream scfq_count_file() would scan for `path` (plain pread, BGZF block-parallel
 * inflate, or serial gzread), produced on the host without any device. Returns bytes written or a negative code. */
int64_t scfq_debug_read_file(const char* path, void* dst, uint64_t cap, uint64_t chunk_bytes);
/* host only: the same bytes with the first member read in two halves — the serial decoder up to the first block boundary at or after
 * `after_bytes` of output, then the decoder that takes a stream over in the middle of a member (exact bit, window, CRC-32 and length
 * so far): the hand-over of the device gzip path when a batch has no room */
int64_t scfq_debug_gz_resume(const char* path, uint64_t after_bytes, void* dst, uint64_t cap, uint64_t chunk_bytes);
/* Diagnostic only: inflate a whole BGZF image (host memory) with the device-side inflate kernel, result to host memory.
 * Returns the inflated size, SCFQ_EARG when the image is not pure BGZF or does not fit into cap, SCFQ_EGZ for a corrupt
 * member (deflate data, ISIZE or CRC-32). */
int64_t scfq_debug_bgzf_inflate(const void* image, uint64_t n, void* out, uint64_t cap);
/* Diagnostic only: milliseconds (best of `reps`) the scan kernel's LOAD STRUCTURE alone (same ranges, same non-temporal
 * LDS-DMA ring, no classification / accounting) needs for the whole tiles of a 4 KiB-aligned device buffer: the
 * practical read-stream ceiling on this device, reported next to the roofline by bench.py. Negative on error. */
double scfq_debug_stream_ms(const void* device_ptr, uint64_t n, int reps);

/* Host only — the two rules scfq_count_file_sharded cuts an ordinary (non-BGZF) .gz file of several members by, for the CPU tests.
 * scfq_debug_gz_member_boundary: the first position at or after `from` where a gzip member demonstrably starts (magic, no reserved
 * flag bit, a header that parses, deflate data that inflates cleanly for its first 64 KiB); the file's size when there is none, a
 * negative code on error.  *first_byte: the first byte the members from there inflate to (-1: none).
 * scfq_debug_gz_shard_fix: a shard's partial (+ histogram) that was scanned as if it began the input (prev_byte -1), put right for the
 * byte that really lies in front of it (true_prev 0..255; -1: it does begin the input) given the shard's first byte; flags: the
 * SCFQ_* flags of the scan (the line-start words are only touched with SCFQ_STRUCT_CHECK). */
int64_t scfq_debug_gz_member_boundary(const char* path, uint64_t from, int* first_byte);
int scfq_debug_gz_shard_fix(scfq_partial* p, uint64_t* hist, int true_prev, int first_byte, uint32_t flags);
/* Diagnostic only: where this process's time went so far.  The library marks its stages (first device call returned, context up,
 * buffers allocated, first copy queued, first kernel queued, session folded ...) with the milliseconds since it was loaded;
 * scfq_debug_stages writes them as one JSON array of [name, ms] pairs and returns its length (the length needed, with nothing
 * written, when cap is too small); scfq_debug_stage_mark lets the host add marks of its own (`sc`: main entered, rows out).
 * `sc fq-count --stats` prints the array; bench.py's cold-process legs carry it. */
int64_t scfq_debug_stages(char* buf, uint64_t cap);
void scfq_debug_stage_mark(const char* what);
/* The scan kernel instance and range geometry of the calling thread's last scan launch, e.g.
 * "fq_scan_tiles<false, 0, 2, true, false> tiles_per_range=100" (same contract as scfq_debug_stages: returns the length, writes when it fits). */
int64_t scfq_debug_last_scan_kernel(char* buf, uint64_t cap);

#ifdef __cplusplus
}
#endif
#endif /* SC_FQCOUNT_DEBUG_H */
