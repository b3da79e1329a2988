/*
 * sc_fqcount.h — C ABI of libsc_fqcount_hip.so, the MI355X (gfx950) drop-in for the
 * `sc fq-count` hot path of danielecook/seq-collection.
 *
 * What this boundary replaces in the reference (all paths relative to the reference tree):
 *   src/fq_count.nim:30-45   open stream (plain / ".gz"), per-line loop, i mod 4 classifier,
 *                            count("G")+count("C"), count("N"), line.len accumulation
 *   src/fq_count.nim:47-51   the five output fields and their `$` formatting
 *   gzip_stream.nim:13-23    readData == zlib gzread (the host inflate semantics kept here)
 * The reference has no FFI of its own for this path; the seam is the Nim proc
 *   fq_count*(fastq: string, basename: bool, absolute: bool)          (src/fq_count.nim:14)
 * A Nim host keeps that proc and calls scfq_count_file() + scfq_format_tsv() in place of
 * lines :30-51 (binding shown in INTEGRATION.md and seq-collection_amd/nim/fq_count.nim).
 *
 * Plain C: pointers and sizes only, no C++ types, no exceptions cross this boundary, the
 * library never calls exit() and never writes to stdout.
 */
#ifndef SC_FQCOUNT_H
#define SC_FQCOUNT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCFQ_ABI_VERSION 1u

/* ---- return codes (0 = success; negative = error) -------------------------------------- */
#define SCFQ_OK       0
#define SCFQ_EOPEN   (-1)  /* cannot open input: host maps to quit_error("Unable to open file: "&path, 2), src/fq_count.nim:35-36 */
#define SCFQ_EGZ     (-2)  /* zlib reported a corrupt / truncated gzip stream */
#define SCFQ_EHIP    (-3)  /* a HIP runtime call or kernel launch failed (no GPU, OOM, ...) */
#define SCFQ_ERCCL   (-4)  /* collective exchange failed */
#define SCFQ_EARG    (-5)  /* bad argument (NULL pointer, struct_size mismatch, bad flag) */
#define SCFQ_EIO     (-6)  /* read error after a successful open */
#define SCFQ_ENOMEM  (-7)  /* host allocation failed */
#define SCFQ_EPIPE   (-9)  /* the reader of an output descriptor went away (write -> EPIPE): `sc fq-dedup | head` ends quietly with
                              status 0 as the reference does for `errno: 32 Broken pipe` (sc.nim:304); every other write error is SCFQ_EIO */
#define SCFQ_ESPEC   (-8)  /* the speculative quality histogram of a shard backed a class that is not the quality line
                              (malformed input cut into shards): count that shard again with SCFQ_HIST_EXACT.
                              scfq_count_file / scfq_count_buffer do this by themselves and never return it. */

/* ---- option flags ------------------------------------------------------------------------ */
#define SCFQ_QUAL_HIST     0x1u  /* also build the 256-bin histogram of quality-line bytes (K3; not in reference) */
#define SCFQ_STRUCT_CHECK  0x2u  /* also count header lines not starting '@' / separator lines not starting '+' (K4) */
#define SCFQ_TIMING        0x4u  /* bracket the scan kernel with HIP events; read back with scfq_last_timing() */
#define SCFQ_PREV_IN_MEMORY 0x8u /* scfq_partial_buffer: the byte before `ptr` is addressable and is the look-behind halo */
#define SCFQ_WAIT_STREAM   0x20u /* order this call after the work already enqueued on scfq_opts.wait_stream (see there) */
#define SCFQ_HIST_EXACT    0x10u /* with SCFQ_QUAL_HIST: build all four class histograms exactly (slower kernel) instead of the
                                    verified-speculative form that only completes the quality class (scfq_partial.hist_class) */

/* ---- results: the counters of src/fq_count.nim:22-28 plus derivation inputs ------------- */
typedef struct scfq_counts {
  uint64_t struct_size;   /* caller sets to sizeof(scfq_counts) before the call */
  uint64_t abi_version;   /* library writes SCFQ_ABI_VERSION */
  uint64_t reads;         /* n_reads   = #{lines i : i mod 4 == 1}      fq_count.nim:40-41 */
  uint64_t gc_bases;      /* gc_cnt    = count("G")+count("C") on i mod 4 == 2   :43 */
  uint64_t n_bases;       /* n_cnt     = count("N")                              :44 */
  uint64_t bases;         /* total_len = sum line.len (EOL stripped)             :45 */
  uint64_t lines;         /* number of lines the reference's `lines(stream)` yields */
  uint64_t newlines;      /* number of '\n' bytes */
  uint64_t input_bytes;   /* bytes scanned (inflated bytes for .gz) */
  uint64_t bad_at;        /* K4: header lines whose first byte is not '@'   (0 unless SCFQ_STRUCT_CHECK) */
  uint64_t bad_plus;      /* K4: separator lines whose first byte is not '+' (0 unless SCFQ_STRUCT_CHECK) */
  uint64_t qual_hist[256];/* K3: byte histogram of quality lines (i mod 4 == 0), EOL bytes excluded; zeros unless SCFQ_QUAL_HIST */
} scfq_counts;

typedef struct scfq_opts {
  uint64_t struct_size;     /* sizeof(scfq_opts) */
  int32_t  n_devices;       /* 0 = use the current / default device only; >0 = shard byte ranges across device_ids[0..n) */
  const int32_t* device_ids;/* may be NULL when n_devices == 0 */
  uint32_t flags;           /* SCFQ_* option flags */
  uint32_t reserved;
  uint64_t chunk_bytes;     /* host->device staging chunk for file / host-buffer ingest; 0 = default (64 MiB) */
  /* ---- fields below exist when struct_size >= 48 (SCFQ_OPTS_V1_SIZE = 40 is still accepted) ---- */
  void*    wait_stream;     /* hipStream_t of the CALLER, read when SCFQ_WAIT_STREAM is set in flags (NULL then names the legacy
                               default stream). Device-pointer arguments (inputs and outputs) are used on library-private
                               non-blocking streams, which do not order against the stream that produced the input or last
                               used the output buffer. With SCFQ_WAIT_STREAM the library records an event on this stream at
                               entry and makes its own streams wait for it, so work already enqueued there completes first.
                               Without it the caller has synchronised (hipStreamSynchronize / hipDeviceSynchronize) before
                               the call. Every entry point synchronises its own streams before it returns, so results are
                               visible to any stream afterwards. */
} scfq_opts;
#define SCFQ_OPTS_V1_SIZE 40u

/*
 * The shard partial: what a contiguous byte range contributes when its line
 * phase at the first byte is unknown. Index r = (number of '\n' seen so far in THIS range) mod 4.
 * Flat u64 so it can be exchanged with one collective (RCCL uint64 / torch int64).
 *   combine  (A (+) B).nl = A.nl + B.nl ; (A (+) B).x[r] = A.x[r] + B.x[(r - A.nl) mod 4]
 * Associative, NOT commutative. Fold in file order from the identity.
 */
#define SCFQ_PARTIAL_WORDS 32
typedef struct scfq_partial {
  uint64_t nl;            /* '\n' count in range */
  uint64_t gc[4];         /* 'G'|'C' bytes per relative class */
  uint64_t n[4];          /* 'N' bytes per relative class */
  uint64_t len[4];        /* bytes that are not '\n' and not a '\r' directly before '\n' */
  uint64_t starts[4];     /* K4: line-start events (first byte of a line lies in this range) */
  uint64_t first_at[4];   /* K4: ... whose first byte is '@' */
  uint64_t first_plus[4]; /* K4: ... whose first byte is '+' */
  uint64_t bytes;         /* bytes covered */
  uint64_t last_byte;     /* value of the last byte covered (undefined when bytes == 0) */
  uint64_t hist_class;    /* K3 side array: 0 = all four class histograms are complete; k+1 (k = 0..3) = only class k is
                             (the speculative form histograms just the lines it verified to be quality lines);
                             5 = none (shards that disagree were combined: finalize returns SCFQ_ESPEC) */
  uint64_t reserved[4];   /* [0]: status word, OR-ed by the combine (scfq_count_file_sharded: a rank whose shard failed); the rest 0
                             (scfq_count_file_sharded carries a gzip shard's first byte, raw CRC-32 and length in them between its ranks) */
} scfq_partial;

#define SCFQ_HIST_WORDS (4 * 256)  /* optional K3 side array: uint64_t hist[4][256], class-major */

typedef struct scfq_timing {
  uint64_t struct_size;
  double scan_kernel_ms;  /* device time of the last scan launch(es) on the library stream (HIP events) */
  double fold_kernel_ms;  /* device time of the partial fold */
  uint64_t scan_bytes;    /* bytes those scan launches covered */
  uint64_t scan_launches;
  double host_fill_ms;    /* host time spent producing chunks (pread / zlib inflate) during the last ingest */
  double ingest_wall_ms;  /* wall time of the last chunked ingest (fill + copy + scan, overlapped) */
  uint64_t h2d_bytes;     /* bytes moved host -> HBM by the last ingest */
  double h2d_ms;          /* device time of those copies on the copy stream (HIP events) */
} scfq_timing;

/* ---- whole-input entry points (what a host binds) ---------------------------------------- */

/* Opens `path`; last three bytes ".gz" (case-sensitive, src/fq_count.nim:31) selects host zlib
 * inflate (gzread semantics incl. concatenated members, gzip_stream.nim:16-17) overlapped with
 * device scans via pinned buffers on a copy stream; otherwise plain pread. */
int scfq_count_file(const char* path, const scfq_opts* opts, scfq_counts* out);

/* Counts an in-memory FASTQ image. is_device != 0: `ptr` is a device pointer on the current
 * device (bytes already resident in HBM: the roofline configuration). */
int scfq_count_buffer(const void* ptr, uint64_t n, int is_device, const scfq_opts* opts, scfq_counts* out);

/* ---- shard-level entry points (multi-GPU / streaming hosts, tests of shard boundaries) --- */

/* Partial of one contiguous byte range. prev_byte: the byte immediately before the range in the
 * file (0..255), or -1 when the range starts the file. Needed because a '\r' directly before a
 * '\n' is not part of the line (look-behind halo of 1 byte). With SCFQ_PREV_IN_MEMORY set in
 * opts->flags prev_byte is ignored and ptr[-1] is read instead.
 * hist: NULL, or uint64_t[SCFQ_HIST_WORDS] (written only when SCFQ_QUAL_HIST is set). */
int scfq_partial_buffer(const void* ptr, uint64_t n, int is_device, int prev_byte,
                        const scfq_opts* opts, scfq_partial* out, uint64_t* hist);

void scfq_partial_identity(scfq_partial* p, uint64_t* hist);
/* acc = acc (+) b   (hist arrays may be NULL) */
int  scfq_partial_combine(scfq_partial* acc, const scfq_partial* b, uint64_t* hist_acc, const uint64_t* hist_b);
/* Interpret a partial folded from the start of the file: select the sequence class, derive
 * lines and reads (ceil(lines/4), src/fq_count.nim:39-41). */
int  scfq_partial_finalize(const scfq_partial* p, const uint64_t* hist, scfq_counts* out);

/* ---- C1: the cross-rank exchange of shard partials -----------------------------
 * The reference is one process with no collective (one call per file, sc.nim:114-116); the exchange exists because the
 * input is byte-range sharded across the GPUs of a node. Every rank contributes the partial of its shard; all ranks
 * receive the same rank-ordered (+) fold (an all-gather of the 32-word partials over RCCL / xGMI and the ordered combine —
 * NOT a sum-allreduce of counters: (+) is not commutative). librccl is dlopen()ed on first use.
 * A communicator runs its RCCL / HIP calls on its own worker thread; every wait has a deadline (timeout_ms, <= 0 means
 * 300 s) and a missing or stuck rank comes back as SCFQ_ERCCL, never as a hang. scfq_comm_error_detail() has the text.
 * One communicator per (process, device); calls on one communicator come from one host thread at a time. */
typedef struct scfq_comm scfq_comm;
#define SCFQ_COMM_ID_BYTES 128   /* sizeof(ncclUniqueId) */
#define SCFQ_COMM_RCCL 0         /* ncclAllGather of ncclUint64 on a private stream of the rank's device */
#define SCFQ_COMM_TCP  1         /* host sockets through rank 0: explicit opt-in for hosts without a common RCCL fabric and for
                                    CPU-only tests of the multi-process path; needs no device; never chosen by the library itself */

/* STDOUT: RCCL prints a version banner on stdout when a process's first communicator comes up. While any scfq_comm_init_* call
 * is in flight (communicator creation plus one warm-up collective) descriptor 1 of the process points at descriptor 2, and what
 * OTHER threads of the host write to stdout in that window lands on stderr. A host that prints from several threads creates
 * its communicators before its first row: scfq_prepare() below does that for the in-process multi-device path.
 * One process per GPU, the host distributes the id itself (rank 0 creates it, every rank passes the same bytes): */
int scfq_comm_unique_id(void* id, uint64_t cap /* >= SCFQ_COMM_ID_BYTES */);
int scfq_comm_init_rank(const void* id, int world, int rank, int device, int timeout_ms, scfq_comm** out);
/* One process per GPU, the library distributes the id: rank 0 listens on host:port (IPv4 name or address, NULL = 127.0.0.1),
 * the other ranks connect (retrying until the deadline). transport = SCFQ_COMM_RCCL | SCFQ_COMM_TCP. */
int scfq_comm_init_rendezvous(const char* host, int port, int world, int rank, int device, int transport, int timeout_ms,
                              scfq_comm** out);
/* One process, n distinct devices (ncclCommInitAll): out[k] is the communicator of device_ids[k], rank k. */
int scfq_comm_init_all(int n, const int32_t* device_ids, int timeout_ms, scfq_comm** out);
int scfq_comm_world(const scfq_comm* c);
int scfq_comm_rank(const scfq_comm* c);
/* 1 once an exchange on this communicator failed or timed out: every later call fails fast with SCFQ_ERCCL (an answer that
 * arrives after its deadline is dropped, never handed to a later exchange); destroy it and create a new one. */
int scfq_comm_is_broken(const scfq_comm* c);
const char* scfq_comm_transport(const scfq_comm* c);   /* "RCCL 2.x.y" | "tcp"; thread-local static storage */
/* folded = P_0 (+) P_1 (+) ... (+) P_{world-1}, identical on every rank. hist: NULL on every rank, or uint64_t[SCFQ_HIST_WORDS]
 * on every rank (then hist_folded receives the folded class histograms). */
int scfq_comm_exchange(scfq_comm* c, const scfq_partial* mine, const uint64_t* hist, scfq_partial* folded,
                       uint64_t* hist_folded, int timeout_ms);
/* The same in two halves, so that a host can scan shard k+1 while the exchange of shard k is in flight; finishes answer
 * starts in order. */
int scfq_comm_exchange_start(scfq_comm* c, const scfq_partial* mine, const uint64_t* hist, int timeout_ms);
int scfq_comm_exchange_finish(scfq_comm* c, scfq_partial* folded, uint64_t* hist_folded, int timeout_ms);
/* all[r * words + k] = word k of rank r (words <= 1056): barriers and max-over-ranks timings of a non-Python host. */
int scfq_comm_allgather_u64(scfq_comm* c, const uint64_t* mine, uint32_t words, uint64_t* all, int timeout_ms);
int scfq_comm_destroy(scfq_comm* c);
const char* scfq_comm_error_detail(void);   /* static, thread-local */

/* Optional warm-up for a host that is about to call scfq_count_file / scfq_count_buffer with these opts: creates the device
 * context(s) (streams, pinned staging) and, for opts->n_devices > 1 with distinct devices, the in-process RCCL communicators —
 * i.e. everything that would otherwise happen inside the first counting call, including the stdout window described above.
 * `sc fq-count` calls it before its first row. Safe to call more than once; returns SCFQ_OK or what the set-up returned. */
int scfq_prepare(const scfq_opts* opts);

/* fq_count of ONE file by all ranks of a communicator: rank r scans bytes [size*r/world, size*(r+1)/world) of `path` — cut at
 * arbitrary byte offsets, one byte of look-behind — on the current device (or opts->device_ids[0]), the partials are
 * exchanged, every rank receives the counters of the whole file.
 * ".gz" input: a BGZF (bgzip) file shards where its members are — rank r takes the members that start in its byte range (the
 * first cut at or after size*r/world where eight members follow one another; the byte in front of a rank's first inflated byte
 * comes from the member before the cut, inflated on the host), inflates them on its device and scans them; the ranks agree on
 * that with one extra all-gather of a word, and fall back together when any of them saw something else in its range.
 * An ordinary gzip file of SEVERAL members (cat a.gz b.gz, pigz -i, a sequencer's writer) shards where members start: a rank's
 * cut is the first position at or after size*r/world with the magic bytes, a header that parses and deflate data that inflates
 * cleanly; it inflates and scans the members of its stretch as if they began the input, and the byte in front of its first
 * inflated byte — the last byte of the rank before it — is put right when the gathered partials are folded.  A file of ONE member
 * (gzip, pigz) or of a few big ones (every rank finds at most one member start inside its share of the file) is cut where deflate
 * BLOCKS start, a member start being a cut of its own and no stretch crossing a member's end: every rank decodes its stretch to symbols, folds what the stretch does to the
 * 32 KiB window into a map and keeps the symbols; the ranks exchange their maps and compose them in rank order, which gives each
 * the window in front of its stretch; then the symbols become bytes, are checksummed and scanned (SCFQ_SHARD_GZ_KEEP=0: nothing
 * is kept, the stretch is decoded a second time); the member's CRC-32 and ISIZE are checked against the join of the stretches'.  What proves a cut, in both schemes, is the rank before it: its members (its chain of blocks) must end exactly where
 * the next rank began.  Whenever the ranks' findings do not join up — and for damaged files — rank 0 inflates and scans the whole
 * file with the readers scfq_count_file() uses (zlib gzread's bytes and errors), the other ranks contribute the identity.
 * Knobs: SCFQ_SHARD_BGZF / SCFQ_SHARD_GZ / SCFQ_SHARD_GZ_BLOCKS = 0 switch the three schemes off.
 * Collective: every rank must call it. */
int scfq_count_file_sharded(const char* path, const scfq_opts* opts, scfq_comm* comm, scfq_counts* out);

/* ---- K5: line index (record-boundary detection) ---------------------------------------------
 * Lines as the reference's `lines(stream)` yields them (src/fq_count.nim:38, src/fq_dedup.nim:42): record i of a FASTQ
 * is lines 4i .. 4i+3. For a device-resident input writes line_off[j] = offset of the first byte of line j for
 * j = 0 .. lines-1 and the sentinel line_off[lines] = offset one past the (real or implied) final '\n', so that line j
 * is bytes [line_off[j], line_off[j+1] - 1) before "\r\n" stripping. line_off_device: device memory of `cap` entries,
 * or NULL to only count. *lines_out is always set. One pass over the input: entries are written as they are found, so
 * when cap < lines + 1 the first cap entries are valid and the index is incomplete (size from *lines_out, call again). */
int scfq_index_lines(const void* device_ptr, uint64_t n, uint64_t* line_off_device, uint64_t cap, uint64_t* lines_out);

/* ---- `sc fq-dedup` (next row of SURVEY.md §8f): src/fq_dedup.nim:14-84 ------------------------
 * De-duplicate a FASTQ by read ID: every record (lines 4i .. 4i+3) whose header line equals the header line of an
 * earlier record is dropped, everything else is echoed, each line as text + '\n' (so "\r\n" comes out as "\n").
 * Replaces the reference's two streaming passes with Bloom filter + CountTable (:29,42-73) by an exact device
 * pipeline over the HBM-resident input: line index, header hashes, radix sort, exact compare inside equal-hash runs,
 * prefix sum of the kept lengths, gather. */
typedef struct scfq_dedup_stats {
  uint64_t struct_size;      /* caller sets to sizeof(scfq_dedup_stats) */
  uint64_t total_reads;      /* n_reads = lines div 4                             src/fq_dedup.nim:49 */
  uint64_t duplicates;       /* n_dups: records dropped                           :65 */
  uint64_t false_positive;   /* the reference's Bloom-filter diagnostic (:76-80); always 0 here: no Bloom filter */
  uint64_t records_out;      /* records echoed */
  uint64_t bytes_out;        /* bytes echoed */
  uint64_t hash_collisions;  /* header pairs with equal 64-bit hash and different text (resolved by the exact compare) */
} scfq_dedup_stats;

/* Input in host (is_device = 0) or device memory; the de-duplicated FASTQ is written to out (host or device memory of
 * out_cap bytes). *out_bytes is always set to the size of the result: out = NULL only sizes (returns SCFQ_OK), a too
 * small out returns SCFQ_EARG. */
int scfq_dedup_buffer(const void* ptr, uint64_t n, int is_device, void* out, uint64_t out_cap, int out_is_device,
                      uint64_t* out_bytes, scfq_dedup_stats* stats);
/* Stages the whole input (".gz" by suffix as src/fq_dedup.nim:32: zlib / BGZF inflate on the host) into HBM,
 * de-duplicates, writes the result to out_fd (what the reference echoes to stdout); out_fd < 0: statistics only. */
int scfq_dedup_file(const char* path, const scfq_opts* opts, int out_fd, scfq_dedup_stats* stats);
const char* scfq_dedup_error_detail(void);

/* Whole (inflated) input of `path` into a device buffer the caller frees with scfq_device_free(). */
int scfq_stage_file(const char* path, const scfq_opts* opts, void** device_ptr_out, uint64_t* n_out);
int scfq_device_free(void* device_ptr);

/* ---- `sc fq-meta` (next row of SURVEY.md §8f): src/fq_meta.nim:197-278 -------------------------
 * One 16-column TSV row (scfq_meta_header() names them, fq_meta.nim:11-26) from the first sample_n records of a FASTQ:
 * machine / flowcell / run / lane parsed from the first header, sequencer guess from the instrument and flow-cell tables,
 * most frequent index, quality range and format guess. Host-side string work over a few hundred lines, as in the
 * reference. SCFQ_META_WHOLE_FILE (addition): min_qual / max_qual and the format columns use the quality-line histogram
 * of the whole file (K3 on the device) instead of the sampled records. Returns the row length (bytes needed excluding NUL)
 * or a negative code; SCFQ_EARG also when the first header has too few ':' fields (IndexError in the reference). */
#define SCFQ_META_WHOLE_FILE 0x1u
const char* scfq_meta_header(void);
int scfq_meta_file_tsv(const char* path, uint32_t sample_n, uint32_t flags, char* out, uint64_t cap);

/* ---- formatting: src/fq_count.nim:47-51 ---------------------------------------------------
 * "<reads>\t<gc_content>\t<gc_bases>\t<n_bases>\t<bases>" without trailing newline;
 * gc_content = gc/(bases-n) as IEEE double printed the way Nim 1.0.6 `$float` does: C "%.16g",
 * then ".0" appended when the text has no '.', no letter; NaN prints "nan".
 * Returns the number of bytes needed (excluding NUL); writes at most cap bytes incl. NUL. */
int scfq_format_tsv(const scfq_counts* c, char* buf, uint64_t cap);

const char* scfq_strerror(int rc);   /* static storage */
const char* scfq_last_error_detail(void); /* static, thread-local: e.g. the failing HIP call */
int  scfq_last_timing(scfq_timing* t);
/* Device memory the library's ingest paths hold in this process (staging, inflate buffers, symbol pools): now, and the most
 * they ever held. The buffers are kept between calls and given back by scfq_shutdown(). */
uint64_t scfq_device_bytes_now(void);
uint64_t scfq_device_bytes_high_water(void);
int  scfq_device_count(void);        /* number of visible HIP devices, or negative on error */
/* Default caller stream of THIS host thread for the entry points that take device pointers but no scfq_opts
 * (scfq_index_lines, scfq_dedup_buffer) and for calls without SCFQ_WAIT_STREAM: same contract as scfq_opts.wait_stream.
 * enable = 0 (the initial state): the caller synchronises before calling. hip_stream = NULL with enable != 0 names the
 * legacy default stream. scfq_get_wait_stream returns the stream, *enabled (may be NULL) whether one is set. */
int   scfq_set_wait_stream(void* hip_stream, int enable);
void* scfq_get_wait_stream(int* enabled);
int  scfq_shutdown(void);            /* frees streams, pinned and device scratch; safe to call twice */

/* Diagnostic entry points (scfq_debug_*: a second device implementation for the parity tests, host-only readers, the
 * stream-ceiling probe) are declared in sc_fqcount_debug.h; a reference-side binding needs none of them. */

/* ---- synthetic workloads of SURVEY.md §8(d) / BASELINE.json configs ------------------------
 * Counter-based generator: record i of a workload is a pure function of (seed, i), so host and
 * device produce identical bytes and any shard can be produced independently.
 * kind: 0 = Illumina 150 bp (config 2/3), 1 = Nanopore-style 500 bp..50 kb (config 5). */
#define SCFQ_SYNTH_ILLUMINA 0
#define SCFQ_SYNTH_NANOPORE 1
typedef struct scfq_synth_info {
  uint64_t struct_size;
  uint64_t records;      /* records generated */
  uint64_t bytes;        /* exact byte length of those records */
  uint64_t gc_bases, n_bases, bases;  /* tallied by the generator itself, independent of the scan */
} scfq_synth_info;

/* Smallest record count whose total length is >= min_bytes (and its exact length). */
int scfq_synth_plan(int kind, uint64_t seed, uint64_t first_record, uint64_t min_bytes, scfq_synth_info* info);
/* Locate byte `offset` of the record stream that starts at record 0: writes the index of the record
 * containing that byte and the stream offset at which that record starts. */
int scfq_synth_locate(int kind, uint64_t seed, uint64_t offset, uint64_t* record, uint64_t* record_start);
/* Generate records [first_record, first_record+records) into host memory (cap >= info->bytes). */
int scfq_synth_host(int kind, uint64_t seed, uint64_t first_record, uint64_t records,
                    void* dst, uint64_t cap, scfq_synth_info* info);
/* Same bytes, produced by a HIP kernel directly in HBM (dst is a device pointer). */
int scfq_synth_device(int kind, uint64_t seed, uint64_t first_record, uint64_t records,
                      void* dst_device, uint64_t cap, scfq_synth_info* info);

#ifdef __cplusplus
}
#endif
#endif /* SC_FQCOUNT_H */
