// scfq_gzdev.hpp — host side of the device gzip inflate (kernels: gz_inflate_kernels.hpp).  Included by scfq_api.hip
// inside its anonymous namespace (uses Ctx, HIPCHK, scan_async, parallel_pieces, kFallbackToHost).
//
// ingest_gz_device(): the COMPRESSED file goes through a pinned ring to HBM in BATCHES of 4096 planned segments (64 KiB of compressed
// data each for files beyond 512 MiB), and the stages of several batches are in flight at once (r5: an event-driven schedule, see "the pipeline"):
//      copy        a copier thread of its own: file -> pinned ring -> one of slots + 2 compressed-byte buffers (copy stream)
//      search      block starts behind every planned segment start (search stream, high priority)
//      decode      one wave per segment, to 16-bit symbols in one of three symbol sets (decode stream)
//      walk + post chain walk (host), window maps / chain, resolve, CRC-32 tiles, scan (compute stream)
// so the PCIe copy and everything behind the decode hide under the decodes, and the device memory in use is that of three
// batches whatever the size of the file (the one-batch form of this path needed 14 x the compressed size in HBM).
// A decoded segment is only believed when the walk reaches its start bit EXACTLY — from the member's first block, through
// every segment's end bit, over member trailers and headers, across batch borders — so a false sync, a corrupt block or a
// truncated file can never contribute a byte: the walk stops there, gaps it can prove (a false sync skipped, the first block
// of a further member) are decoded in a further round, and anything else hands the whole file to the host readers
// (kFallbackToHost; the caller starts its session again, so what earlier batches folded is dropped), which reproduce
// gzread's behaviour for it byte by byte (partial output, SCFQ_EGZ).  CRC-32 and ISIZE of every member are checked before
// the result is handed out.

// one half of the pinned ring the compressed bytes of a file cross in (see ingest_gz_device_batches: "the pinned ring")
inline uint64_t gz_ring_piece(uint64_t comp_bytes, bool context_is_new) {
  static const int ring_env = env_int("SCFQ_GZ_DEVICE_RING_MB", 0);
  // Files of up to 1 GiB compressed cross in 16 MiB pieces (r4: pinning 2 x 64 MiB costs a process 26 ms, 2 x 16 MiB 5 ms — for a 0.5 GB file
  // the bigger pieces saved 6 ms of copying).  Beyond 1 GiB: 128 MiB pieces — but not in a context's FIRST session, a process's only one,
  // which stays on the context's 2 x 16 MiB (r5): a process over the 10 GB member takes the same 190 - 260 ms from "context up" to "folded"
  // with pieces of 16, 32 or 64 MiB (profiles/r05/cold_ring_ab.txt; its pipeline is not bound by the copy), the runtime does nothing else
  // while it pins — not even queue another thread's copies (profiles/r05/cold_marks.txt) — and a second pinned allocation is 22 - 30 ms.
  // A long-running host's later sessions get the big pieces, worth 13 ms per 10 GB once everything else is warm.
  return ring_env > 0 ? ((uint64_t)std::min(256, std::max(4, ring_env)) << 20)
                      : (comp_bytes > (1ull << 30) && !context_is_new) ? (128ull << 20) : (16ull << 20);
}

inline bool gz_device_enabled() {
  static const bool v = [] { const char* e = std::getenv("SCFQ_GZ_DEVICE"); return e ? e[0] != '0' : true; }();
  return v;
}

// every "not on the device" decision says where it was taken (SCFQ_VERBOSE)
inline int gz_decline(int line, int code = kFallbackToHost) {
  if (trace_on()) std::fprintf(stderr, "scfq gzdev: declined at scfq_gzdev.hpp:%d\n", line);
  return code;
}
#define SCFQ_GZ_DECLINE gz_decline(__LINE__)

template <typename T>
int gz_grow(T** p, uint64_t* cap, uint64_t want_bytes, bool pinned = false) {
  if (*cap >= want_bytes) return SCFQ_OK;
  if (*p) { if (pinned) (void)hipHostFree(*p); else (void)hipFree(*p); }
  *p = nullptr;
  *cap = 0;
  const uint64_t bytes = want_bytes + want_bytes / 8 + 4096;
  const hipError_t e = pinned ? hipHostMalloc(reinterpret_cast<void**>(p), bytes, hipHostMallocDefault) : hipMalloc(reinterpret_cast<void**>(p), bytes);
  if (e != hipSuccess) { (void)hipGetLastError(); *p = nullptr; return SCFQ_GZ_DECLINE; }      // not enough memory: the host path needs none of this
  *cap = bytes;
  return SCFQ_OK;
}

// a device buffer made big enough; kFallbackToHost when the device cannot give the memory (the host path needs none of this)
inline double& gz_alloc_ms() { static thread_local double v = 0; return v; }      // host time inside allocations during the current call (SCFQ_VERBOSE)
inline int gz_buf(GzDevBuffers& g, DevBuf& b, uint64_t want) {
  int64_t delta = 0;
  const auto t0 = std::chrono::steady_clock::now();
  const hipError_t e = b.ensure(want, &g.retired, &delta);
  gz_alloc_ms() += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  note_dev_bytes(delta);
  if (e != hipSuccess && trace_on()) std::fprintf(stderr, "scfq gzdev: no device memory for %.3f GB (%s)\n", (double)want / 1e9, hipGetErrorString(e));
  return e == hipSuccess ? SCFQ_OK : kFallbackToHost;
}
inline int gz_take(SymPool& pool, uint64_t n_syms, uint64_t* off) {
  int64_t delta = 0;
  const auto t0 = std::chrono::steady_clock::now();
  const hipError_t e = pool.take(n_syms, off, &delta);
  gz_alloc_ms() += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  note_dev_bytes(delta);
  if (e != hipSuccess && trace_on()) std::fprintf(stderr, "scfq gzdev: no device memory for another %.1f GB of symbols (%s)\n", 2.0 * (double)n_syms / 1e9, hipGetErrorString(e));
  return e == hipSuccess ? SCFQ_OK : kFallbackToHost;
}
// buffers that were replaced during a call (a kernel in flight may still have been reading them): the device is idle now
inline void gz_free_retired(GzDevBuffers* g) {
  for (void* p : g->retired) (void)hipFree(p);
  g->retired.clear();
}

inline void gz_free(GzDevBuffers* g) {
  note_dev_bytes(-(int64_t)g->held());
  gz_free_retired(g);
  for (int b = 0; b < kGzMaxComp; ++b) {
    g->comp[b].release();
    if (g->ev_copy[b]) (void)hipEventDestroy(g->ev_copy[b]);
    if (g->ev_found[b]) (void)hipEventDestroy(g->ev_found[b]);
    g->ev_copy[b] = g->ev_found[b] = nullptr;
    g->d_search[b] = g->h_search[b] = nullptr;
  }
  for (int b = 0; b < kGzMaxSlots; ++b) {
    g->slot[b].sym.release();
    g->slot[b].d_meta = g->slot[b].h_meta = nullptr;
    g->d_pmeta[b] = g->h_pmeta[b] = nullptr;
    if (g->ev_dec[b]) (void)hipEventDestroy(g->ev_dec[b]);
    if (g->ev_post[b]) (void)hipEventDestroy(g->ev_post[b]);
    g->ev_dec[b] = g->ev_post[b] = nullptr;
  }
  for (int b = 0; b < kGzDecodeStreams; ++b) {
    if (g->s_decode[b]) (void)hipStreamDestroy(g->s_decode[b]);
    g->s_decode[b] = nullptr;
  }
  for (int b = 0; b < 2; ++b) {
    if (g->fg_stage[b]) (void)hipFree(g->fg_stage[b]);
    g->fg_stage[b] = nullptr;
    if (g->ev_fg[b]) (void)hipEventDestroy(g->ev_fg[b]);
    g->ev_fg[b] = nullptr;
  }
  g->fg_cap = 0;
  if (g->h_ring) (void)hipHostFree(g->h_ring);
  g->h_ring = nullptr; g->ring_piece = 0;
  for (int b = 0; b < 2; ++b) { if (g->ev_ring[b]) (void)hipEventDestroy(g->ev_ring[b]); g->ev_ring[b] = nullptr; }
  if (g->d_tables) (void)hipFree(g->d_tables);
  if (g->h_tables) (void)hipHostFree(g->h_tables);
  g->d_tables = g->h_tables = g->h_crc = nullptr;
  g->tables_cap = g->htables_cap = 0;
  g->decode_warmed = false;
  g->copies_warmed = false;
  g->out.release(); g->win.release(); g->maps.release(); g->gwin.release(); g->crc.release();
  if (g->d_wcarry) (void)hipFree(g->d_wcarry);
  g->d_wcarry = nullptr;
  g->wcarry_cap = 0;
  if (g->s_search) (void)hipStreamDestroy(g->s_search);
  if (g->s_gap) (void)hipStreamDestroy(g->s_gap);
  g->s_search = g->s_gap = nullptr;
}

// x^(8 n) mod P and products in the reflected representation (as the kernels: bit 31 = x^0)
inline uint32_t gz_mulmod(uint32_t a, uint32_t b) {
  uint32_t p = 0;
  for (int k = 0; k < 32; ++k) {
    p ^= b & (0u - ((a >> (31 - k)) & 1u));
    b = (b >> 1) ^ (0xEDB88320u & (0u - (b & 1u)));
  }
  return p;
}
inline uint32_t gz_xpow8n(uint64_t n) {
  uint32_t p = 1u << 31, sq = 1u << 30;
  for (uint64_t bits = n << 3; bits; bits >>= 1) {
    if (bits & 1u) p = gz_mulmod(sq, p);
    sq = gz_mulmod(sq, sq);
  }
  return p;
}

struct GzPart {            // the bytes of one member inside one batch: a run of CRC tiles
  uint32_t member;         // running member number
  uint64_t len;            // bytes
  uint64_t tile_at, n_tiles;
};
struct GzMemberEnd { uint32_t member; uint32_t crc, isize; };

// A STRETCH of one member's deflate stream, from one block boundary to another: what one rank of a sharded count works on when the
// file is a single member (scfq_count_file_sharded).  The member's bytes in front of the stretch are not known when the rank starts,
// so it goes over its stretch twice:
//   pass 1 (map_only)  search, decode to symbols, prove the chain from start_bit to stop_bit exactly — and fold what the stretch does
//                      to the 32 KiB window into ONE map (gz_window_maps per group of chain entries, gz_map_fold over the groups).
//                      The ranks exchange their maps; composed in rank order from the member's start they give every rank the window
//                      in front of its stretch.
//   pass 2             the ordinary pipeline — windows, bytes, CRC tiles, scan — begun at start_bit with that window.
// The member's CRC-32 and length are the join of the stretches' (x^(8 |part|) as between batches), checked by the caller.
struct GzStretch {
  uint64_t start_bit = 0;          // the stretch's first block (0: it begins the member, behind its header)
  uint64_t stop_bit = 0;           // the block boundary it ends at (0: it ends with the member)
  bool map_only = false;
  const uint8_t* window = nullptr; // pass 2: the 32 KiB in front of start_bit ...
  uint32_t valid = 0;              // ... of which the last `valid` bytes exist (32768 once the member is that long)
  // results
  uint64_t out_bytes = 0;          // bytes the stretch inflates to
  bool member_ended = false;       // the member's final block lies in the stretch (stop_bit == 0 only)
  uint32_t trailer_crc = 0, trailer_isize = 0;      // ... and its trailer
  uint64_t end_byte = 0;           // ... and the offset just behind it
  std::vector<uint16_t> map;       // pass 1: 32768 symbols — a literal, or 0x8000 | j = byte j of the window in front of the stretch
  uint32_t crc_raw = 0;            // pass 2: raw CRC-32 (zero initial value, no final inversion) of the stretch's bytes
  int first_byte = -1;             // pass 2
  // ONE pass instead of two: with `exchange` set the call keeps the stretch's proven symbols (packed: two bytes per inflated byte), and when
  // its map is known calls exchange(*this, ok) — ok false when the stretch could not be proven: the callback is a collective, every rank
  // must make it —, which fills `window` / `valid` from the other ranks' maps (0: go on; else give up); then the symbols become bytes.
  std::function<int(GzStretch&, bool)> exchange;
};

int ingest_gz_device_batches(Ctx* c, GzDevBuffers& g, const uint8_t* img, uint64_t fsize, uint32_t flags, bool timing, uint64_t* end_off, int fd, uint64_t fd_off, GzStretch* sx) {
  FileBytes fbytes;
  fbytes.img = img; fbytes.fd = fd; fbytes.fd_off = fd_off;
  using namespace scfq_dinflate;
  using clk = std::chrono::steady_clock;
  const auto t_begin = clk::now();
  gz_alloc_ms() = 0;
  double stage_ms[3] = {0, 0, 0};
  static const bool verbose = std::getenv("SCFQ_VERBOSE") != nullptr;
  const bool keep = sx && (bool)sx->exchange;      // one pass: the symbols are kept until the other ranks' maps have given the window
  const bool stretch_from = sx && sx->start_bit != 0, stretch_to = sx && sx->stop_bit != 0, map_only = sx && (sx->map_only || keep);
  // (every way out of this function in front of the exchange makes it, with "not proven": the exchange is a collective)
  struct ExchangeGuard { GzStretch* sx; bool made = false; ~ExchangeGuard() { if (sx && !made) (void)sx->exchange(*sx, false); } } exchange_guard{keep ? sx : nullptr};
  const long h0 = stretch_from ? (long)(sx->start_bit >> 3) : scfq_gzfast::member_header(img, (size_t)fsize);
  if (h0 <= 0 || fsize < 64) return SCFQ_GZ_DECLINE;
  if (sx && ((stretch_to && (sx->stop_bit <= sx->start_bit || (sx->stop_bit >> 3) >= fsize)) || (uint64_t)h0 + 64 > fsize)) return SCFQ_GZ_DECLINE;
  // (a stretch that ends at a block boundary: the planner sees the file end there; the copies still reach a margin further, so that
  // the last segment's block is whole)
  const uint64_t lim_byte = stretch_to ? std::min<uint64_t>(fsize, (sx->stop_bit >> 3) + 1) : fsize;
  int rc;
  for (int b = 0; b < kGzMaxComp; ++b) {
    if (!g.ev_copy[b]) HIPCHK(hipEventCreateWithFlags(&g.ev_copy[b], hipEventDisableTiming));
    if (!g.ev_found[b]) HIPCHK(hipEventCreateWithFlags(&g.ev_found[b], hipEventDisableTiming));
  }
  for (int b = 0; b < kGzMaxSlots; ++b) {
    if (!g.ev_dec[b]) HIPCHK(hipEventCreateWithFlags(&g.ev_dec[b], hipEventDisableTiming));
    if (!g.ev_post[b]) HIPCHK(hipEventCreateWithFlags(&g.ev_post[b], hipEventDisableTiming));
  }
  // SETS OF SYMBOLS in flight (SCFQ_GZ_DEVICE_SLOTS = 2 .. 6).  A set is held from a batch's decode to its resolve; the decode of batch k + slots
  // waits for the resolve of batch k.  The decode kernels are what a warm call consists of, and a decode kernel alone leaves the device
  // half empty: a batch is ~3700 waves on 5120 slots and lasts as long as its slowest wave (one or two deflate blocks per segment) — the
  // whole 10 GB member as ONE dispatch of 37066 segments, longest first, takes 53.7 ms where ten batch dispatches one after the other take
  // 73 (profiles/r05/gz_one_batch.txt).  With the event-driven schedule below, every further set lets one more decode kernel be queued
  // behind the running ones, so retiring waves are replaced at once; what it costs is 2.4 GB of device memory per set, which a PROCESS
  // pays for when it exits (the driver wipes what is freed) — so a context's first session, a process's only one, takes fewer sets.
  static const int slots_env = env_int("SCFQ_GZ_DEVICE_SLOTS", 0);
  const uint32_t n_slots = (uint32_t)std::min<int>(kGzMaxSlots, std::max(2, slots_env > 0 ? slots_env : (c->n_sessions <= 1 ? env_int("SCFQ_GZ_DEVICE_SLOTS_FIRST", 3) : 3)));
  const uint32_t n_comp = std::min<uint32_t>(kGzMaxComp, n_slots + 2);      // compressed-byte buffers: the copy and the search run two batches ahead of the decodes

  // ---- plan ----------------------------------------------------------------------------------------------------------------
  const uint64_t data0 = (uint64_t)h0;
  const uint64_t first_bit = stretch_from ? sx->start_bit : data0 * 8;      // the chain's first block
  const uint64_t comp = lim_byte - data0;
  // Segments of 48 to 320 KiB of compressed data (below); batches of about 4096 segments — the decode kernel's resident
  // waves — of equal size.  Two batches in flight is what pays (10 GB of FASTQ, 2.4 GB
  // compressed: one batch 199 ms, two 165 ms, five 217 ms): the second half of the file crosses PCIe while the first is decoded,
  // and a batch's windows, bytes, CRC and scan run under the next one's decode; with more, smaller batches the copies and short
  // kernels crawl between the long-running decode waves.
  static const int seg_kb_env = env_int("SCFQ_GZ_DEVICE_SEGMENT_KB", 0);
  static const uint32_t batch_segs = (uint32_t)std::max(2, env_int("SCFQ_GZ_DEVICE_BATCH_SEGMENTS", 4096));
  // (a file of up to 512 MiB is cut into about 4096 segments — one wave each, the device filled once — of at least 48 KiB; a
  // bigger one into about 8192 of at most 320 KiB: every segment costs 32768 marker symbols and a step of the window chain)
  // (r3: at most 64 KiB.  Bigger segments are a little faster once the buffers exist — 10 GB: 155 ms with 292 KiB segments in two
  // batches, 166 ms with 128 KiB, 177 ms with 64 KiB — and hold 39 / 18 / 10 GB of device memory for it.  That memory is what a
  // PROCESS pays for: the driver wipes what a process frees at ~20 GB/s after it exits and the next process's allocations wait for
  // it, so `for f in *.gz; do sc fq-count $f; done` spends 1.7 / 0.7 / 0.3 s per file waiting: profiles/r03/gz_segment_size.txt,
  // gz_cold.jsonl)
  // (r5, with the feed that writes device memory from the host's threads: 128 KiB segments — batches of 512 MiB, half the window-chain steps and
  // marker symbols per byte — take a warm 10 GB call from 88.7 - 91.8 ms to 84.8 - 86.9, 96 / 160 KiB 87 - 88, 192 KiB 94 - 95, and hold 24.4 GB
  // where 64 KiB hold 14.7: profiles/r05/gz_final_knobs_ab.txt.  A long-running host takes them from its context's second session on, for a
  // whole file of more than 1 GiB, when the device has the room; a process's one session keeps 64 KiB — it pays for memory at its exit.)
  const uint64_t target_segs = 4096;
  bool wide_segments = false;
  if (seg_kb_env <= 0 && !stretch_from && comp > (1ull << 30) && c->n_sessions > 1) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = 0; }
    static const int test_free_gb = env_int("SCFQ_TEST_DEVICE_FREE_GB", -1);
    if (test_free_gb >= 0) free_b = std::min<size_t>(free_b, (size_t)test_free_gb << 30);
    wide_segments = (uint64_t)free_b + g.held() >= (48ull << 30);
  }
  const uint64_t seg_bytes = seg_kb_env > 0 ? ((((uint64_t)std::max(32, seg_kb_env)) << 10) + 4095) & ~4095ull
                                            : comp <= (512ull << 20) ? std::min<uint64_t>(128u << 10, std::max<uint64_t>(16u << 10, ((comp + target_segs - 1) / target_segs + 4095) & ~4095ull))
                                                                     : wide_segments ? (128u << 10) : (64u << 10);      // (up to 512 MiB: one batch of ~4096 segments, the device filled once, <= 9 GB held)
  // Output room of a segment = `ratio_est` symbols per compressed byte it spans + 128 Ki (it runs on to the end of a block),
  // behind its 32768 markers.  ratio_est comes from the file itself: the host inflates the first 192 KiB of the first member
  // (a millisecond) and adds a third; a segment that needs more ends with kGzErrOverflow and is decoded again, alone, with four
  // times the room (the walk below treats it like a gap) — round 2 sized every segment for 7 symbols per byte, held 50 - 70 GB for
  // a 10 GB file, and sent the whole file to the host when one segment compressed better than that.
  static const double ratio_env = std::atof(std::getenv("SCFQ_GZ_DEVICE_RATIO") ? std::getenv("SCFQ_GZ_DEVICE_RATIO") : "0");
  double ratio_est = ratio_env > 0 ? ratio_env : 6.0;
  uint32_t lit_mask = 0;                 // 32-value classes of bytes the file's first 192 KiB (inflated: a MB or so) do not hold: the search's plausibility test
  {
    const uint64_t sample_in = std::min<uint64_t>(fsize - data0, 192u << 10);
    std::vector<uint8_t> sample_out((size_t)kGzWindow + (8u << 20));
    auto dec = std::unique_ptr<scfq_inflate::Decoder>(new scfq_inflate::Decoder());
    if (stretch_from) { dec->begin_at_bit(img, img + data0 + sample_in, first_bit); dec->total_out = kGzWindow; }      // (back-references reach into a window of zeros: the ratio is what is wanted)
    else dec->begin(img + data0, img + data0 + sample_in);
    uint8_t* o = sample_out.data() + kGzWindow;
    (void)dec->run(o, sample_out.data() + sample_out.size());      // (ends with "truncated" at the end of the sample: what came out counts)
    const uint64_t got_out = (uint64_t)(o - (sample_out.data() + kGzWindow)), got_in = std::max<uint64_t>(1, (dec->bitpos() - (stretch_from ? first_bit : 0)) / 8);
    // (long runs — zeros, one record repeated: ratios in the hundreds — are the host decoder's home ground, gigabytes per second,
    // and the lane-parallel loop's worst case, one symbol per round: such a file is declined here, before anything is queued)
    if (ratio_env <= 0 && got_in >= 1024 && (double)got_out / (double)got_in > 24.0) return SCFQ_GZ_DECLINE;
    static const double est_mul = std::atof(std::getenv("SCFQ_GZ_DEVICE_RATIO_MARGIN") ? std::getenv("SCFQ_GZ_DEVICE_RATIO_MARGIN") : "1.15");
    if (ratio_env <= 0 && got_out >= 65536 && got_in >= 4096) ratio_est = std::min(32.0, std::max(3.0, est_mul * (double)got_out / (double)got_in + 0.25));
    static const bool use_mask = env_int("SCFQ_GZ_DEVICE_LITERAL_MASK", 1) != 0;
    if (use_mask && got_out >= 65536) {
      // (only the upper half of the alphabet is ever ruled out: FASTQ is ASCII by definition, while which letters or digits a file
      // uses may well change after its first megabyte — lowercase soft-masked bases, another quality range)
      uint32_t seen = 0;
      const uint8_t* so = sample_out.data() + kGzWindow;
      for (uint64_t q = 0; q < got_out; ++q) seen |= 1u << (so[q] >> 5);
      lit_mask = (seen & 0xF0u) ? 0u : 0xF0u;
    }
  }
  const uint64_t n_plan = std::max<uint64_t>(1, (comp + seg_bytes - 1) / seg_bytes);
  std::vector<uint64_t> bstart{0};                     // planned segments [bstart[k], bstart[k + 1]) make batch k
  {
    const uint64_t n_batches = (n_plan + batch_segs - 1) / batch_segs, per = (n_plan + n_batches - 1) / n_batches;
    for (uint64_t at = 0; at < n_plan;) { at = std::min<uint64_t>(n_plan, at + per); bstart.push_back(at); }
    // The first batch cut in two, a quarter and the rest (SCFQ_GZ_DEVICE_FIRST_BATCH_DIV=1: not): the first decode starts after 64 MiB have
    // crossed PCIe instead of 256.  A warm call is bound by the copy and does not notice (r3 / r4 measured it slightly slower with their
    // schedules: 160.6 against 156.4 ms); a process's first call, whose pipeline starts 50 ms of setup late, gets its first kernel ~8 ms sooner.
    static const int first_div = std::max(1, env_int("SCFQ_GZ_DEVICE_FIRST_BATCH_DIV", 1));
    if (n_batches >= 3 && first_div > 1 && bstart[1] / first_div >= 64) bstart.insert(bstart.begin() + 1, bstart[1] / first_div);
  }
  const uint32_t nb = (uint32_t)bstart.size() - 1;
  const uint64_t margin = 4ull << 20;                  // a batch's last segment runs on to the end of its block
  const uint64_t comp_pad = 256;
  const uint64_t end_bit = stretch_to ? sx->stop_bit : fsize * 8;
  // geometry of batch k: bytes [byte0, copy_end) of the file are on the device, its territory ends at byte1
  auto p0_of = [&](uint32_t k) { return bstart[k]; };
  auto p1_of = [&](uint32_t k) { return bstart[k + 1]; };
  auto byte0_of = [&](uint32_t k) { return k == 0 ? (stretch_from ? (data0 & ~4095ull) : 0ull) : ((data0 + p0_of(k) * seg_bytes) & ~4095ull); };
  auto byte1_of = [&](uint32_t k) { return p1_of(k) == n_plan ? lim_byte : data0 + p1_of(k) * seg_bytes; };
  auto copy_end_of = [&](uint32_t k) { return std::min<uint64_t>(fsize, byte1_of(k) + margin); };
  const uint64_t batch_comp_max = std::min<uint64_t>(fsize, (uint64_t)std::min<uint64_t>(batch_segs, n_plan) * seg_bytes + margin + 8192 + (stretch_from ? 4096 : 0));
  // SCFQ_GZ_DEVICE_HOST_WRITES=1 (files of several batches): no pinned ring and no copy engine between the file and the device — the copier's
  // threads pread a batch's bytes straight into one of two FINE-GRAINED device buffers (mapped into the process: posted writes over PCIe, one
  // pass over host memory), and a device-to-device copy moves them into the batch's ordinary buffer, which is what the kernels read (read in
  // place, uncached, the decode is 12 % slower: profiles/r05/gz_host_writes_ab.txt)
  static const bool host_writes_env = env_int("SCFQ_GZ_DEVICE_HOST_WRITES", 1) != 0;
  bool host_writes = host_writes_env && nb > 1;
  // (r5, late) A context's FIRST session stages pieces of 32 MiB, not whole batches: the first write to a page of such a mapping is a fault, and a
  // process's first two batches took 40 and 18 ms to read instead of 6.5 with 2 x 290 MiB to touch — its session 144 ms instead of 156.  Later
  // sessions stage a batch at a time (one device-to-device copy, no waits between pieces: a warm call 85 - 86 ms against 88 - 89 with 32 or 64 MiB
  // pieces): profiles/r05/cold_staging_pieces_ab.txt.  SCFQ_GZ_DEVICE_STAGING_MB = n: pieces of n MiB always; 0: always a batch.
  static const int staging_mb = env_int("SCFQ_GZ_DEVICE_STAGING_MB", -1);
  const uint64_t fg_piece = staging_mb > 0 ? ((uint64_t)std::max(8, staging_mb) << 20)
                          : (staging_mb < 0 && c->n_sessions <= 1) ? (32ull << 20) : ((batch_comp_max + 4095) & ~4095ull);
  if (host_writes) {
    const uint64_t want = fg_piece + comp_pad + 4096;
    if (g.fg_cap < want) {
      for (int b = 0; b < 2; ++b) { if (g.fg_stage[b]) { g.retired.push_back(g.fg_stage[b]); g.fg_stage[b] = nullptr; } }
      note_dev_bytes(-(int64_t)(2 * g.fg_cap));
      g.fg_cap = 0;
      const uint64_t bytes = (want + want / 8 + 4095) & ~4095ull;
      for (int b = 0; b < 2 && host_writes; ++b)
        if (hipExtMallocWithFlags(reinterpret_cast<void**>(&g.fg_stage[b]), bytes, hipDeviceMallocFinegrained) != hipSuccess) { (void)hipGetLastError(); g.fg_stage[b] = nullptr; host_writes = false; }
      if (host_writes) { g.fg_cap = bytes; note_dev_bytes((int64_t)(2 * bytes)); }
      else for (int b = 0; b < 2; ++b) { if (g.fg_stage[b]) (void)hipFree(g.fg_stage[b]); g.fg_stage[b] = nullptr; }
    }
    for (int b = 0; b < 2 && host_writes; ++b) if (!g.ev_fg[b]) HIPCHK(hipEventCreateWithFlags(&g.ev_fg[b], hipEventDisableTiming));
  }
  // (r5) A process's first call spends 25 - 30 ms between here and the copier thread's start — three streams, a dozen allocations — and the
  // first batch's bytes need nothing of that: a thread reads them into the first staging buffer meanwhile (page cache to device memory,
  // no runtime call), and the copier, when it gets to batch 0, only waits for it (profiles/r05/cold_prefill_ab.txt).
  struct Prefill { std::thread th; ~Prefill() { if (th.joinable()) th.join(); } } prefill;
  static const bool prefill_env = env_int("SCFQ_GZ_DEVICE_PREFILL", 1) != 0;
  uint32_t n_prefilled = 0;              // pieces of batch 0 the thread reads: both staging buffers' worth
  if (host_writes && prefill_env) {
    const uint64_t pb0 = byte0_of(0), pb1 = copy_end_of(0);
    n_prefilled = (uint32_t)std::min<uint64_t>(2, (pb1 - pb0 + fg_piece - 1) / fg_piece);
    uint8_t* const dst0 = g.fg_stage[0];
    uint8_t* const dst1 = g.fg_stage[1];
    const FileBytes fb = fbytes;
    const uint32_t np = n_prefilled;
    prefill.th = std::thread([fb, pb0, pb1, dst0, dst1, np, fg_piece] {
      for (uint32_t i = 0; i < np; ++i) {
        const uint64_t off = pb0 + (uint64_t)i * fg_piece, len = std::min<uint64_t>(fg_piece, pb1 - off);
        uint8_t* const d = i ? dst1 : dst0;
        copy_file_bytes(fb, off, d, len, true);
        if (off + len == pb1) std::memset(d + len, 0, 256);      // (comp_pad, behind the batch's last byte)
      }
      std::atomic_thread_fence(std::memory_order_seq_cst);
      trace("gzip engine: the first pieces of the first batch are in the staging buffers");
    });
    trace("gzip engine: the first batch's bytes are being read (staging buffers allocated)");
  }
  // ---- streams.  A file of ONE batch is a chain — copy, search, decode, walk, windows, bytes, scan — and runs on the context's two
  // streams (copy + search on one, decode and everything behind it on the other): creating a stream costs 10 - 15 ms, and an
  // engine's three were half of what a small file's first call paid.  Files of several batches get the engine's own streams: the
  // search of batch k + 1, the decodes of two batches and the post-processing of a third overlap.
  if (nb > 1 && (rc = want_copy_stream(c))) return rc;      // (a file of one batch is a chain: the context's one stream carries all of it)
  // ONE decode stream to begin with (SCFQ_GZ_DEVICE_DECODE_STREAMS = 1 .. 3): the batches arrive at the rate the compressed bytes cross PCIe, one per
  // 7 ms, which is what a decode kernel lasts — in order on one stream they run back to back; on two or three streams (r4: two) the call
  // is no shorter (88 - 93 ms either way on one box, profiles/r05/gz_schedule_ab.txt) and a process pays 10 - 20 ms per stream it creates.
  // (a context's first session: one; later sessions add a second, so that a faster link than this box's finds the decode kernels overlapping)
  static const int dec_env = env_int("SCFQ_GZ_DEVICE_DECODE_STREAMS", 0);
  const uint32_t n_dec_streams = (uint32_t)std::min<int>(kGzDecodeStreams, std::max(1, dec_env > 0 ? dec_env : (c->n_sessions <= 1 ? 1 : 2)));
  hipStream_t s_search = c->copy, s_dec[kGzDecodeStreams] = {c->compute, c->compute, c->compute};
  if (nb > 1) {
    if (!g.s_search) {
      // the search's workgroups are small and short, and the decode of the next batch cannot be cut into segments before they are
      // through: their stream has priority over the decode streams, whose waves run for tens of milliseconds
      int least = 0, greatest = 0;
      static const int hi = env_int("SCFQ_GZ_DEVICE_SEARCH_PRIORITY", 1);
      if (hi && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest) HIPCHK(hipStreamCreateWithPriority(&g.s_search, hipStreamNonBlocking, greatest));
      else HIPCHK(hipStreamCreateWithFlags(&g.s_search, hipStreamNonBlocking));
    }
    for (uint32_t b = 0; b < n_dec_streams; ++b) {
      if (!g.s_decode[b]) {
        // The decode kernels fill the device with waves that run for tens of milliseconds, and whatever else the pipeline launches
        // meanwhile (copies, search, window chain, resolve, CRC, scan) has to get in between.  SCFQ_GZ_DEVICE_RESERVE_CUS=n keeps
        // the decode streams off n CUs (a CU mask); measured, that costs the decode more than it gives the rest: off by default.
        static const int reserve = std::max(0, env_int("SCFQ_GZ_DEVICE_RESERVE_CUS", 0));
        hipError_t e = hipErrorNotSupported;
        if (reserve > 0 && c->n_cu > 2 * reserve) {
          std::vector<uint32_t> mask((size_t)(c->n_cu + 31) / 32, 0u);
          // (mask bit 32 x + j is CU j of XCD x, and workgroups go round the XCDs in turn: the same number of CUs is left out in every XCD,
          // or the XCD that lost most sets the pace — measured: 16 CUs taken from one XCD made the decode 1.6 x slower)
          static const int layout = env_int("SCFQ_GZ_DEVICE_MASK_LAYOUT", 1);
          const int per = c->n_cu / 8, r = (reserve + 7) / 8;
          for (int i = 0; i < c->n_cu; ++i) {
            const bool keep = (layout == 1 && c->n_cu % 8 == 0) ? (i % per) < per - r : i < c->n_cu - reserve;
            if (keep) mask[(size_t)i >> 5] |= 1u << (i & 31);
          }
          e = hipExtStreamCreateWithCUMask(&g.s_decode[b], (uint32_t)mask.size(), mask.data());
          if (e != hipSuccess) { (void)hipGetLastError(); g.s_decode[b] = nullptr; }
        }
        if (e != hipSuccess) {
          static const int low = env_int("SCFQ_GZ_DEVICE_DECODE_LOW_PRIORITY", 0);
          int least = 0, greatest = 0;
          if (low && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest) HIPCHK(hipStreamCreateWithPriority(&g.s_decode[b], hipStreamNonBlocking, least));
          else HIPCHK(hipStreamCreateWithFlags(&g.s_decode[b], hipStreamNonBlocking));
        }
      }
      s_dec[b] = g.s_decode[b];
    }
    for (uint32_t b = n_dec_streams; b < (uint32_t)kGzDecodeStreams; ++b) s_dec[b] = s_dec[b % n_dec_streams];
    s_search = g.s_search;
  }

  // The decode kernel keeps 72 bytes of scratch per lane (its symbol loop is a real function call), and the runtime sets the
  // device's scratch up inside the FIRST launch of such a kernel: 35 - 50 ms of a process's first call, spent on the host thread.
  // An empty launch on a helper thread takes that off the critical path: it runs under the tables' and the pinned ring's allocation
  // and the first batch's copy.
  struct Warm {
    std::thread th;
    ~Warm() { if (th.joinable()) th.join(); }
  } warm;
  if (!g.decode_warmed) {
    g.decode_warmed = true;
    hipStream_t sd = s_dec[0];
    const int dev = c->dev;
    warm.th = std::thread([sd, dev] {
      if (hipSetDevice(dev) != hipSuccess) return;
      hipLaunchKernelGGL(gz_segment_decode, dim3(1), dim3(64 * kWavesPerWg), kWavesPerWg * kGzWaveLdsBytes, sd, (const uint8_t*)nullptr, 0ull, (const GzSeg*)nullptr, 0u,
                         (uint16_t*)nullptr, (GzSegOut*)nullptr, 0u);
      (void)hipGetLastError();
      trace("gzip engine: gz_segment_decode's empty first launch returned (device scratch set up)");
    });
  }
  // ---- the pinned ring the compressed bytes cross in (two pieces: one is filled while the other crosses PCIe).  It belongs to the ENGINE and
  // its pieces grow with the file (gz_ring_piece: 16 MiB up to 1 GiB compressed, 64 MiB beyond that in a context's first session, 128 MiB
  // from its second on) — and pinning 2 x 64 MiB costs a process 22 - 34 ms.  Round 4 pinned it on a helper thread under the device
  // buffers' allocation and then WAITED for it in front of the first copy: 22 of the 50 ms between "context up" and the first byte
  // moving (profiles/r05/cold_marks_before.txt).  Now a process's first call starts on the context's small ring (2 x 16 MiB, 3 - 5 ms) and
  // the copier changes over to the big one, between two pieces, as soon as the helper thread has pinned it.
  const uint64_t want_piece = gz_ring_piece(comp, c->n_sessions <= 1);
  for (int b = 0; b < 2; ++b) if (!g.ev_ring[b]) HIPCHK(hipEventCreateWithFlags(&g.ev_ring[b], hipEventDisableTiming));
  struct RingMaker {
    GzDevBuffers& g;
    std::thread th;
    std::atomic<int> state{0};           // 0: none asked for / being pinned, 1: ready, 2: refused
    uint8_t* ring = nullptr;
    uint64_t piece = 0;
    void join() { if (th.joinable()) th.join(); }
    // (on every way out: the thread joined, and a ring that was pinned belongs to the engine from then on — the copier has been joined
    // by then, nothing reads the fields any more)
    ~RingMaker() { join(); if (ring && state.load() == 1 && g.h_ring != ring) { if (!g.h_ring) { g.h_ring = ring; g.ring_piece = piece; } else (void)hipHostFree(ring); } }
  } ring_maker{g};
  auto start_ring_maker = [&] {
    const int dev = c->dev;
    ring_maker.piece = want_piece;
    RingMaker* rm = &ring_maker;
    ring_maker.th = std::thread([rm, dev] {
      void* p = nullptr;
      if (hipSetDevice(dev) != hipSuccess || hipHostMalloc(&p, 2 * rm->piece, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); rm->state.store(2, std::memory_order_release); return; }
      rm->ring = static_cast<uint8_t*>(p);
      trace("gzip engine: the big pinned ring is there");
      rm->state.store(1, std::memory_order_release);
    });
  };
  // (an engine whose ring is too small for this file — a long-running host that met a bigger file — waits for the new one below, as round 4
  // did: pinned under the device buffers' allocation.  A process's FIRST ring is pinned only once those allocations are through — the
  // runtime takes the two one after the other anyway: side by side, both were done 40 ms later than either alone)
  static const bool host_writes_early = env_int("SCFQ_GZ_DEVICE_HOST_WRITES", 1) != 0;      // (no ring then: see "SCFQ_GZ_DEVICE_HOST_WRITES" below)
  if (g.h_ring && g.ring_piece < want_piece && !host_writes_early) start_ring_maker();
  const uint32_t spare = 64 + batch_segs / 16;         // gap segments of later rounds
  const uint32_t max_seg = batch_segs + spare;
  const uint64_t kSegSlack = 98304;       // (zlib's level-6 blocks inflate to ~60 KB, memLevel 9's to ~130 KB: those overflow now and then and are decoded again)
  auto seg_cap = [&](uint64_t start_bit, uint64_t stop_bit, double mult) {
    const double c = (double)((stop_bit - start_bit + 7) / 8) * ratio_est * mult + (double)kSegSlack;
    return (uint64_t)std::min(c, (double)0x3F000000u) & ~7ull;
  };
  // Bounds (not allocations): a batch may inflate to 40 bytes per compressed byte — one that needs more (long runs: better served by
  // the host's decoder anyway) makes the rest of the file the host's.  What is ALLOCATED follows the need, batch by batch.
  const uint64_t kMaxRatio = 40;
  const uint64_t res_out = std::min<uint64_t>(160ull << 30, batch_comp_max * kMaxRatio + (uint64_t)max_seg * kSegSlack) + 2 * kStagePad;
  // slot meta (device and pinned mirror, same offsets): segs | outs; the searches have their own ring of four (from | found): the
  // search of batch k + 1 is in flight while batch k - 1, of the same parity, is being cut into segments from ITS search's results
  const uint64_t off_segs = 0, off_outs = off_segs + sizeof(GzSeg) * max_seg, slot_meta = off_outs + sizeof(GzSegOut) * max_seg + 64;
  const uint64_t off_from = 0, off_found = off_from + 8ull * max_seg, search_meta = (off_found + 8ull * max_seg + 255) & ~255ull;
  // post meta (per parity): chain | first | work entry | work tile | gap from | gap found | group chain | group first | chain's first group
  static const uint32_t group = (uint32_t)std::min(64, std::max(2, env_int("SCFQ_GZ_DEVICE_CHAIN_GROUP", 64)));      // chain entries per window map
  const uint64_t max_work = std::min<uint64_t>(res_out, batch_comp_max * 16 + (uint64_t)max_seg * kSegSlack) / kResolveTile + max_seg + 8;
  const uint64_t max_groups = max_seg / group + max_seg + 2;               // (every member chain of a batch ends with a short group; a member has a segment)
  const uint64_t offp_chain = 0, offp_first = offp_chain + sizeof(GzChain) * max_seg, offp_we = offp_first + 4ull * (max_seg + 2),
                 offp_wt = offp_we + 4ull * max_work, offp_gfrom = (offp_wt + 4ull * max_work + 7) & ~7ull, offp_gfound = offp_gfrom + 8ull * max_seg,
                 offp_gchain = offp_gfound + 8ull * max_seg, offp_gfirst = offp_gchain + sizeof(GzChain) * max_groups,
                 offp_mfirst = offp_gfirst + 4ull * (max_groups + 2), post_meta = offp_mfirst + 4ull * (max_groups + 2) + 64;
  const uint64_t res_crc = 4 * (16 + (comp * kMaxRatio) / kCrcTile + 2ull * n_plan + 4ull * nb + 8192);      // tiles of the whole file + a part per member and batch
  for (uint32_t b = 0; b < std::min(nb, n_comp); ++b) if ((rc = gz_buf(g, g.comp[b], batch_comp_max + comp_pad + 4096))) return rc;
  if ((rc = gz_buf(g, g.win, (uint64_t)kGzWindow * max_seg)) || (rc = gz_buf(g, g.crc, res_crc))) return rc;
  trace("gzip engine: compressed-byte, window and CRC buffers allocated");
  {
    // the tables of both slots: ONE device and ONE pinned allocation (a pinned allocation costs milliseconds whatever its size)
    const uint32_t ns = std::min(nb, n_slots), nq = std::min(nb, n_comp);
    const uint64_t per = ((slot_meta + post_meta + 255) & ~255ull), crc_room = (res_crc + 255) & ~255ull;
    const uint64_t t0 = g.tables_cap;
    if ((rc = gz_grow(&g.d_tables, &g.tables_cap, per * ns + search_meta * nq)) || (rc = gz_grow(&g.h_tables, &g.htables_cap, per * ns + search_meta * nq + crc_room, true))) return rc;
    for (uint32_t q = 0; q < nq; ++q) { g.d_search[q] = g.d_tables + per * ns + search_meta * q; g.h_search[q] = g.h_tables + per * ns + search_meta * q; }
    note_dev_bytes((int64_t)g.tables_cap - (int64_t)t0);
    for (uint32_t b = 0; b < ns; ++b) {
      g.slot[b].d_meta = g.d_tables + per * b;   g.slot[b].h_meta = g.h_tables + per * b;
      g.d_pmeta[b] = g.slot[b].d_meta + slot_meta; g.h_pmeta[b] = g.slot[b].h_meta + slot_meta;
    }
    g.h_crc = g.h_tables + per * ns + search_meta * nq;
  }
  {
    const uint64_t w0 = g.wcarry_cap;
    if ((rc = gz_grow(&g.d_wcarry, &g.wcarry_cap, 2ull * kGzWindow))) return rc;
    note_dev_bytes((int64_t)g.wcarry_cap - (int64_t)w0);
  }
  uint8_t* d_out = nullptr;                            // (set when the first batch's bytes get their room)
  trace("gzip engine: plan made, tables and compressed-byte buffers allocated");
  const double alloc_ms = std::chrono::duration<double, std::milli>(clk::now() - t_begin).count();
  if (verbose) std::fprintf(stderr, "scfq gzdev: plan + tables in %.1f ms: %u batch(es), segments of %llu KiB, %.2f symbols per compressed byte assumed, literal classes the search rules out 0x%02x\n", alloc_ms, nb,
                            (unsigned long long)(seg_bytes >> 10), ratio_est, lit_mask);
  // the ring in use to begin with: the engine's own when it is big enough; else the context's small one, the copier changing over when the
  // engine's has been pinned (r3: with 16 / 64 / 128 MiB pieces the orchestrating thread of a WARM call waited 94 / 75 / 39 ms for the
  // copier, profiles/r03/gz_device_variants.txt)
  struct Ring { uint8_t* half[2] = {nullptr, nullptr}; uint64_t piece = 0; hipEvent_t ev[2] = {nullptr, nullptr}; uint32_t it = 0; };
  Ring ring_start, ring_big;
  if (host_writes) {
    // (no ring at all)
  } else if (g.ring_piece >= want_piece) {
    ring_start.half[0] = g.h_ring; ring_start.half[1] = g.h_ring + g.ring_piece; ring_start.piece = g.ring_piece; ring_start.ev[0] = g.ev_ring[0]; ring_start.ev[1] = g.ev_ring[1];
  } else {
    if (g.h_ring) {
      // (an engine whose ring is too small for this file — a long-running host that met a bigger file: wait for the new one, as round 4 did)
      ring_maker.join();
      if (ring_maker.state.load() == 1) { (void)hipHostFree(g.h_ring); g.h_ring = ring_maker.ring; g.ring_piece = want_piece; }
      ring_start.half[0] = g.h_ring; ring_start.half[1] = g.h_ring + g.ring_piece; ring_start.piece = g.ring_piece; ring_start.ev[0] = g.ev_ring[0]; ring_start.ev[1] = g.ev_ring[1];
    } else {
      if ((rc = ensure_staging(c, std::max<uint64_t>(c->stage_cap, 16ull << 20), true))) return rc;
      if (want_piece > c->stage_cap) start_ring_maker();      // (a file whose pieces are no bigger than the context's ring stays on that ring)
      ring_start.half[0] = c->h_pin[0]; ring_start.half[1] = c->h_pin[1]; ring_start.piece = std::min<uint64_t>(c->stage_cap, want_piece); ring_start.ev[0] = c->ev_copied[0]; ring_start.ev[1] = c->ev_copied[1];
    }
  }
  trace("gzip engine: pinned ring ready");

  // ---- measurement aid (SCFQ_VERBOSE): device time of the stages, summed over the batches ---------------------------------------
  struct Span { hipEvent_t a = nullptr, b = nullptr; };
  std::vector<Span> sp_copy, sp_search, sp_decode, sp_chain, sp_resolve, sp_crc, sp_scan;
  struct SpanFree {
    std::vector<std::vector<Span>*> all;
    ~SpanFree() { for (auto* v : all) for (Span& s : *v) { if (s.a) (void)hipEventDestroy(s.a); if (s.b) (void)hipEventDestroy(s.b); } }
  } span_free{{&sp_copy, &sp_search, &sp_decode, &sp_chain, &sp_resolve, &sp_crc, &sp_scan}};
  auto span_begin = [&](std::vector<Span>& v, hipStream_t st) { if (!verbose) return; Span s; (void)hipEventCreate(&s.a); (void)hipEventCreate(&s.b); (void)hipEventRecord(s.a, st); v.push_back(s); };
  auto span_end = [&](std::vector<Span>& v, hipStream_t st) { if (!verbose) return; (void)hipEventRecord(v.back().b, st); };

  // ---- state that travels from batch to batch ------------------------------------------------------------------------------
  uint64_t pos = first_bit;              // the exact bit the chain has reached
  bool finished = false;                 // the last member's final block has been walked and no further member follows
  uint64_t end_byte = 0;                 // ... and where that was: the offset just behind its trailer
  uint32_t member_no = 0;                // running member that `pos` lies in
  uint32_t valid = sx ? (map_only ? (uint32_t)kGzWindow : sx->valid) : 0;      // bytes of that member's history in front of pos (<= 32768; pass 1 of a stretch: taken as full, the map tells)
  int wcarry = 0;                        // which of the two carried windows is the current one
  int stretch_first_byte = -1;
  uint64_t total_out = 0;
  bool have_prev_out = false;
  uint64_t prev_out_bytes = 0;           // size of the previous batch's output (its last byte is the next batch's look-behind)
  uint64_t tiles_used = 0;
  std::vector<GzPart> parts;
  std::vector<GzMemberEnd> member_ends;
  uint32_t n_planned_total = 0, n_decoded_total = 0, n_chain_total = 0, n_gap_rounds = 0, n_overflows = 0;
  const char* why = nullptr;             // why a batch was handed to the host (kFallbackRest)
  std::map<uint64_t, double> overflow_mult;      // start bit of a segment that ran out of room -> the multiple of the assumed room it was last given
  double fill_ms = 0, walk_ms = 0, h_copier_wait_ms = 0, h_search_wait_ms = 0, h_dec_wait_ms = 0, h_post_wait_ms = 0;
  Ring* ring = &ring_start;              // (the copier thread's: the ring in use, its pieces alternating over the whole file)
  std::vector<uint32_t> n_seg_of(nb + 1, 0);
  std::vector<uint64_t> pool_used_of(nb + 1, 0);
  HIPCHK(hipMemsetAsync(g.crc.p, 0, 64, c->compute));          // word 0: the resolve kernels' error status for the whole file
  std::vector<uint16_t> identity_map;
  if (map_only) {
    if ((rc = gz_buf(g, g.gwin, 4ull * kGzWindow))) return rc;      // the stretch's running map and its double (2 x 32768 symbols)
    identity_map.resize(kGzWindow);
    for (uint32_t i = 0; i < kGzWindow; ++i) identity_map[i] = (uint16_t)(0x8000u | i);
    HIPCHK(hipMemcpyAsync(g.gwin.p, identity_map.data(), 2ull * kGzWindow, hipMemcpyHostToDevice, c->compute));
    HIPCHK(hipStreamSynchronize(c->compute));      // (pageable source)
  } else if (sx && sx->valid) {
    // pass 2 of a stretch: the window in front of it is the carried window of "batch -1"
    HIPCHK(hipMemcpyAsync(g.d_wcarry, sx->window, kGzWindow, hipMemcpyHostToDevice, c->compute));
    HIPCHK(hipStreamSynchronize(c->compute));
  }

  // ---- the COPIER: a thread of its own moves the compressed bytes — file mapping -> pinned ring -> comp[k % 4] — batch after batch,
  // up to four batches ahead of the walk.  On the orchestrating thread (round 2, and this round until the last day) the copies
  // were 100 of the 166 ms that thread was busy for a 10 GB file, and the thread was the pipeline's bottleneck: it waited 25 ms
  // for decodes, the decodes waited for it.  ev_copy[k % 4] is recorded behind a batch's last piece; `enq` counts the batches whose
  // copies are queued, `allowed` the batches whose buffer is free again (batch k + 4 re-uses batch k's once that has been walked).
  struct Copier {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    uint32_t enq = 0, allowed = 4;      // (allowed: set to the number of compressed-byte buffers before the thread starts)
    bool stop = false;
    int rc = SCFQ_OK;
    std::string err;
    double evsync_ms = 0, memcpy_ms = 0, enqueue_ms = 0, fill_ms = 0;
    uint64_t bytes = 0;
    std::vector<Span> spans;
    ~Copier() {
      { std::lock_guard<std::mutex> lk(mu); stop = true; }
      cv.notify_all();
      if (th.joinable()) th.join();
    }
  } cp;
  // (a second copy stream for the second half of every piece: measured, see DESIGN.md §5 "(r4) The warm path")
  static const int copy_streams = env_int("SCFQ_GZ_DEVICE_COPY_STREAMS", 1);
  hipStream_t s_copy2 = nullptr;
  hipEvent_t ev_copy2 = nullptr;
  struct Copy2Free { hipStream_t& s; hipEvent_t& e; ~Copy2Free() { if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); } if (e) (void)hipEventDestroy(e); } } copy2_free{s_copy2, ev_copy2};
  if (copy_streams >= 2 && nb > 1) {
    HIPCHK(hipStreamCreateWithFlags(&s_copy2, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&ev_copy2, hipEventDisableTiming));
  }
  uint32_t fg_it = 0;                                // (the copier thread's: staging pieces written so far in this call)
  auto copy_batch = [&](uint32_t k) -> int {       // (runs on the copier thread)
    const auto tf = clk::now();
    const int cb = (int)(k % n_comp);
    const uint64_t b0 = byte0_of(k), b1 = copy_end_of(k);
    if (verbose) { Span sp; (void)hipEventCreate(&sp.a); (void)hipEventCreate(&sp.b); (void)hipEventRecord(sp.a, c->copy); cp.spans.push_back(sp); }
    if (host_writes) {
      for (uint64_t off = b0; off < b1; off += fg_piece, ++fg_it) {
        const int sb = (int)(fg_it & 1u);
        const uint64_t len = std::min<uint64_t>(fg_piece, b1 - off);
        const bool last = off + len == b1;
        auto t0 = clk::now();
        if (fg_it >= 2) HIPCHK(hipEventSynchronize(g.ev_fg[sb]));          // the device-to-device copy of the piece before last has read this buffer
        auto t1 = clk::now();
        if (k == 0 && fg_it < n_prefilled) {
          if (prefill.th.joinable()) prefill.th.join();                    // (read while the streams and buffers were being made)
        } else {
          copy_file_bytes(fbytes, off, g.fg_stage[sb], len, true);
          if (last) std::memset(g.fg_stage[sb] + len, 0, comp_pad);
        }
        std::atomic_thread_fence(std::memory_order_seq_cst);
        auto t2 = clk::now();
        HIPCHK(hipMemcpyAsync(g.comp[cb].p + (off - b0), g.fg_stage[sb], (size_t)(len + (last ? comp_pad : 0)), hipMemcpyDeviceToDevice, c->copy));
        HIPCHK(hipEventRecord(g.ev_fg[sb], c->copy));
        cp.evsync_ms += std::chrono::duration<double, std::milli>(t1 - t0).count();
        cp.memcpy_ms += std::chrono::duration<double, std::milli>(t2 - t1).count();
      }
      if (verbose) (void)hipEventRecord(cp.spans.back().b, c->copy);
      HIPCHK(hipEventRecord(g.ev_copy[cb], c->copy));
      cp.bytes += b1 - b0;
      if (k == 0) trace("gzip engine: first batch's compressed bytes written to the device");
      cp.fill_ms += std::chrono::duration<double, std::milli>(clk::now() - tf).count();
      return SCFQ_OK;
    }
    for (uint64_t off = b0, len = 0; off < b1; off += len) {
      if (ring == &ring_start && !g.h_ring && ring_maker.state.load(std::memory_order_acquire) == 1) {
        // the big ring is there: from this piece on (what is in flight out of the small one finishes by itself; nobody frees that ring)
        ring_big.half[0] = ring_maker.ring; ring_big.half[1] = ring_maker.ring + want_piece; ring_big.piece = want_piece; ring_big.ev[0] = g.ev_ring[0]; ring_big.ev[1] = g.ev_ring[1];
        ring = &ring_big;
      }
      const int pb = (int)(ring->it & 1);
      len = std::min(ring->piece, b1 - off);
      auto t0 = clk::now();
      if (ring->it >= 2) HIPCHK(hipEventSynchronize(ring->ev[pb]));
      auto t1 = clk::now();
      copy_file_bytes(fbytes, off, ring->half[pb], len);
      auto t2 = clk::now();
      if (s_copy2 && len >= (8ull << 20)) {
        // two halves on two streams: two copy engines share the link (SCFQ_GZ_DEVICE_COPY_STREAMS=2, files of several batches)
        const uint64_t h1 = (len / 2 + 4095) & ~4095ull;
        HIPCHK(hipMemcpyAsync(g.comp[cb].p + (off - b0), ring->half[pb], (size_t)h1, hipMemcpyHostToDevice, c->copy));
        HIPCHK(hipMemcpyAsync(g.comp[cb].p + (off - b0) + h1, ring->half[pb] + h1, (size_t)(len - h1), hipMemcpyHostToDevice, s_copy2));
        HIPCHK(hipEventRecord(ev_copy2, s_copy2));
        HIPCHK(hipStreamWaitEvent(c->copy, ev_copy2, 0));      // (everything behind this on the copy stream — and every event recorded there — covers both halves)
      } else {
        HIPCHK(hipMemcpyAsync(g.comp[cb].p + (off - b0), ring->half[pb], (size_t)len, hipMemcpyHostToDevice, c->copy));
      }
      ++ring->it;
      HIPCHK(hipEventRecord(ring->ev[pb], c->copy));
      auto t3 = clk::now();
      cp.evsync_ms += std::chrono::duration<double, std::milli>(t1 - t0).count();
      cp.memcpy_ms += std::chrono::duration<double, std::milli>(t2 - t1).count();
      cp.enqueue_ms += std::chrono::duration<double, std::milli>(t3 - t2).count();
    }
    HIPCHK(hipMemsetAsync(g.comp[cb].p + (b1 - b0), 0, comp_pad, c->copy));
    if (verbose) (void)hipEventRecord(cp.spans.back().b, c->copy);
    HIPCHK(hipEventRecord(g.ev_copy[cb], c->copy));
    cp.bytes += b1 - b0;
    if (k == 0) trace("gzip engine: first batch's compressed bytes queued for the device");
    cp.fill_ms += std::chrono::duration<double, std::milli>(clk::now() - tf).count();
    return SCFQ_OK;
  };
  cp.allowed = n_comp;
  cp.th = std::thread([&] {
    if (hipSetDevice(c->dev) != hipSuccess) { std::lock_guard<std::mutex> lk(cp.mu); cp.rc = SCFQ_EHIP; cp.err = "hipSetDevice on the copier thread"; cp.cv.notify_all(); return; }
    for (uint32_t k = 0; k < nb; ++k) {
      {
        std::unique_lock<std::mutex> lk(cp.mu);
        cp.cv.wait(lk, [&] { return cp.stop || k < cp.allowed; });
        if (cp.stop) return;
      }
      const int r = copy_batch(k);
      {
        std::lock_guard<std::mutex> lk(cp.mu);
        if (r) { cp.rc = r; cp.err = g_err; }
        cp.enq = k + 1;
      }
      cp.cv.notify_all();
      if (r) return;
    }
  });
  struct CopierJoin {      // (declared behind everything the copier thread touches: runs first on every way out of this function)
    Copier& cp;
    void now() { { std::lock_guard<std::mutex> lk(cp.mu); cp.stop = true; } cp.cv.notify_all(); if (cp.th.joinable()) cp.th.join(); }
    ~CopierJoin() { now(); }
  } copier_join{cp};
  // the walk of batch k is through: batch k + n_comp may take its buffer
  auto release_comp = [&](uint32_t k) { { std::lock_guard<std::mutex> lk(cp.mu); cp.allowed = std::max(cp.allowed, k + n_comp + 1); } cp.cv.notify_all(); };

  // ---- stage A(k): the block-start search of batch k, queued behind its copy — launched one iteration before its results are waited for ----
  auto stage_a = [&](uint32_t k) -> int {
    const int cb = (int)(k % n_comp);
    const uint64_t b0 = byte0_of(k), b1 = copy_end_of(k);
    {
      auto t0 = clk::now();
      std::unique_lock<std::mutex> lk(cp.mu);
      cp.cv.wait(lk, [&] { return cp.enq > k || cp.rc != SCFQ_OK; });
      h_copier_wait_ms += std::chrono::duration<double, std::milli>(clk::now() - t0).count();
      if (cp.rc != SCFQ_OK) { std::snprintf(g_err, sizeof g_err, "%s", cp.err.c_str()); return cp.rc; }
    }
    // search: the planned segments of this batch (the file's very first one starts exactly at the member's first block)
    uint64_t* h_from = reinterpret_cast<uint64_t*>(g.h_search[cb] + off_from);
    uint64_t* h_found = reinterpret_cast<uint64_t*>(g.h_search[cb] + off_found);
    const uint64_t p0 = p0_of(k);
    const uint32_t np = (uint32_t)(p1_of(k) - p0);
    for (uint32_t s = 0; s < np; ++s) { h_from[s] = (data0 + (p0 + s) * seg_bytes) * 8; h_found[s] = ~0ull; }
    const uint32_t s0 = (k == 0) ? 1u : 0u;
    if (k == 0) h_found[0] = first_bit;
    HIPCHK(hipStreamWaitEvent(s_search, g.ev_copy[cb], 0));
    if (np > s0) {
      const uint8_t* vbase = g.comp[cb].p - b0;          // virtual base: byte i of the file is vbase[i] for i in [b0, b1 + pad)
      span_begin(sp_search, s_search);
      HIPCHK(hipMemcpyAsync(g.d_search[cb] + off_from, h_from, 8ull * np, hipMemcpyHostToDevice, s_search));
      hipLaunchKernelGGL(gz_sync_search, dim3(np - s0), dim3(kSyncThreads), 0, s_search, reinterpret_cast<const uint64_t*>(vbase), b1 * 8,
                         reinterpret_cast<const uint64_t*>(g.d_search[cb] + off_from) + s0, np - s0, seg_bytes * 8,
                         reinterpret_cast<uint64_t*>(g.d_search[cb] + off_found) + s0, lit_mask);
      HIPCHK(hipGetLastError());
      HIPCHK(hipMemcpyAsync(h_found + s0, g.d_search[cb] + off_found + 8ull * s0, 8ull * (np - s0), hipMemcpyDeviceToHost, s_search));
      span_end(sp_search, s_search);
      if (k == 0) trace("gzip engine: first search kernel queued");
    }
    HIPCHK(hipEventRecord(g.ev_found[cb], s_search));
    n_planned_total += np;
    return SCFQ_OK;
  };
  // ---- stage B(k): the segments of batch k (its last one stops at the first start of batch k + 1), decode ---------------------
  auto stage_b = [&](uint32_t k) -> int {
    GzSlot& sl = g.slot[k % n_slots];
    const uint64_t* h_found = reinterpret_cast<const uint64_t*>(g.h_search[k % n_comp] + off_found);
    GzSeg* h_segs = reinterpret_cast<GzSeg*>(sl.h_meta + off_segs);
    const uint32_t np = (uint32_t)(p1_of(k) - p0_of(k));
    const uint64_t limit = copy_end_of(k) * 8;
    const uint64_t next_first = (k + 1 < nb) ? byte1_of(k) * 8 : end_bit;     // where the last segment stops
    // (always at the batch's border: the wave runs on to the end of the block that crosses it, which is where the next batch's first
    // segment starts when the search found that block start; the walk checks it and fills a gap otherwise.  The decode of batch k
    // then does not wait for the copy and the search of batch k + 1.)
    uint32_t n_seg = 0;
    uint64_t last_start = 0, pool_used = 0;
    for (uint32_t s = 0; s < np; ++s) {
      const uint64_t st = h_found[s];
      if (st == ~0ull || (n_seg && st <= last_start) || st + 64 >= limit || st >= next_first) continue;     // none found: the segment before runs on
      if (n_seg) h_segs[n_seg - 1].stop_bit = st;
      h_segs[n_seg].start_bit = st;
      h_segs[n_seg].stop_bit = next_first;
      last_start = st;
      ++n_seg;
    }
    // Longest first (r4; SCFQ_GZ_DEVICE_SORT_SEGMENTS=0: file order).  A workgroup holds its LDS until its slowest wave is through, and the
    // LDS is what limits the decode kernel's occupancy: with the segments of a batch in order of length the workgroups are of one kind
    // each, the short ones end early and their room goes to the next batch's workgroups, which are queued on the other decode stream.
    // Measured on the 10 GB member, three alternating pairs on one box (profiles/r04/gz_sort_segments_ab.jsonl): 111 - 112 ms against
    // 120 - 123, decode kernels 78 ms against 88.  The walk finds segments by their start bit: their order in the table is free.
    static const int sort_segs = env_int("SCFQ_GZ_DEVICE_SORT_SEGMENTS", 1);
    if (sort_segs && n_seg > 1)
      std::sort(h_segs, h_segs + n_seg, [](const GzSeg& a, const GzSeg& b) { return a.stop_bit - a.start_bit > b.stop_bit - b.start_bit; });
    sl.sym.rewind();
    for (uint32_t q = 0; q < n_seg; ++q) {
      const uint64_t cap = seg_cap(h_segs[q].start_bit, h_segs[q].stop_bit, 1.0);
      if (gz_take(sl.sym, kGzWindow + cap, &h_segs[q].sym_off)) { why = "no device memory for a batch's symbols"; return gz_decline(__LINE__, kFallbackRest); }      // (the batches that are through stay)
      h_segs[q].cap = (uint32_t)cap;     // better compression than assumed: overflow status, and the walk has that segment decoded again
      h_segs[q].reserved = 0;
      pool_used += kGzWindow + cap;
    }
    n_seg_of[k] = n_seg;
    pool_used_of[k] = pool_used;
    hipStream_t sd = s_dec[k % kGzDecodeStreams];
    if (warm.th.joinable()) warm.th.join();
    if (k >= n_slots) HIPCHK(hipStreamWaitEvent(sd, g.ev_post[k % n_slots], 0));      // the slot's symbols were read by the resolve of the batch that had it before
    HIPCHK(hipStreamWaitEvent(sd, g.ev_copy[k % n_comp], 0));
    if (n_seg) {
      const uint8_t* vbase = g.comp[k % n_comp].p - byte0_of(k);
      span_begin(sp_decode, sd);
      HIPCHK(hipMemcpyAsync(sl.d_meta + off_segs, h_segs, sizeof(GzSeg) * n_seg, hipMemcpyHostToDevice, sd));
      hipLaunchKernelGGL(gz_segment_decode, dim3((n_seg + kWavesPerWg - 1) / kWavesPerWg), dim3(64 * kWavesPerWg), kWavesPerWg * kGzWaveLdsBytes, sd,
                         vbase, copy_end_of(k), reinterpret_cast<const GzSeg*>(sl.d_meta + off_segs), n_seg, sl.sym.base(),
                         reinterpret_cast<GzSegOut*>(sl.d_meta + off_outs), inflate_serial_loop());
      HIPCHK(hipGetLastError());
      HIPCHK(hipMemcpyAsync(sl.h_meta + off_outs, sl.d_meta + off_outs, sizeof(GzSegOut) * n_seg, hipMemcpyDeviceToHost, sd));
      span_end(sp_decode, sd);
    }
    HIPCHK(hipEventRecord(g.ev_dec[k % n_slots], sd));
    if (k == 0) trace("gzip engine: first decode kernel queued");
    n_decoded_total += n_seg;
    return SCFQ_OK;
  };

  // ---- a stretch in ONE pass (GzStretch::exchange): what the first half of every batch leaves for later -----------------------------
  struct PartHere { uint32_t member; uint64_t off, len; };
  struct BatchKept {
    uint32_t k = 0, n_chain = 0, n_chains = 0;
    uint64_t n_work = 0, batch_out = 0;
    std::vector<PartHere> parts_here;
    std::vector<GzChain> chain;                  // sym_off: into the store of kept symbols
    std::vector<uint32_t> first, we, wt;
  };
  std::vector<BatchKept> kept;
  SymPool keep_store;                            // the proven chain entries' output symbols, packed: 2 bytes per inflated byte of the stretch
  struct KeepFree { SymPool& p; ~KeepFree() { note_dev_bytes(-(int64_t)p.bytes()); p.release(); } } keep_free{keep_store};
  DevBuf pack_tab;                               // gz_pack_symbols' tables, a slice per batch
  struct PackFree { DevBuf& b; ~PackFree() { b.release(); } } pack_free{pack_tab};
  if (keep) {
    const hipError_t e = hipMalloc(reinterpret_cast<void**>(&pack_tab.p), 24ull * max_seg * nb);
    if (e != hipSuccess) { (void)hipGetLastError(); pack_tab.p = nullptr; return SCFQ_GZ_DECLINE; }
    pack_tab.cap = 24ull * max_seg * nb;
  }

  // ---- the second half of a batch: its part of the chain becomes windows, bytes, CRC tiles and a scan.  The chain's tables are in
  // g.h_pmeta[pp] (chain | first | work entry | work tile); d_sym: where the entries' sym_off count from.  carry_in / carry_out: the
  // first chain goes on inside a member begun earlier / the last chain's member goes on in the next batch.
  auto post_process = [&](uint32_t k, int pp, uint16_t* d_sym, uint32_t n_chain, uint32_t n_chains, uint64_t n_work, uint64_t batch_out,
                          const std::vector<PartHere>& parts_here, bool carry_in, bool carry_out) -> int {
    const uint32_t* h_first = reinterpret_cast<const uint32_t*>(g.h_pmeta[pp] + offp_first);
    // room for what this batch turned out to need.  A bigger buffer than the last batch's is a NEW buffer (the old one, which the
    // last batch's scan may still be reading, is freed when the call ends): that batch's last byte — this batch's look-behind —
    // moves over with a one-byte copy
    uint8_t* const old_out = d_out;
    bool parked = false;
    if (int r = gz_buf(g, g.out, std::max<uint64_t>(batch_out, (uint64_t)((double)(copy_end_of(k) - byte0_of(k)) * ratio_est * 0.8)) + 2 * kStagePad)) return r;
    d_out = g.out.p + kStagePad;
    if (have_prev_out && old_out != d_out) {
      HIPCHK(hipMemcpyAsync(d_out - 1, old_out + prev_out_bytes - 1, 1, hipMemcpyDeviceToDevice, c->compute));
      parked = true;
    }
        uint8_t* const d_win = g.win.p;
    // the byte in front of this batch's output is the last byte of the batch before it: parked below the buffer before that is overwritten
    if (have_prev_out && !parked) HIPCHK(hipMemcpyAsync(d_out - 1, d_out + prev_out_bytes - 1, 1, hipMemcpyDeviceToDevice, c->compute));
    HIPCHK(hipMemcpyAsync(g.d_pmeta[pp] + offp_chain, g.h_pmeta[pp] + offp_chain, sizeof(GzChain) * n_chain, hipMemcpyHostToDevice, c->compute));
    HIPCHK(hipMemcpyAsync(g.d_pmeta[pp] + offp_first, g.h_pmeta[pp] + offp_first, 4ull * (n_chains + 1), hipMemcpyHostToDevice, c->compute));
    if (n_work) {
      HIPCHK(hipMemcpyAsync(g.d_pmeta[pp] + offp_we, g.h_pmeta[pp] + offp_we, 4ull * n_work, hipMemcpyHostToDevice, c->compute));
      HIPCHK(hipMemcpyAsync(g.d_pmeta[pp] + offp_wt, g.h_pmeta[pp] + offp_wt, 4ull * n_work, hipMemcpyHostToDevice, c->compute));
    }
    const uint8_t* w_in = carry_in ? g.d_wcarry + (uint64_t)wcarry * kGzWindow : nullptr;
    uint8_t* w_out = carry_out ? g.d_wcarry + (uint64_t)(wcarry ^ 1) * kGzWindow : nullptr;
    const GzChain* d_chain = reinterpret_cast<const GzChain*>(g.d_pmeta[pp] + offp_chain);
    span_begin(sp_chain, c->compute);
    if (n_chain <= n_chains + group) {
      // short chains: one walk per member
      hipLaunchKernelGGL(gz_window_chain, dim3(n_chains), dim3(1024), 0, c->compute, d_chain, reinterpret_cast<const uint32_t*>(g.d_pmeta[pp] + offp_first),
                         d_sym, d_win, w_in, w_out, (const uint8_t*)nullptr);
    } else {
      // groups of `group` entries inside every member's chain: maps, windows in front of the groups, windows in front of the entries
      GzChain* h_gchain = reinterpret_cast<GzChain*>(g.h_pmeta[pp] + offp_gchain);
      uint32_t* h_gfirst = reinterpret_cast<uint32_t*>(g.h_pmeta[pp] + offp_gfirst);
      uint32_t* h_mfirst = reinterpret_cast<uint32_t*>(g.h_pmeta[pp] + offp_mfirst);
      uint32_t n_groups = 0;
      for (uint32_t ch = 0; ch < n_chains; ++ch) {
        h_mfirst[ch] = n_groups;
        for (uint32_t q = h_first[ch]; q < h_first[ch + 1]; q += group) {
          if (n_groups >= max_groups) return SCFQ_GZ_DECLINE;      // (cannot happen: max_groups covers a group per member and per 64 entries)
          h_gfirst[n_groups] = q;
          h_gchain[n_groups] = GzChain{};
          h_gchain[n_groups].sym_off = (uint64_t)n_groups * kGzWindow;       // (map g lies one window further: the form of a segment's symbols)
          h_gchain[n_groups].n_sym = kGzWindow;
          ++n_groups;
        }
      }
      h_mfirst[n_chains] = n_groups;
      h_gfirst[n_groups] = n_chain;
      // (a bigger map buffer than the last batch's is a new one: that batch's window kernels are behind this batch's on the same stream)
      if (int r = gz_buf(g, g.maps, 2ull * kGzWindow * (n_groups + 1))) return r;
      if (int r = gz_buf(g, g.gwin, (uint64_t)kGzWindow * n_groups)) return r;
      HIPCHK(hipMemcpyAsync(g.d_pmeta[pp] + offp_gchain, h_gchain, sizeof(GzChain) * n_groups, hipMemcpyHostToDevice, c->compute));
      HIPCHK(hipMemcpyAsync(g.d_pmeta[pp] + offp_gfirst, h_gfirst, 4ull * (n_groups + 1), hipMemcpyHostToDevice, c->compute));
      HIPCHK(hipMemcpyAsync(g.d_pmeta[pp] + offp_mfirst, h_mfirst, 4ull * (n_chains + 1), hipMemcpyHostToDevice, c->compute));
      const uint32_t* d_gfirst = reinterpret_cast<const uint32_t*>(g.d_pmeta[pp] + offp_gfirst);
      hipLaunchKernelGGL(gz_window_maps, dim3(n_groups), dim3(1024), 0, c->compute, d_chain, d_gfirst, d_sym, reinterpret_cast<uint16_t*>(g.maps.p));
      hipLaunchKernelGGL(gz_window_chain, dim3(n_chains), dim3(1024), 0, c->compute, reinterpret_cast<const GzChain*>(g.d_pmeta[pp] + offp_gchain),
                         reinterpret_cast<const uint32_t*>(g.d_pmeta[pp] + offp_mfirst), reinterpret_cast<const uint16_t*>(g.maps.p), g.gwin.p, w_in, w_out,
                         (const uint8_t*)nullptr);
      hipLaunchKernelGGL(gz_window_chain, dim3(n_groups), dim3(1024), 0, c->compute, d_chain, d_gfirst, d_sym, d_win, (const uint8_t*)nullptr,
                         (uint8_t*)nullptr, (const uint8_t*)g.gwin.p);
    }
    HIPCHK(hipGetLastError());
    span_end(sp_chain, c->compute);
    if (carry_out) wcarry ^= 1;
    if (n_work) {
      span_begin(sp_resolve, c->compute);
      // (r5: the same kernel reading the window where it lies instead of from LDS — it could start on CUs whose LDS the decode's workgroups
      // hold — made the call 2 - 3 ms longer, CUs kept free of the decode 18 ms: profiles/r05/gz_post_starvation_ab.txt)
      hipLaunchKernelGGL(gz_resolve, dim3((unsigned)n_work), dim3(256), 0, c->compute, reinterpret_cast<const GzChain*>(g.d_pmeta[pp] + offp_chain),
                         reinterpret_cast<const uint32_t*>(g.d_pmeta[pp] + offp_we), reinterpret_cast<const uint32_t*>(g.d_pmeta[pp] + offp_wt), d_sym,
                         d_win, d_out, reinterpret_cast<uint32_t*>(g.crc.p));
      HIPCHK(hipGetLastError());
      span_end(sp_resolve, c->compute);
    }
    span_begin(sp_crc, c->compute);
    for (const PartHere& ph : parts_here) {
      const uint64_t nt = (ph.len + kCrcTile - 1) / kCrcTile;
      if (nt) {
        hipLaunchKernelGGL(gz_crc32_tiles, dim3((unsigned)nt), dim3(256), 0, c->compute, d_out + ph.off, ph.len, nt * kCrcTile - ph.len,
                           reinterpret_cast<uint32_t*>(g.crc.p) + 16 + tiles_used);
        HIPCHK(hipGetLastError());
      }
      parts.push_back(GzPart{ph.member, ph.len, tiles_used, nt});
      tiles_used += nt;
    }
    span_end(sp_crc, c->compute);
    if (sx && !have_prev_out && batch_out) {
      // (the byte a shard of a sharded count begins with: what the byte in front of it — known to the rank before — decides at the fold)
      uint8_t fb = 0;
      HIPCHK(hipMemcpyAsync(&fb, d_out, 1, hipMemcpyDeviceToHost, c->compute));
      HIPCHK(hipStreamSynchronize(c->compute));
      stretch_first_byte = fb;
    }
    if (batch_out) {
      span_begin(sp_scan, c->compute);
      rc = scan_async(c, d_out, batch_out, have_prev_out ? -2 : -1, flags & ~SCFQ_PREV_IN_MEMORY, timing);
      if (rc) return rc;
      span_end(sp_scan, c->compute);
      have_prev_out = true;
      prev_out_bytes = batch_out;
    }
      return SCFQ_OK;
  };

  // ---- stage C(k): walk, windows, bytes, CRC tiles, scan -----------------------------------------------------------------------
  auto stage_c = [&](uint32_t k) -> int {
    GzSlot& sl = g.slot[k % n_slots];
    const int pp = (int)(k % n_slots);
    GzSeg* h_segs = reinterpret_cast<GzSeg*>(sl.h_meta + off_segs);
    GzSegOut* h_outs = reinterpret_cast<GzSegOut*>(sl.h_meta + off_outs);
    { auto t0 = clk::now(); HIPCHK(hipEventSynchronize(g.ev_dec[pp])); h_dec_wait_ms += std::chrono::duration<double, std::milli>(clk::now() - t0).count(); }
    if (k >= n_slots) { auto t0 = clk::now(); HIPCHK(hipEventSynchronize(g.ev_post[pp])); h_post_wait_ms += std::chrono::duration<double, std::milli>(clk::now() - t0).count(); }      // h_pmeta[pp] / d_pmeta[pp] were batch k - 2's
    // (SCFQ_GZ_DEVICE_TEST_REST_AT=k: batch k declares itself out of room — the tests of the hand-over to the host's decoder)
    static const int test_rest_at = env_int("SCFQ_GZ_DEVICE_TEST_REST_AT", -1);
    if (test_rest_at >= 0 && k == (uint32_t)test_rest_at) { why = "SCFQ_GZ_DEVICE_TEST_REST_AT"; return gz_decline(__LINE__, kFallbackRest); }
    const auto tw = clk::now();
    uint64_t* h_gfrom = reinterpret_cast<uint64_t*>(g.h_pmeta[pp] + offp_gfrom);
    uint64_t* h_gfound = reinterpret_cast<uint64_t*>(g.h_pmeta[pp] + offp_gfound);
    uint32_t n_seg = n_seg_of[k];
    uint64_t pool_used = pool_used_of[k];
    const uint8_t* vbase = g.comp[k % n_comp].p - byte0_of(k);
    const bool last_batch = k + 1 == nb;
    const uint64_t territory_end = last_batch ? end_bit : byte1_of(k) * 8, limit = copy_end_of(k) * 8;
    struct Entry { uint32_t seg; uint32_t member; };
    std::vector<Entry> chain;
    uint64_t pos_end = pos;
    uint32_t member_end_no = member_no;
    bool finished_end = false;
    uint64_t end_byte_here = 0;
    std::vector<GzMemberEnd> ends_here;
    for (int round = 0;; ++round) {
      std::map<uint64_t, uint32_t> by_start;
      for (uint32_t s = 0; s < n_seg; ++s) {      // the first decode of a start bit wins (later ones are identical) — unless it ran out of room: then the last one does
        auto ins = by_start.emplace(h_segs[s].start_bit, s);
        if (!ins.second && h_outs[ins.first->second].status == kGzErrOverflow) ins.first->second = s;
      }
      chain.clear(); ends_here.clear();
      finished_end = false;
      struct Gap { uint64_t first, second; double mult; };
      std::vector<Gap> gaps;
      bool tentative = false, done = false;
      uint64_t p = pos;
      uint32_t mno = member_no;
      while (!done) {
        if (!last_batch && p >= territory_end) { done = true; break; }          // the next batch carries on from here
        if (stretch_to && p >= end_bit) {
          // the stretch ends at a block boundary the next rank begins at: the chain must arrive there EXACTLY (a boundary that was
          // no block start after all is the caller's to deal with: every rank falls back)
          if (p != end_bit) return SCFQ_GZ_DECLINE;
          finished_end = true;
          done = true;
          break;
        }
        auto it = by_start.find(p);
        if (it == by_start.end()) {
          // nothing was decoded from this exact bit (the first block of a further member, or the segment planned here started
          // at a false sync and the one before ran over it): decode [p, next known start) in the next round
          auto nx = by_start.upper_bound(p);
          uint64_t stop = nx != by_start.end() ? nx->first : (last_batch ? end_bit : std::min(territory_end + 8 * seg_bytes, limit - 4096));
          if (stop <= p) return SCFQ_GZ_DECLINE;
          // (one wave would decode a gap that long for seconds — a run of stored blocks, a block of megabytes: nothing the search
          // looks for starts in it.  Nothing suspect about the data: the host's decoder takes the file from the chain's position,
          // what the batches before folded stays)
          if (stop - p > 64 * 8 * seg_bytes) { why = "a stretch of more than 64 segments without a block start the search accepts"; return gz_decline(__LINE__, kFallbackRest); }
          gaps.push_back(Gap{p, stop, 1.0});
          if (verbose) std::fprintf(stderr, "scfq gzdev:   batch %u: gap at bit %llu (%.1f %% of the file) up to %llu\n", k, (unsigned long long)p,
                                    100.0 * (double)p / (double)end_bit, (unsigned long long)stop);
          tentative = true;
          if (nx == by_start.end()) break;
          p = nx->first;                 // (assume the gap segment arrives exactly there; the next round's walk checks it)
          continue;
        }
        const GzSegOut& r = h_outs[it->second];
        if (r.status == kGzErrOverflow) {
          // this stretch compresses better than the room its segment was given: like a gap, it is decoded again in the next
          // round — cut into pieces, every piece with four times the room per compressed byte (16, 64 times if that is not enough)
          double& m = overflow_mult[p];
          m = m > 0 ? m * 4.0 : 4.0;
          if (m > 300.0) { why = "a segment overflowed 256 times the assumed room"; return gz_decline(__LINE__, kFallbackRest); }
          auto nx = std::next(it);
          const uint64_t stop = nx != by_start.end() ? nx->first : (last_batch ? end_bit : std::min(territory_end + 8 * seg_bytes, limit - 4096));
          if (stop <= p) return SCFQ_GZ_DECLINE;
          by_start.erase(it);
          gaps.push_back(Gap{p, stop, m});
          ++n_overflows;
          if (verbose) std::fprintf(stderr, "scfq gzdev:   batch %u: segment at bit %llu needs more than %.1f symbols per byte: again with %.0f x the room\n", k,
                                    (unsigned long long)p, ratio_est * m / 4.0, m);
          tentative = true;
          if (nx == by_start.end()) break;
          p = nx->first;
          continue;
        }
        if (r.status != kGzOk && r.status != kGzMemberEnd) {
          if (tentative) break;          // may not even be on the real path: decide after the gaps are decoded
          if (verbose) std::fprintf(stderr, "scfq gzdev: segment at bit %llu ended with status %u: host path\n", (unsigned long long)p, r.status);
          return SCFQ_GZ_DECLINE;
        }
        if (r.end_bit <= p) return SCFQ_GZ_DECLINE;
        chain.push_back(Entry{it->second, mno});
        if (r.status == kGzMemberEnd) {
          if (stretch_to) return SCFQ_GZ_DECLINE;              // (a stretch that was to end at a block boundary: the file is not ONE member after all)
          const uint64_t q = (r.end_bit + 7) >> 3;             // the trailer starts on the next byte boundary
          if (q + 8 > fsize) return SCFQ_GZ_DECLINE;           // truncated trailer: gzread's error, from the host path
          GzMemberEnd me;
          me.member = mno;
          me.crc = (uint32_t)img[q] | ((uint32_t)img[q + 1] << 8) | ((uint32_t)img[q + 2] << 16) | ((uint32_t)img[q + 3] << 24);
          me.isize = (uint32_t)img[q + 4] | ((uint32_t)img[q + 5] << 8) | ((uint32_t)img[q + 6] << 16) | ((uint32_t)img[q + 7] << 24);
          ends_here.push_back(me);
          const long h = scfq_gzfast::member_header(img + q + 8, (size_t)(fsize - (q + 8)));
          if (h < 0) return SCFQ_GZ_DECLINE;                   // a damaged further header: the host path decides
          if (h == 0) { finished_end = true; end_byte_here = q + 8; done = true; break; }   // end of file, or trailing garbage (ignored, as gzread does)
          if (sx) return SCFQ_GZ_DECLINE;                      // (a further member behind a stretch: not ONE member)
          p = (q + 8 + (uint64_t)h) * 8;
          ++mno;      // (any number of members: the parts of a member, its trailer and the window state travel from batch to batch)
        } else {
          p = r.end_bit;
          if (!stretch_to && p + 8 >= end_bit) return SCFQ_GZ_DECLINE;        // the data ends inside a member: truncated file
        }
      }
      if (done && !tentative) { pos_end = p; member_end_no = mno; break; }
      if (gaps.empty() || round >= 4) {
        if (verbose) std::fprintf(stderr, "scfq gzdev: batch %u: chain not closed after %d rounds (%zu gaps): host path\n", k, round, gaps.size());
        return SCFQ_GZ_DECLINE;
      }
      // (a gap round — a search, a handful of segments decoded, the host waiting for both — runs on its own high-priority stream when the
      // file has several batches: queued on the compute stream it stood behind the previous batch's window kernels, which wait for the
      // LDS of a CU that two decode workgroups still hold: 9 ms per round instead of 2 - 3)
      hipStream_t s_gap = c->compute;
      if (nb > 1) {
        if (!g.s_gap) {
          int least = 0, greatest = 0;
          if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest) HIPCHK(hipStreamCreateWithPriority(&g.s_gap, hipStreamNonBlocking, greatest));
          else HIPCHK(hipStreamCreateWithFlags(&g.s_gap, hipStreamNonBlocking));
        }
        s_gap = g.s_gap;
      }
      // A gap is as long as a planned segment and one wave would take a whole decode phase for it: it is cut into pieces
      // the same way the file was — its exact start, then block starts searched at equal steps inside it
      std::vector<Gap> pieces;       // (start, stop, room multiplier)
      {
        const uint32_t kSub = 8;
        std::vector<uint64_t> from;
        for (auto& gp : gaps) {
          if (by_start.count(gp.first)) continue;          // (decoded in an earlier round after all)
          const uint64_t span = gp.second - gp.first;
          const uint32_t nparts = span >= (uint64_t)kSub * 8 * 4096 ? kSub : 1;
          for (uint32_t j = 1; j < nparts; ++j) from.push_back(gp.first + span * j / nparts);
        }
        std::vector<uint64_t> found(from.size(), ~0ull);
        if (!from.empty() && from.size() <= max_seg) {
          std::memcpy(h_gfrom, from.data(), 8 * from.size());
          HIPCHK(hipMemcpyAsync(g.d_pmeta[pp] + offp_gfrom, h_gfrom, 8 * from.size(), hipMemcpyHostToDevice, s_gap));
          hipLaunchKernelGGL(gz_sync_search, dim3((unsigned)from.size()), dim3(kSyncThreads), 0, s_gap, reinterpret_cast<const uint64_t*>(vbase), limit,
                             reinterpret_cast<const uint64_t*>(g.d_pmeta[pp] + offp_gfrom), (uint32_t)from.size(), seg_bytes * 8,
                             reinterpret_cast<uint64_t*>(g.d_pmeta[pp] + offp_gfound), lit_mask);
          HIPCHK(hipGetLastError());
          HIPCHK(hipMemcpyAsync(h_gfound, g.d_pmeta[pp] + offp_gfound, 8 * from.size(), hipMemcpyDeviceToHost, s_gap));
          HIPCHK(hipStreamSynchronize(s_gap));
          for (size_t q = 0; q < from.size(); ++q) found[q] = h_gfound[q];
        }
        size_t fk = 0;
        for (auto& gp : gaps) {
          if (by_start.count(gp.first)) continue;
          const uint64_t span = gp.second - gp.first;
          const uint32_t nparts = span >= (uint64_t)kSub * 8 * 4096 ? kSub : 1;
          std::vector<uint64_t> st{gp.first};
          for (uint32_t j = 1; j < nparts; ++j, ++fk)
            if (fk < found.size() && found[fk] != ~0ull && found[fk] > st.back() && found[fk] < gp.second && !by_start.count(found[fk])) st.push_back(found[fk]);
          for (size_t q = 0; q < st.size(); ++q) pieces.push_back(Gap{st[q], q + 1 < st.size() ? st[q + 1] : gp.second, gp.mult});
        }
      }
      if (n_seg + pieces.size() > max_seg) { why = "more gap segments than a batch has room for"; return gz_decline(__LINE__, kFallbackRest); }
      const uint32_t first = n_seg;
      for (auto& gp : pieces) {
        GzSeg& sg = h_segs[n_seg];
        sg.start_bit = gp.first;
        sg.stop_bit = gp.second;
        const uint64_t cap = seg_cap(gp.first, gp.second, gp.mult);
        if (pool_used + kGzWindow + cap > 2 * res_out) { why = "a batch's symbols outgrow every bound"; return gz_decline(__LINE__, kFallbackRest); }
        if (int r = gz_take(sl.sym, kGzWindow + cap, &sg.sym_off)) return r;      // (a new chunk if need be: the decode of the next batch, in the other slot, is not disturbed)
        sg.cap = (uint32_t)cap;
        sg.reserved = 0;
        pool_used += kGzWindow + cap;
        ++n_seg;
      }
      if (n_seg == first) return SCFQ_GZ_DECLINE;
      if (verbose) std::fprintf(stderr, "scfq gzdev:   batch %u round %d: %u gap segments\n", k, round + 1, n_seg - first);
      ++n_gap_rounds;
      HIPCHK(hipMemcpyAsync(sl.d_meta + off_segs + sizeof(GzSeg) * first, h_segs + first, sizeof(GzSeg) * (n_seg - first), hipMemcpyHostToDevice, s_gap));
      hipLaunchKernelGGL(gz_segment_decode, dim3((n_seg - first + kWavesPerWg - 1) / kWavesPerWg), dim3(64 * kWavesPerWg), kWavesPerWg * kGzWaveLdsBytes, s_gap,
                         vbase, copy_end_of(k), reinterpret_cast<const GzSeg*>(sl.d_meta + off_segs) + first, n_seg - first, sl.sym.base(),
                         reinterpret_cast<GzSegOut*>(sl.d_meta + off_outs) + first, inflate_serial_loop());
      HIPCHK(hipGetLastError());
      HIPCHK(hipMemcpyAsync(h_outs + first, sl.d_meta + off_outs + sizeof(GzSegOut) * first, sizeof(GzSegOut) * (n_seg - first), hipMemcpyDeviceToHost, s_gap));
      HIPCHK(hipStreamSynchronize(s_gap));
      n_decoded_total += n_seg - first;
    }
    n_seg_of[k] = n_seg;

    // ---- the batch's part of the chain: windows, bytes, CRC tiles, scan --------------------------------------------------------
    const uint32_t n_chain = (uint32_t)chain.size();
    n_chain_total += n_chain;
    GzChain* h_chain = reinterpret_cast<GzChain*>(g.h_pmeta[pp] + offp_chain);
    uint32_t* h_first = reinterpret_cast<uint32_t*>(g.h_pmeta[pp] + offp_first);
    uint32_t* h_we = reinterpret_cast<uint32_t*>(g.h_pmeta[pp] + offp_we);
    uint32_t* h_wt = reinterpret_cast<uint32_t*>(g.h_pmeta[pp] + offp_wt);
    uint64_t batch_out = 0, n_work = 0;
    uint32_t n_chains = 0, valid_end = valid;
    std::vector<PartHere> parts_here;
    {
      uint32_t v = valid, cur_m = 0xFFFFFFFFu;
      for (uint32_t q = 0; q < n_chain; ++q) {
        const uint32_t s = chain[q].seg;
        if (chain[q].member != cur_m) {
          v = (chain[q].member == member_no) ? valid : 0u;   // a further member starts with an empty window
          cur_m = chain[q].member;
          h_first[n_chains++] = q;
          parts_here.push_back(PartHere{cur_m, batch_out, 0});
        }
        h_chain[q].sym_off = h_segs[s].sym_off;
        h_chain[q].out_off = batch_out;
        h_chain[q].n_sym = h_outs[s].n_sym;
        h_chain[q].valid_before = v;
        h_chain[q].chain_id = n_chains - 1;
        h_chain[q].reserved = 0;
        batch_out += h_outs[s].n_sym;
        parts_here.back().len += h_outs[s].n_sym;
        v = (uint32_t)std::min<uint64_t>(kGzWindow, (uint64_t)v + h_outs[s].n_sym);
        const uint32_t nt = (h_outs[s].n_sym + kResolveTile - 1) / kResolveTile;
        if (n_work + nt > max_work) { why = "more resolve tiles than a batch's tables hold"; return gz_decline(__LINE__, kFallbackRest); }
        for (uint32_t t = 0; t < nt; ++t) { h_we[n_work] = q; h_wt[n_work] = t; ++n_work; }
      }
      h_first[n_chains] = n_chain;
      if (n_chain) valid_end = (!finished_end && chain[n_chain - 1].member == member_end_no) ? v : 0u;
    }
    if (batch_out + 2 * kStagePad > res_out) { why = "a batch inflates to more than the reserved range"; return gz_decline(__LINE__, kFallbackRest); }
    {
      // (every test that can send this batch elsewhere comes before the first launch that writes the batch's bytes)
      uint64_t nt_all = 0;
      for (const PartHere& ph : parts_here) nt_all += (ph.len + kCrcTile - 1) / kCrcTile;
      if (4 * (16 + tiles_used + nt_all) > res_crc) { why = "more CRC tiles than the reserved range holds"; return gz_decline(__LINE__, kFallbackRest); }
    }
    walk_ms += std::chrono::duration<double, std::milli>(clk::now() - tw).count();
    if (map_only) {
      // pass 1 of a stretch: no windows, no bytes — what this batch's chain does to the window, folded into the stretch's running map
      if (n_chain) {
        if (n_chains != 1) return SCFQ_GZ_DECLINE;
        GzChain* h_gchain = reinterpret_cast<GzChain*>(g.h_pmeta[pp] + offp_gchain);      // (unused by the kernels below; kept zero)
        uint32_t* h_gfirst = reinterpret_cast<uint32_t*>(g.h_pmeta[pp] + offp_gfirst);
        uint32_t n_groups = 0;
        for (uint32_t q = 0; q < n_chain; q += group) {
          if (n_groups >= max_groups) return SCFQ_GZ_DECLINE;
          h_gfirst[n_groups] = q;
          h_gchain[n_groups] = GzChain{};
          ++n_groups;
        }
        h_gfirst[n_groups] = n_chain;
        if (int r = gz_buf(g, g.maps, 2ull * kGzWindow * (n_groups + 1))) return r;
        HIPCHK(hipMemcpyAsync(g.d_pmeta[pp] + offp_chain, g.h_pmeta[pp] + offp_chain, sizeof(GzChain) * n_chain, hipMemcpyHostToDevice, c->compute));
        HIPCHK(hipMemcpyAsync(g.d_pmeta[pp] + offp_gfirst, h_gfirst, 4ull * (n_groups + 1), hipMemcpyHostToDevice, c->compute));
        span_begin(sp_chain, c->compute);
        hipLaunchKernelGGL(gz_window_maps, dim3(n_groups), dim3(1024), 0, c->compute, reinterpret_cast<const GzChain*>(g.d_pmeta[pp] + offp_chain),
                           reinterpret_cast<const uint32_t*>(g.d_pmeta[pp] + offp_gfirst), sl.sym.base(), reinterpret_cast<uint16_t*>(g.maps.p));
        hipLaunchKernelGGL(gz_map_fold, dim3(1), dim3(1024), 0, c->compute, reinterpret_cast<const uint16_t*>(g.maps.p), n_groups,
                           reinterpret_cast<uint16_t*>(g.gwin.p), reinterpret_cast<uint16_t*>(g.gwin.p) + kGzWindow);
        HIPCHK(hipGetLastError());
        span_end(sp_chain, c->compute);
        if (keep) {
          // the entries' output symbols move to the store (this batch's pool is the batch after next's), the tables wait on the host
          BatchKept bk;
          bk.k = k; bk.n_chain = n_chain; bk.n_chains = n_chains; bk.n_work = n_work; bk.batch_out = batch_out;
          bk.parts_here = parts_here;
          bk.chain.assign(h_chain, h_chain + n_chain);
          bk.first.assign(h_first, h_first + n_chains + 1);
          bk.we.assign(h_we, h_we + n_work);
          bk.wt.assign(h_wt, h_wt + n_work);
          std::vector<uint64_t> tab(3ull * n_chain);
          for (uint32_t q = 0; q < n_chain; ++q) {
            uint64_t off = 0;
            if (gz_take(keep_store, std::max<uint64_t>(8, ((uint64_t)h_chain[q].n_sym + 7) & ~7ull), &off)) return SCFQ_GZ_DECLINE;
            tab[3ull * q] = h_chain[q].sym_off + kGzWindow;
            tab[3ull * q + 1] = off;
            tab[3ull * q + 2] = h_chain[q].n_sym;
            bk.chain[q].sym_off = off - kGzWindow;       // (the kernels add the marker prefix's length back: 64-bit wrap-around)
          }
          uint64_t* d_tab = reinterpret_cast<uint64_t*>(pack_tab.p) + 3ull * max_seg * k;
          HIPCHK(hipMemcpy(d_tab, tab.data(), 24ull * n_chain, hipMemcpyHostToDevice));
          hipLaunchKernelGGL(gz_pack_symbols, dim3(n_chain), dim3(256), 0, c->compute, sl.sym.base(), keep_store.base(), d_tab);
          HIPCHK(hipGetLastError());
          kept.push_back(std::move(bk));
        }
      }
      HIPCHK(hipEventRecord(g.ev_post[pp], c->compute));
      total_out += batch_out;
      pos = pos_end;
      member_no = member_end_no;
      finished = finished_end;
      if (finished_end && !stretch_to) { end_byte = end_byte_here; for (const GzMemberEnd& me : ends_here) member_ends.push_back(me); }
      release_comp(k);
      return SCFQ_OK;
    }
    if (n_chain) {
      const bool carry_in = chain[0].member == member_no && valid > 0;                                     // the first chain goes on inside a member begun earlier
      const bool carry_out = !finished_end && chain[n_chain - 1].member == member_end_no;               // the last chain's member goes on in the next batch
      if (int r = post_process(k, pp, sl.sym.base(), n_chain, n_chains, n_work, batch_out, parts_here, carry_in, carry_out)) return r;
    }
    HIPCHK(hipEventRecord(g.ev_post[pp], c->compute));
    for (const GzMemberEnd& me : ends_here) member_ends.push_back(me);
    total_out += batch_out;
    pos = pos_end;
    member_no = member_end_no;
    valid = valid_end;
    finished = finished_end;
    if (finished_end) end_byte = end_byte_here;
    release_comp(k);
    if (k == 0) trace("gzip engine: first batch walked, its windows / bytes / CRC / scan queued");
    return SCFQ_OK;
  };

  // ---- every member's ISIZE and CRC-32 (members that ended so far; what is left over is the prefix of the member `pos` lies in) --
  size_t parts_checked = 0, ends_checked = 0;
  uint32_t prefix_raw = 0;
  uint64_t prefix_len = 0;
  auto check_members = [&]() -> int {
    HIPCHK(hipMemcpyAsync(g.h_crc, g.crc.p, 4 * (16 + tiles_used), hipMemcpyDeviceToHost, c->compute));
    HIPCHK(hipStreamSynchronize(c->compute));
    const uint32_t* hc = reinterpret_cast<const uint32_t*>(g.h_crc);
    if (hc[0]) return SCFQ_GZ_DECLINE;                 // a reference before a member's start
    const uint32_t x_tile = gz_xpow8n(kCrcTile);
    auto fold_part = [&](const GzPart& pt, uint32_t* raw, uint64_t* len) {
      uint32_t r = 0;
      for (uint64_t t = 0; t < pt.n_tiles; ++t) r = gz_mulmod(x_tile, r) ^ hc[16 + pt.tile_at + t];
      *raw = gz_mulmod(gz_xpow8n(pt.len), *raw) ^ r;
      *len += pt.len;
    };
    for (; ends_checked < member_ends.size(); ++ends_checked) {
      const GzMemberEnd& me = member_ends[ends_checked];
      uint32_t raw = 0;
      uint64_t len = 0;       // (64 bits: ISIZE is the length modulo 2^32, a member may be longer)
      for (; parts_checked < parts.size() && parts[parts_checked].member == me.member; ++parts_checked) fold_part(parts[parts_checked], &raw, &len);
      if (sx) {
        // a stretch of a member that other ranks hold the rest of: its raw CRC-32 and length go to the caller, who joins the
        // stretches' and checks them against this trailer
        sx->crc_raw = raw;
        sx->out_bytes = len;
        sx->member_ended = true;
        sx->trailer_crc = me.crc;
        sx->trailer_isize = me.isize;
        continue;
      }
      const uint32_t crc = raw ^ gz_mulmod(gz_xpow8n(len), 0xFFFFFFFFu) ^ 0xFFFFFFFFu;
      if ((uint32_t)(len & 0xFFFFFFFFull) != me.isize || crc != me.crc) {
        if (verbose) std::fprintf(stderr, "scfq gzdev: member %u: %llu bytes, CRC-32 %08x; its trailer says %u, %08x: host path\n", me.member,
                                  (unsigned long long)len, crc, me.isize, me.crc);
        return SCFQ_GZ_DECLINE;
      }
    }
    prefix_raw = 0; prefix_len = 0;
    for (size_t q = parts_checked; q < parts.size(); ++q) {
      if (parts[q].member != member_no) return SCFQ_GZ_DECLINE;
      fold_part(parts[q], &prefix_raw, &prefix_len);
    }
    return SCFQ_OK;
  };

  // ---- the pipeline (r5: event-driven) ------------------------------------------------------------------------------------------------
  // Three cursors over the batches: ns searches queued, nd decodes queued, nw batches walked.  Whatever CAN be queued is queued, in this
  // order of urgency: a search as soon as the batch's compressed bytes are on their way (the copier runs n_comp batches ahead of the
  // walk at most), a decode as soon as its search is through and a set of symbols is free (batch k's set is free once batch
  // k - n_slots has been walked: its resolve is queued by then, and the decode waits for it on the device), the walk of the oldest batch once its
  // decode is through.  Round 4's loop did one search, one decode and one walk per iteration in a fixed order, each stage waiting for
  // its own event: at most two decode kernels were ever queued, and the second only once the first one's predecessor had been walked
  // — the device ran decode kernels one after the other with gaps of 1 - 3 ms (profiles/r04/gz_timeline_after_border_stop.txt).
  static const bool warm_copies = env_int("SCFQ_GZ_DEVICE_WARM_COPIES", 1) != 0;
  if (warm_copies && !g.copies_warmed && nb > 1) {
    // (r5) With the host-writes feed no host-to-device copy precedes the first search's table, and a process's FIRST copy of tens of KB in
    // either direction costs 9 - 10 ms (the runtime brings a copy engine's queue up; a copy of 8 bytes goes another way and costs nothing): the
    // orchestrating thread spends them here, where it would otherwise wait for the first batch's bytes, instead of between their arrival and
    // the first search (profiles/r05/cold_first_launches.txt)
    g.copies_warmed = true;
    HIPCHK(hipMemcpyAsync(g.d_search[0], g.h_search[0], 8ull * max_seg, hipMemcpyHostToDevice, s_search));
    HIPCHK(hipMemcpyAsync(g.h_search[0] + off_found, g.d_search[0] + off_found, 8ull * max_seg, hipMemcpyDeviceToHost, s_search));
    // (... and the first device-to-device copy — the staging buffer's bytes moving on — brings the runtime's copy kernels in: 9 ms, which
    // fell between the first batch's arrival and its search in two processes of three)
    if (host_writes && g.win.cap >= 2 * 65536) HIPCHK(hipMemcpyAsync(g.win.p + 65536, g.win.p, 65536, hipMemcpyDeviceToDevice, c->copy));
    HIPCHK(hipStreamSynchronize(s_search));            // (the first search's table is written into h_search[0] next)
    if (host_writes) HIPCHK(hipStreamSynchronize(c->copy));
    trace("gzip engine: the search stream's first copies done");
  }
  int fail = SCFQ_OK;
  {
    uint32_t ns = 0, nd = 0, nw = 0;
    auto copy_queued = [&](uint32_t k) { std::lock_guard<std::mutex> lk(cp.mu); return cp.enq > k || cp.rc != SCFQ_OK; };
    auto event_done = [&](hipEvent_t ev, int* err) {
      const hipError_t e = hipEventQuery(ev);
      if (e == hipSuccess) return true;
      (void)hipGetLastError();             // (hipErrorNotReady is remembered as the thread's last error: the launches' checks must not see it)
      if (e != hipErrorNotReady) { std::snprintf(g_err, sizeof g_err, "hipEventQuery -> %s", hipGetErrorString(e)); *err = SCFQ_EHIP; }
      return false;
    };
    auto timed = [&](int which, auto&& f) { const auto t0 = clk::now(); const int r = f(); stage_ms[which] += std::chrono::duration<double, std::milli>(clk::now() - t0).count(); return r; };
    while (!fail && !finished && nw < nb) {
      bool progressed = false;
      while (!fail && ns < nb && ns < nw + n_comp && copy_queued(ns)) { fail = timed(0, [&] { return stage_a(ns); }); ++ns; progressed = true; }
      while (!fail && nd < ns && nd < nw + n_slots && event_done(g.ev_found[nd % n_comp], &fail)) { fail = timed(1, [&] { return stage_b(nd); }); ++nd; progressed = true; }
      if (!fail && nw < nd && event_done(g.ev_dec[nw % n_slots], &fail)) { fail = timed(2, [&] { return stage_c(nw); }); ++nw; progressed = true; }
      if (!fail && !progressed) {
        // nothing to queue: the next thing to happen is a copy, a search or a decode finishing.  Short sleeps instead of a blocking
        // wait — whichever of the three comes first is acted on within ~30 us (one host thread; a wait for one event would sit out the others)
        const auto t0 = clk::now();
        std::this_thread::sleep_for(std::chrono::microseconds(20));
        const double w = std::chrono::duration<double, std::milli>(clk::now() - t0).count();
        if (nw < nd) h_dec_wait_ms += w; else if (nd < ns) h_search_wait_ms += w; else h_copier_wait_ms += w;
      }
    }
  }
  copier_join.now();                                    // (batches behind the end of the last member are not copied any more; the ring is free for a hand-over)
  if (fail < 0) return fail;
  if (!fail && !finished) return SCFQ_GZ_DECLINE;      // the data ended inside a member
  bool resumed = false;
  if (fail) {
    // A batch the device path cannot take for want of room (kFallbackRest: nothing suspect about the data).  What the batches before
    // it folded STAYS: the host's decoder carries on from the exact bit the chain has reached, with the window in front of it and
    // the CRC-32 / length of the member so far — instead of round 2's restart of the whole file on the host.  (Before the first
    // batch is through, and for anything that smells of damaged data, the whole file is still the host's: its readers are
    // gzread byte for byte, error text included.)
    static const bool resume_on = env_int("SCFQ_GZ_DEVICE_RESUME", 1) != 0;
    if (fail != kFallbackRest || total_out == 0 || !resume_on || sx) return SCFQ_GZ_DECLINE;
    for (hipStream_t st : {c->copy, s_search, s_dec[0], s_dec[1], s_dec[2], c->compute}) HIPCHK(hipStreamSynchronize(st));
    if ((rc = check_members())) return rc;
    std::vector<uint8_t> window(kGzWindow, 0);
    if (valid) HIPCHK(hipMemcpy(window.data(), g.d_wcarry + (uint64_t)wcarry * kGzWindow, kGzWindow, hipMemcpyDeviceToHost));
    int prev = -1;
    if (have_prev_out) { uint8_t pb = 0; HIPCHK(hipMemcpy(&pb, d_out + prev_out_bytes - 1, 1, hipMemcpyDeviceToHost)); prev = pb; }
    const uint32_t crc_prefix = prefix_raw ^ gz_mulmod(gz_xpow8n(prefix_len), 0xFFFFFFFFu) ^ 0xFFFFFFFFu;
    if (verbose) std::fprintf(stderr, "scfq gzdev: %s: %llu bytes are through, the host's decoder takes the file from bit %llu (%.1f %%)\n", why ? why : "a batch without room",
                              (unsigned long long)total_out, (unsigned long long)pos, 100.0 * (double)pos / (double)end_bit);
    struct RestSource : Source {
      scfq_gzfast::Resume rs;
      int64_t fill(uint8_t* dst, uint64_t cap) override { const int64_t r = rs.next_chunk(dst, cap); return r < 0 ? (int64_t)SCFQ_EGZ : r; }
    } rest;
    rest.rs.open(img, (size_t)fsize, pos, window.data(), valid, crc_prefix, prefix_len);
    rc = ingest(c, rest, prev, flags, 64ull << 20, timing);
    if (rc) return rc;
    end_byte = rest.rs.end_offset();
    resumed = true;
  } else if (keep) {
    // ---- the stretch's map goes out, the window in front of it comes back, and the kept symbols become bytes ------------------------
    sx->out_bytes = total_out;
    sx->member_ended = !stretch_to;
    sx->end_byte = end_byte;
    sx->map.resize(kGzWindow);
    HIPCHK(hipStreamSynchronize(c->compute));
    HIPCHK(hipMemcpy(sx->map.data(), g.gwin.p, 2ull * kGzWindow, hipMemcpyDeviceToHost));
    exchange_guard.made = true;
    if (sx->exchange(*sx, true)) return SCFQ_GZ_DECLINE;
    valid = sx->valid;
    wcarry = 0;
    if (valid) HIPCHK(hipMemcpy(g.d_wcarry, sx->window, kGzWindow, hipMemcpyHostToDevice));
    for (size_t i = 0; i < kept.size(); ++i) {
      BatchKept& bk = kept[i];
      const int pp = (int)(i % n_slots);
      if (i >= n_slots) HIPCHK(hipEventSynchronize(g.ev_post[pp]));      // the pinned tables of this parity were the batch's before last
      GzChain* h_chain = reinterpret_cast<GzChain*>(g.h_pmeta[pp] + offp_chain);
      uint32_t v = valid;
      for (uint32_t q = 0; q < bk.n_chain; ++q) {
        h_chain[q] = bk.chain[q];
        h_chain[q].valid_before = v;
        v = (uint32_t)std::min<uint64_t>(kGzWindow, (uint64_t)v + bk.chain[q].n_sym);
      }
      std::memcpy(g.h_pmeta[pp] + offp_first, bk.first.data(), 4ull * bk.first.size());
      std::memcpy(g.h_pmeta[pp] + offp_we, bk.we.data(), 4ull * bk.we.size());
      std::memcpy(g.h_pmeta[pp] + offp_wt, bk.wt.data(), 4ull * bk.wt.size());
      if ((rc = post_process(bk.k, pp, keep_store.base(), bk.n_chain, bk.n_chains, bk.n_work, bk.batch_out, bk.parts_here, valid > 0, i + 1 < kept.size()))) return rc;
      HIPCHK(hipEventRecord(g.ev_post[pp], c->compute));
      valid = v;
    }
    if ((rc = check_members())) return rc;
  } else if (map_only) {
    // (pass 1 of a stretch: no bytes, no CRC)
  } else if ((rc = check_members())) {
    return rc;
  } else if (parts_checked != parts.size() && !stretch_to) {
    return SCFQ_GZ_DECLINE;
  }
  if (sx) {
    sx->end_byte = end_byte;
    if (map_only && !keep) {
      sx->out_bytes = total_out;
      sx->member_ended = !stretch_to;
      sx->map.resize(kGzWindow);
      HIPCHK(hipStreamSynchronize(c->compute));
      HIPCHK(hipMemcpy(sx->map.data(), g.gwin.p, 2ull * kGzWindow, hipMemcpyDeviceToHost));
    } else {
      if (stretch_to) { sx->crc_raw = prefix_raw; sx->out_bytes = prefix_len; }      // (a stretch that ends with the member: set by check_members)
      if (sx->out_bytes != total_out) return SCFQ_GZ_DECLINE;
      sx->first_byte = stretch_first_byte;
    }
  }
  if (end_off) *end_off = end_byte;
  fill_ms += cp.fill_ms;
  c->timing.h2d_bytes += cp.bytes;
  for (const Span& sp : cp.spans) sp_copy.push_back(sp);
  cp.spans.clear();
  c->timing.host_fill_ms += fill_ms;
  c->timing.ingest_wall_ms += std::chrono::duration<double, std::milli>(clk::now() - t_begin).count();
  if (verbose) {
    for (hipStream_t st : {c->copy, s_search, s_dec[0], s_dec[1], s_dec[2]}) (void)hipStreamSynchronize(st);
    auto sum = [](const std::vector<Span>& v) { double t = 0; for (const Span& s : v) { float ms = 0; if (s.a && s.b && hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) t += ms; } return t; };
    std::fprintf(stderr, "scfq gzdev: %-28s %8.2f ms\n", "copy to HBM", sum(sp_copy));
    std::fprintf(stderr, "scfq gzdev: %-28s %8.2f ms\n", "block-start search", sum(sp_search));
    std::fprintf(stderr, "scfq gzdev: %-28s %8.2f ms\n", "segment decode", sum(sp_decode));
    std::fprintf(stderr, "scfq gzdev: %-28s %8.2f ms\n", "chain walk", walk_ms);
    std::fprintf(stderr, "scfq gzdev: %-28s %8.2f ms\n", "window chain", sum(sp_chain));
    std::fprintf(stderr, "scfq gzdev: %-28s %8.2f ms\n", "resolve", sum(sp_resolve));
    std::fprintf(stderr, "scfq gzdev: %-28s %8.2f ms\n", "crc tiles", sum(sp_crc));
    std::fprintf(stderr, "scfq gzdev: %-28s %8.2f ms\n", "scan", sum(sp_scan));
    std::fprintf(stderr, "scfq gzdev: copier thread: pinned-buffer waits %.1f, memcpy %.1f, copy enqueue %.1f ms; orchestrating thread: copier waits %.1f, search waits %.1f, decode waits %.1f, post waits %.1f ms\n",
                 cp.evsync_ms, cp.memcpy_ms, cp.enqueue_ms, h_copier_wait_ms, h_search_wait_ms, h_dec_wait_ms, h_post_wait_ms);
    std::fprintf(stderr, "scfq gzdev: host time in stage A (copy, search) %.1f, B (plan, decode launch) %.1f, C (walk, windows, bytes, scan) %.1f ms; of all that inside device allocations %.1f ms\n",
                 stage_ms[0], stage_ms[1], stage_ms[2], gz_alloc_ms());
    std::fprintf(stderr, "scfq gzdev: %-28s %8.2f ms\n", "wall", std::chrono::duration<double, std::milli>(clk::now() - t_begin).count());
    std::fprintf(stderr, "scfq gzdev: %u batch(es), %u segments planned, %u decoded (%u gap rounds, %u segments given more room), %u on the chain, %zu member(s), %llu bytes inflated%s\n", nb,
                 n_planned_total, n_decoded_total, n_gap_rounds, n_overflows, n_chain_total, member_ends.size(), (unsigned long long)total_out, resumed ? " on the device, the rest on the host" : "");
    std::fprintf(stderr, "scfq gzdev: device memory high water %.2f GB (this path holds %.2f GB now)\n", (double)g_dev_high.load() / 1e9, (double)g.held() / 1e9);
#ifdef SCFQ_SPROF
    { unsigned long long w[8]; HIPCHK(hipMemcpyFromSymbol(w, HIP_SYMBOL(scfq_dinflate::g_sprof), sizeof w));
      const double ch = (double)std::max<unsigned long long>(1, w[3]);
      std::fprintf(stderr, "sprof: %llu chunks of 65536 positions; cycles per chunk (thread 0's clock): stage %.0f, fields + Kraft %.0f, deep test %.0f; %.1f candidates reach the deep test per chunk\n",
                   w[3], w[0] / ch, w[1] / ch, w[2] / ch, w[4] / ch);
      unsigned long long z[8] = {0}; HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(scfq_dinflate::g_sprof), z, sizeof z)); }
#endif
  }
  return SCFQ_OK;
}

// An engine of the device's pool for the length of the call (GzShared, scfq_api.hip); every stream of the path is idle when this
// returns, whatever the outcome — the scan of the last batch included, which reads the engine's output buffer: the buffers belong to
// the next caller.
// end_off (optional): the offset just behind the last member's trailer — fsize unless bytes that are not a gzip member follow it
// fd / fd_off (optional): the file img is mapped from, img[0] being its byte fd_off — the copies into the pinned ring then pread
int ingest_gz_device(Ctx* c, const uint8_t* img, uint64_t fsize, uint32_t flags, bool timing, uint64_t* end_off = nullptr, int fd = -1, uint64_t fd_off = 0, GzStretch* sx = nullptr) {
  GzShared& gs = gz_shared(c->dev);
  static const int n_engines = std::min((int)GzShared::kMax, std::max(1, env_int("SCFQ_GZ_DEVICE_ENGINES", 4)));
  // (a STRETCH — one rank's share of a sharded member — is never "big" and only waits for a free engine: its call holds the engine across a
  // collective — the exchange of the window maps —, so ranks that are threads of one process on one device would otherwise wait for each
  // other's engines until the exchange times out.  More such ranks than engines, SCFQ_GZ_DEVICE_ENGINES = 4, must be separate processes.)
  const bool big = !sx && fsize > (1ull << 30);
  int e = -1;
  {
    std::unique_lock<std::mutex> lk(gs.mu);
    // (a big file waits for the pool to drain, and while one waits no further small file is admitted: a steady stream of small
    // sessions would otherwise keep n_busy above zero for ever and the big one — which holds its context all along — would starve)
    if (big) ++gs.big_waiting;
    gs.cv.wait(lk, [&] { return sx ? gs.n_busy < n_engines : (!gs.big_running && (big ? gs.n_busy == 0 : (gs.big_waiting == 0 && gs.n_busy < n_engines))); });
    if (big) --gs.big_waiting;
    for (int k = 0; k < n_engines; ++k) if (!gs.busy[k]) { e = k; break; }
    gs.busy[e] = true;
    ++gs.n_busy;
    gs.big_running = big;
  }
  GzDevBuffers& g = gs.buf[e];
  const int rc = ingest_gz_device_batches(c, g, img, fsize, flags, timing, end_off, fd, fd_off, sx);
  if (c->copy) (void)hipStreamSynchronize(c->copy);
  if (g.s_search) (void)hipStreamSynchronize(g.s_search);
  if (g.s_gap) (void)hipStreamSynchronize(g.s_gap);
  for (int b = 0; b < kGzDecodeStreams; ++b) if (g.s_decode[b]) (void)hipStreamSynchronize(g.s_decode[b]);
  if (c->compute) (void)hipStreamSynchronize(c->compute);
  gz_free_retired(&g);
  {
    std::lock_guard<std::mutex> lk(gs.mu);
    gs.busy[e] = false;
    --gs.n_busy;
    if (big) gs.big_running = false;
  }
  gs.cv.notify_all();
  return rc;
}
