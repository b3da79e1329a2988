// scfq_gzdev.hpp — host side of the device gzip inflate (kernels: gz_inflate_kernels.hpp).  Included by scfq_api.hip
// inside its anonymous namespace (uses Ctx, HIPCHK, scan_async, parallel_pieces, kFallbackToHost).
//
// ingest_gz_device(): the COMPRESSED file goes through the pinned ring to HBM in BATCHES of 4096 planned segments (128 KiB
// of compressed data each), and three batches are in flight at once:
//      batch k+2   copy (copy stream) + block-start search (search stream)
//      batch k+1   segment decode to 16-bit symbols (one of two decode streams)
//      batch k     chain walk (host), window chain, resolve, CRC-32 tiles, scan (compute stream)
// so the PCIe copy and everything behind the decode hide under the decode of the next batches, the decode kernels of two
// batches overlap (the slow waves at the end of one no longer idle the device), and the device memory in use is that of
// three batches whatever the size of the file (the one-batch form of this path needed 14 x the compressed size in HBM).
// A decoded segment is only believed when the walk reaches its start bit EXACTLY — from the member's first block, through
// every segment's end bit, over member trailers and headers, across batch borders — so a false sync, a corrupt block or a
// truncated file can never contribute a byte: the walk stops there, gaps it can prove (a false sync skipped, the first block
// of a further member) are decoded in a further round, and anything else hands the whole file to the host readers
// (kFallbackToHost; the caller starts its session again, so what earlier batches folded is dropped), which reproduce
// gzread's behaviour for it byte by byte (partial output, SCFQ_EGZ).  CRC-32 and ISIZE of every member are checked before
// the result is handed out.

inline bool gz_device_enabled() {
  static const bool v = [] { const char* e = std::getenv("SCFQ_GZ_DEVICE"); return e ? e[0] != '0' : true; }();
  return v;
}

template <typename T>
int gz_grow(T** p, uint64_t* cap, uint64_t want_bytes, bool pinned = false) {
  if (*cap >= want_bytes) return SCFQ_OK;
  if (*p) { if (pinned) (void)hipHostFree(*p); else (void)hipFree(*p); }
  *p = nullptr;
  *cap = 0;
  const uint64_t bytes = want_bytes + want_bytes / 8 + 4096;
  const hipError_t e = pinned ? hipHostMalloc(reinterpret_cast<void**>(p), bytes, hipHostMallocDefault) : hipMalloc(reinterpret_cast<void**>(p), bytes);
  if (e != hipSuccess) { (void)hipGetLastError(); *p = nullptr; return kFallbackToHost; }      // not enough memory: the host path needs none of this
  *cap = bytes;
  return SCFQ_OK;
}

inline void gz_free(GzDevBuffers* g) {
  for (int b = 0; b < 3; ++b) {
    if (g->d_comp[b]) (void)hipFree(g->d_comp[b]);
    if (g->ev_copy[b]) (void)hipEventDestroy(g->ev_copy[b]);
  }
  for (int b = 0; b < 2; ++b) {
    if (g->slot[b].d_sym) (void)hipFree(g->slot[b].d_sym);
    if (g->slot[b].d_meta) (void)hipFree(g->slot[b].d_meta);
    if (g->slot[b].h_meta) (void)hipHostFree(g->slot[b].h_meta);
    if (g->d_pmeta[b]) (void)hipFree(g->d_pmeta[b]);
    if (g->h_pmeta[b]) (void)hipHostFree(g->h_pmeta[b]);
    if (g->ev_dec[b]) (void)hipEventDestroy(g->ev_dec[b]);
    if (g->ev_post[b]) (void)hipEventDestroy(g->ev_post[b]);
    if (g->s_decode[b]) (void)hipStreamDestroy(g->s_decode[b]);
  }
  if (g->d_out) (void)hipFree(g->d_out);
  if (g->d_win) (void)hipFree(g->d_win);
  if (g->d_wcarry) (void)hipFree(g->d_wcarry);
  if (g->d_maps) (void)hipFree(g->d_maps);
  if (g->d_gwin) (void)hipFree(g->d_gwin);
  if (g->d_crc) (void)hipFree(g->d_crc);
  if (g->h_crc) (void)hipHostFree(g->h_crc);
  if (g->s_search) (void)hipStreamDestroy(g->s_search);
  *g = GzDevBuffers{};
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

int ingest_gz_device_batches(Ctx* c, const uint8_t* img, uint64_t fsize, uint32_t flags, bool timing) {
  using namespace scfq_dinflate;
  using clk = std::chrono::steady_clock;
  const auto t_begin = clk::now();
  static const bool verbose = std::getenv("SCFQ_VERBOSE") != nullptr;
  const long h0 = scfq_gzfast::member_header(img, (size_t)fsize);
  if (h0 <= 0 || fsize < 64) return kFallbackToHost;
  GzDevBuffers& g = gz_shared(c->dev).buf;             // (the caller holds its mutex)
  int rc;
  if (!g.s_search) HIPCHK(hipStreamCreateWithFlags(&g.s_search, hipStreamNonBlocking));
  for (int b = 0; b < 3; ++b) if (!g.ev_copy[b]) HIPCHK(hipEventCreateWithFlags(&g.ev_copy[b], hipEventDisableTiming));
  for (int b = 0; b < 2; ++b) {
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
    if (!g.ev_dec[b]) HIPCHK(hipEventCreateWithFlags(&g.ev_dec[b], hipEventDisableTiming));
    if (!g.ev_post[b]) HIPCHK(hipEventCreateWithFlags(&g.ev_post[b], hipEventDisableTiming));
  }

  // ---- plan ----------------------------------------------------------------------------------------------------------------
  const uint64_t data0 = (uint64_t)h0;
  const uint64_t comp = fsize - data0;
  // Segments of 48 to 320 KiB of compressed data (below); batches of about 4096 segments — the decode kernel's resident
  // waves — of equal size.  Two batches in flight is what pays (10 GB of FASTQ, 2.4 GB
  // compressed: one batch 199 ms, two 165 ms, five 217 ms): the second half of the file crosses PCIe while the first is decoded,
  // and a batch's windows, bytes, CRC and scan run under the next one's decode; with more, smaller batches the copies and short
  // kernels crawl between the long-running decode waves.
  static const int seg_kb_env = env_int("SCFQ_GZ_DEVICE_SEGMENT_KB", 0);
  static const uint32_t batch_segs = (uint32_t)std::max(2, env_int("SCFQ_GZ_DEVICE_BATCH_SEGMENTS", 4096));
  // (a file of up to 512 MiB is cut into about 4096 segments — one wave each, the device filled once — of at least 48 KiB; a
  // bigger one into about 8192 of at most 320 KiB: every segment costs 32768 marker symbols and a step of the window chain)
  const uint64_t target_segs = comp <= (512ull << 20) ? 4096 : 8192;
  const uint64_t seg_bytes = seg_kb_env > 0 ? ((((uint64_t)std::max(32, seg_kb_env)) << 10) + 4095) & ~4095ull
                                            : std::min<uint64_t>(320u << 10, std::max<uint64_t>(48u << 10, ((comp + target_segs - 1) / target_segs + 4095) & ~4095ull));
  static const uint64_t ratio = (uint64_t)std::max(2, env_int("SCFQ_GZ_DEVICE_MAX_RATIO", 7));   // output symbols a segment may produce per compressed byte
  const uint64_t n_plan = std::max<uint64_t>(1, (comp + seg_bytes - 1) / seg_bytes);
  std::vector<uint64_t> bstart{0};                     // planned segments [bstart[k], bstart[k + 1]) make batch k
  {
    const uint64_t n_batches = (n_plan + batch_segs - 1) / batch_segs, per = (n_plan + n_batches - 1) / n_batches;
    for (uint64_t at = 0; at < n_plan;) { at = std::min<uint64_t>(n_plan, at + per); bstart.push_back(at); }
  }
  const uint32_t nb = (uint32_t)bstart.size() - 1;
  const uint64_t margin = 4ull << 20;                  // a batch's last segment runs on to the end of its block
  const uint64_t comp_pad = 256;
  const uint64_t end_bit = fsize * 8;
  // geometry of batch k: bytes [byte0, copy_end) of the file are on the device, its territory ends at byte1
  auto p0_of = [&](uint32_t k) { return bstart[k]; };
  auto p1_of = [&](uint32_t k) { return bstart[k + 1]; };
  auto byte0_of = [&](uint32_t k) { return k == 0 ? 0ull : ((data0 + p0_of(k) * seg_bytes) & ~4095ull); };
  auto byte1_of = [&](uint32_t k) { return p1_of(k) == n_plan ? fsize : data0 + p1_of(k) * seg_bytes; };
  auto copy_end_of = [&](uint32_t k) { return std::min<uint64_t>(fsize, byte1_of(k) + margin); };
  const uint64_t batch_comp_max = std::min<uint64_t>(fsize, (uint64_t)batch_segs * seg_bytes + margin + 8192);
  const uint32_t spare = 64 + batch_segs / 16;         // room for gap segments of later rounds
  const uint32_t max_seg = batch_segs + spare;
  // output room of a segment: `ratio` symbols per compressed byte it spans + 256 Ki (it runs on to the end of a block), behind
  // its 32768 markers; a slot's pool holds every planned segment of a batch plus `spare` gap segments of up to four segments' span
  auto seg_cap = [&](uint64_t start_bit, uint64_t stop_bit) { return (((stop_bit - start_bit + 7) / 8) * ratio + 262144 + 7) & ~7ull; };
  const uint64_t pool_syms = batch_comp_max * ratio + (uint64_t)max_seg * (kGzWindow + 262144 + 8) + (uint64_t)spare * 4 * seg_bytes * ratio;
  const uint64_t out_max = batch_comp_max * ratio + (uint64_t)max_seg * 262144;     // inflated bytes of one batch
  // slot meta (device and pinned mirror, same offsets): from | found | segs | outs
  const uint64_t off_from = 0, off_found = off_from + 8ull * max_seg, off_segs = off_found + 8ull * max_seg,
                 off_outs = off_segs + sizeof(GzSeg) * max_seg, slot_meta = off_outs + sizeof(GzSegOut) * max_seg + 64;
  // post meta (per parity): chain | first | work entry | work tile | gap from | gap found | group chain | group first | chain's first group
  static const uint32_t group = (uint32_t)std::min(64, std::max(2, env_int("SCFQ_GZ_DEVICE_CHAIN_GROUP", 64)));      // chain entries per window map
  const uint64_t max_work = out_max / kResolveTile + max_seg + 8;
  const uint64_t max_groups = max_seg / group + 1026 + 2;                  // (every member chain of a batch ends with a short group)
  const uint64_t offp_chain = 0, offp_first = offp_chain + sizeof(GzChain) * max_seg, offp_we = offp_first + 4ull * (max_seg + 2),
                 offp_wt = offp_we + 4ull * max_work, offp_gfrom = (offp_wt + 4ull * max_work + 7) & ~7ull, offp_gfound = offp_gfrom + 8ull * max_seg,
                 offp_gchain = offp_gfound + 8ull * max_seg, offp_gfirst = offp_gchain + sizeof(GzChain) * max_groups,
                 offp_mfirst = offp_gfirst + 4ull * (max_groups + 2), post_meta = offp_mfirst + 4ull * (max_groups + 2) + 64;
  const uint64_t crc_tiles_max = (uint64_t)nb * (out_max / kCrcTile + 2) + 4200;      // (one part per member and batch)
  for (uint32_t b = 0; b < std::min(nb, 3u); ++b)
    if ((rc = gz_grow(&g.d_comp[b], &g.comp_cap[b], batch_comp_max + comp_pad + 4096))) return rc;
  for (uint32_t b = 0; b < std::min(nb, 2u); ++b) {
    GzSlot& sl = g.slot[b];
    if ((rc = gz_grow(&sl.d_sym, &sl.sym_cap, 2ull * pool_syms)) || (rc = gz_grow(&sl.d_meta, &sl.meta_cap, slot_meta)) ||
        (rc = gz_grow(&sl.h_meta, &sl.hmeta_cap, slot_meta, true)) || (rc = gz_grow(&g.d_pmeta[b], &g.pmeta_cap[b], post_meta)) ||
        (rc = gz_grow(&g.h_pmeta[b], &g.hpmeta_cap[b], post_meta, true)))
      return rc;
  }
  if ((rc = gz_grow(&g.d_out, &g.out_cap, out_max + 2 * kStagePad)) || (rc = gz_grow(&g.d_win, &g.win_cap, (uint64_t)kGzWindow * max_seg)) ||
      (rc = gz_grow(&g.d_wcarry, &g.wcarry_cap, 2ull * kGzWindow)) || (rc = gz_grow(&g.d_maps, &g.maps_cap, 2ull * kGzWindow * (max_groups + 1))) ||
      (rc = gz_grow(&g.d_gwin, &g.gwin_cap, (uint64_t)kGzWindow * max_groups)) || (rc = gz_grow(&g.d_crc, &g.crc_cap, 4 * (16 + crc_tiles_max))) ||
      (rc = gz_grow(&g.h_crc, &g.hcrc_cap, 4 * (16 + crc_tiles_max), true)))
    return rc;
  uint8_t* const d_out = g.d_out + kStagePad;
  const double alloc_ms = std::chrono::duration<double, std::milli>(clk::now() - t_begin).count();
  if (verbose && alloc_ms > 1.0) std::fprintf(stderr, "scfq gzdev: buffers grown in %.1f ms (kept in the context for the next call)\n", alloc_ms);
  const uint64_t pin_chunk = 64ull << 20;
  rc = ensure_staging(c, pin_chunk, true);
  if (rc) return rc;

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
  uint64_t pos = data0 * 8;              // the exact bit the chain has reached
  bool finished = false;                 // the last member's final block has been walked and no further member follows
  uint32_t member_no = 0;                // running member that `pos` lies in
  uint32_t valid = 0;                    // bytes of that member's history in front of pos (<= 32768)
  int wcarry = 0;                        // which of the two carried windows is the current one
  uint64_t total_out = 0;
  bool have_prev_out = false;
  uint64_t prev_out_bytes = 0;           // size of the previous batch's output (its last byte is the next batch's look-behind)
  uint64_t tiles_used = 0;
  std::vector<GzPart> parts;
  std::vector<GzMemberEnd> member_ends;
  uint32_t n_planned_total = 0, n_decoded_total = 0, n_chain_total = 0, n_gap_rounds = 0;
  double fill_ms = 0, walk_ms = 0, h_evsync_ms = 0, h_memcpy_ms = 0, h_enqueue_ms = 0, h_search_wait_ms = 0, h_dec_wait_ms = 0, h_post_wait_ms = 0;
  uint32_t pin_it = 0;                   // the pinned staging buffers alternate over the whole file
  std::vector<uint32_t> n_seg_of(nb + 1, 0);
  std::vector<uint64_t> pool_used_of(nb + 1, 0);
  HIPCHK(hipMemsetAsync(g.d_crc, 0, 64, c->compute));          // word 0: the resolve kernels' error status for the whole file

  // ---- stage A(k): bytes of batch k to d_comp[k % 3], block-start search ----------------------------------------------------
  auto stage_a = [&](uint32_t k) -> int {
    const auto tf = clk::now();
    const int cb = (int)(k % 3);
    const uint64_t b0 = byte0_of(k), b1 = copy_end_of(k);
    span_begin(sp_copy, c->copy);
    for (uint64_t off = b0; off < b1; off += pin_chunk, ++pin_it) {
      const int pb = (int)(pin_it & 1);
      const uint64_t len = std::min(pin_chunk, b1 - off);
      auto t0 = clk::now();
      if (pin_it >= 2) HIPCHK(hipEventSynchronize(c->ev_copied[pb]));
      auto t1 = clk::now();
      const uint8_t* src = img + off;
      parallel_pieces(len, [&](uint64_t o, uint64_t l) { std::memcpy(c->h_pin[pb] + o, src + o, l); return 0; });
      auto t2 = clk::now();
      HIPCHK(hipMemcpyAsync(g.d_comp[cb] + (off - b0), c->h_pin[pb], (size_t)len, hipMemcpyHostToDevice, c->copy));
      HIPCHK(hipEventRecord(c->ev_copied[pb], c->copy));
      auto t3 = clk::now();
      h_evsync_ms += std::chrono::duration<double, std::milli>(t1 - t0).count();
      h_memcpy_ms += std::chrono::duration<double, std::milli>(t2 - t1).count();
      h_enqueue_ms += std::chrono::duration<double, std::milli>(t3 - t2).count();
    }
    HIPCHK(hipMemsetAsync(g.d_comp[cb] + (b1 - b0), 0, comp_pad, c->copy));
    span_end(sp_copy, c->copy);
    HIPCHK(hipEventRecord(g.ev_copy[cb], c->copy));
    c->timing.h2d_bytes += b1 - b0;
    fill_ms += std::chrono::duration<double, std::milli>(clk::now() - tf).count();
    // search: the planned segments of this batch (the file's very first one starts exactly at the member's first block)
    GzSlot& sl = g.slot[k & 1];
    uint64_t* h_from = reinterpret_cast<uint64_t*>(sl.h_meta + off_from);
    uint64_t* h_found = reinterpret_cast<uint64_t*>(sl.h_meta + off_found);
    const uint64_t p0 = p0_of(k);
    const uint32_t np = (uint32_t)(p1_of(k) - p0);
    for (uint32_t s = 0; s < np; ++s) { h_from[s] = (data0 + (p0 + s) * seg_bytes) * 8; h_found[s] = ~0ull; }
    const uint32_t s0 = (k == 0) ? 1u : 0u;
    if (k == 0) h_found[0] = data0 * 8;
    HIPCHK(hipStreamWaitEvent(g.s_search, g.ev_copy[cb], 0));
    if (np > s0) {
      const uint8_t* vbase = g.d_comp[cb] - b0;          // virtual base: byte i of the file is vbase[i] for i in [b0, b1 + pad)
      span_begin(sp_search, g.s_search);
      HIPCHK(hipMemcpyAsync(sl.d_meta + off_from, h_from, 8ull * np, hipMemcpyHostToDevice, g.s_search));
      hipLaunchKernelGGL(gz_sync_search, dim3(np - s0), dim3(kSyncThreads), 0, g.s_search, reinterpret_cast<const uint64_t*>(vbase), b1 * 8,
                         reinterpret_cast<const uint64_t*>(sl.d_meta + off_from) + s0, np - s0, seg_bytes * 8,
                         reinterpret_cast<uint64_t*>(sl.d_meta + off_found) + s0);
      HIPCHK(hipGetLastError());
      HIPCHK(hipMemcpyAsync(h_found + s0, sl.d_meta + off_found + 8ull * s0, 8ull * (np - s0), hipMemcpyDeviceToHost, g.s_search));
      span_end(sp_search, g.s_search);
    }
    { auto t0 = clk::now(); HIPCHK(hipStreamSynchronize(g.s_search)); h_search_wait_ms += std::chrono::duration<double, std::milli>(clk::now() - t0).count(); }
    n_planned_total += np;
    return SCFQ_OK;
  };

  // ---- stage B(k): the segments of batch k (its last one stops at the first start of batch k + 1), decode ---------------------
  auto stage_b = [&](uint32_t k) -> int {
    GzSlot& sl = g.slot[k & 1];
    const uint64_t* h_found = reinterpret_cast<const uint64_t*>(sl.h_meta + off_found);
    GzSeg* h_segs = reinterpret_cast<GzSeg*>(sl.h_meta + off_segs);
    const uint32_t np = (uint32_t)(p1_of(k) - p0_of(k));
    const uint64_t limit = copy_end_of(k) * 8;
    uint64_t next_first = (k + 1 < nb) ? byte1_of(k) * 8 : end_bit;     // where the last segment stops
    if (k + 1 < nb) {
      const uint64_t* nf = reinterpret_cast<const uint64_t*>(g.slot[(k + 1) & 1].h_meta + off_found);
      const uint32_t npn = (uint32_t)(p1_of(k + 1) - p0_of(k + 1));
      for (uint32_t s = 0; s < npn; ++s)
        if (nf[s] != ~0ull) { next_first = nf[s]; break; }
      if (next_first + 4096 > limit) next_first = byte1_of(k) * 8;      // (too far: stop at the border and let the walk find the gap)
    }
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
    for (uint32_t q = 0; q < n_seg; ++q) {
      const uint64_t cap = std::min<uint64_t>(seg_cap(h_segs[q].start_bit, h_segs[q].stop_bit), 0x3F000000u);
      h_segs[q].sym_off = pool_used;
      h_segs[q].cap = (uint32_t)cap;     // better compression than `ratio`: overflow status, and the file goes to the host path
      h_segs[q].reserved = 0;
      pool_used += kGzWindow + cap;
    }
    if (pool_used > pool_syms) return kFallbackToHost;
    n_seg_of[k] = n_seg;
    pool_used_of[k] = pool_used;
    hipStream_t sd = g.s_decode[k & 1];
    if (k >= 2) HIPCHK(hipStreamWaitEvent(sd, g.ev_post[k & 1], 0));      // the slot's symbols were read by batch k - 2's resolve
    HIPCHK(hipStreamWaitEvent(sd, g.ev_copy[k % 3], 0));
    if (n_seg) {
      const uint8_t* vbase = g.d_comp[k % 3] - byte0_of(k);
      span_begin(sp_decode, sd);
      HIPCHK(hipMemcpyAsync(sl.d_meta + off_segs, h_segs, sizeof(GzSeg) * n_seg, hipMemcpyHostToDevice, sd));
      hipLaunchKernelGGL(gz_segment_decode, dim3((n_seg + kWavesPerWg - 1) / kWavesPerWg), dim3(64 * kWavesPerWg), kWavesPerWg * kWaveLdsBytes, sd,
                         vbase, copy_end_of(k), reinterpret_cast<const GzSeg*>(sl.d_meta + off_segs), n_seg, sl.d_sym,
                         reinterpret_cast<GzSegOut*>(sl.d_meta + off_outs), inflate_serial_loop());
      HIPCHK(hipGetLastError());
      HIPCHK(hipMemcpyAsync(sl.h_meta + off_outs, sl.d_meta + off_outs, sizeof(GzSegOut) * n_seg, hipMemcpyDeviceToHost, sd));
      span_end(sp_decode, sd);
    }
    HIPCHK(hipEventRecord(g.ev_dec[k & 1], sd));
    n_decoded_total += n_seg;
    return SCFQ_OK;
  };

  // ---- stage C(k): walk, windows, bytes, CRC tiles, scan -----------------------------------------------------------------------
  auto stage_c = [&](uint32_t k) -> int {
    GzSlot& sl = g.slot[k & 1];
    const int pp = (int)(k & 1);
    GzSeg* h_segs = reinterpret_cast<GzSeg*>(sl.h_meta + off_segs);
    GzSegOut* h_outs = reinterpret_cast<GzSegOut*>(sl.h_meta + off_outs);
    { auto t0 = clk::now(); HIPCHK(hipEventSynchronize(g.ev_dec[pp])); h_dec_wait_ms += std::chrono::duration<double, std::milli>(clk::now() - t0).count(); }
    if (k >= 2) { auto t0 = clk::now(); HIPCHK(hipEventSynchronize(g.ev_post[pp])); h_post_wait_ms += std::chrono::duration<double, std::milli>(clk::now() - t0).count(); }      // h_pmeta[pp] / d_pmeta[pp] were batch k - 2's
    const auto tw = clk::now();
    uint64_t* h_gfrom = reinterpret_cast<uint64_t*>(g.h_pmeta[pp] + offp_gfrom);
    uint64_t* h_gfound = reinterpret_cast<uint64_t*>(g.h_pmeta[pp] + offp_gfound);
    uint32_t n_seg = n_seg_of[k];
    uint64_t pool_used = pool_used_of[k];
    const uint8_t* vbase = g.d_comp[k % 3] - byte0_of(k);
    const bool last_batch = k + 1 == nb;
    const uint64_t territory_end = last_batch ? end_bit : byte1_of(k) * 8, limit = copy_end_of(k) * 8;
    struct Entry { uint32_t seg; uint32_t member; };
    std::vector<Entry> chain;
    uint64_t pos_end = pos;
    uint32_t member_end_no = member_no;
    bool finished_end = false;
    std::vector<GzMemberEnd> ends_here;
    for (int round = 0;; ++round) {
      std::map<uint64_t, uint32_t> by_start;
      for (uint32_t s = 0; s < n_seg; ++s) by_start.emplace(h_segs[s].start_bit, s);      // (the first decode of a start bit wins; later ones are identical)
      chain.clear(); ends_here.clear();
      finished_end = false;
      std::vector<std::pair<uint64_t, uint64_t>> gaps;
      bool tentative = false, done = false;
      uint64_t p = pos;
      uint32_t mno = member_no;
      while (!done) {
        if (!last_batch && p >= territory_end) { done = true; break; }          // the next batch carries on from here
        auto it = by_start.find(p);
        if (it == by_start.end()) {
          // nothing was decoded from this exact bit (the first block of a further member, or the segment planned here started
          // at a false sync and the one before ran over it): decode [p, next known start) in the next round
          auto nx = by_start.upper_bound(p);
          uint64_t stop = nx != by_start.end() ? nx->first : (last_batch ? end_bit : std::min(territory_end + 8 * seg_bytes, limit - 4096));
          if (stop <= p || stop - p > 64 * 8 * seg_bytes) return kFallbackToHost;     // (one wave would decode a gap that long for seconds)
          gaps.emplace_back(p, stop);
          if (verbose) std::fprintf(stderr, "scfq gzdev:   batch %u: gap at bit %llu (%.1f %% of the file) up to %llu\n", k, (unsigned long long)p,
                                    100.0 * (double)p / (double)end_bit, (unsigned long long)stop);
          tentative = true;
          if (nx == by_start.end()) break;
          p = nx->first;                 // (assume the gap segment arrives exactly there; the next round's walk checks it)
          continue;
        }
        const GzSegOut& r = h_outs[it->second];
        if (r.status != kGzOk && r.status != kGzMemberEnd) {
          if (tentative) break;          // may not even be on the real path: decide after the gaps are decoded
          if (verbose) std::fprintf(stderr, "scfq gzdev: segment at bit %llu ended with status %u: host path\n", (unsigned long long)p, r.status);
          return kFallbackToHost;
        }
        if (r.end_bit <= p) return kFallbackToHost;
        chain.push_back(Entry{it->second, mno});
        if (r.status == kGzMemberEnd) {
          const uint64_t q = (r.end_bit + 7) >> 3;             // the trailer starts on the next byte boundary
          if (q + 8 > fsize) return kFallbackToHost;           // truncated trailer: gzread's error, from the host path
          GzMemberEnd me;
          me.member = mno;
          me.crc = (uint32_t)img[q] | ((uint32_t)img[q + 1] << 8) | ((uint32_t)img[q + 2] << 16) | ((uint32_t)img[q + 3] << 24);
          me.isize = (uint32_t)img[q + 4] | ((uint32_t)img[q + 5] << 8) | ((uint32_t)img[q + 6] << 16) | ((uint32_t)img[q + 7] << 24);
          ends_here.push_back(me);
          const long h = scfq_gzfast::member_header(img + q + 8, (size_t)(fsize - (q + 8)));
          if (h < 0) return kFallbackToHost;                   // a damaged further header: the host path decides
          if (h == 0) { finished_end = true; done = true; break; }   // end of file, or trailing garbage (ignored, as gzread does)
          p = (q + 8 + (uint64_t)h) * 8;
          if (++mno > 1024) return kFallbackToHost;            // (files of very many small members: the host's serial reader)
        } else {
          p = r.end_bit;
          if (p + 8 >= end_bit) return kFallbackToHost;        // the data ends inside a member: truncated file
        }
      }
      if (done && !tentative) { pos_end = p; member_end_no = mno; break; }
      if (gaps.empty() || round >= 4) {
        if (verbose) std::fprintf(stderr, "scfq gzdev: batch %u: chain not closed after %d rounds (%zu gaps): host path\n", k, round, gaps.size());
        return kFallbackToHost;
      }
      // A gap is as long as a planned segment and one wave would take a whole decode phase for it: it is cut into pieces
      // the same way the file was — its exact start, then block starts searched at equal steps inside it
      std::vector<std::pair<uint64_t, uint64_t>> pieces;       // (start, stop)
      {
        const uint32_t kSub = 8;
        std::vector<uint64_t> from;
        for (auto& gp : gaps) {
          if (by_start.count(gp.first)) continue;
          const uint64_t span = gp.second - gp.first;
          const uint32_t nparts = span >= (uint64_t)kSub * 8 * 4096 ? kSub : 1;
          for (uint32_t j = 1; j < nparts; ++j) from.push_back(gp.first + span * j / nparts);
        }
        std::vector<uint64_t> found(from.size(), ~0ull);
        if (!from.empty() && from.size() <= max_seg) {
          std::memcpy(h_gfrom, from.data(), 8 * from.size());
          HIPCHK(hipMemcpyAsync(g.d_pmeta[pp] + offp_gfrom, h_gfrom, 8 * from.size(), hipMemcpyHostToDevice, c->compute));
          hipLaunchKernelGGL(gz_sync_search, dim3((unsigned)from.size()), dim3(kSyncThreads), 0, c->compute, reinterpret_cast<const uint64_t*>(vbase), limit,
                             reinterpret_cast<const uint64_t*>(g.d_pmeta[pp] + offp_gfrom), (uint32_t)from.size(), seg_bytes * 8,
                             reinterpret_cast<uint64_t*>(g.d_pmeta[pp] + offp_gfound));
          HIPCHK(hipGetLastError());
          HIPCHK(hipMemcpyAsync(h_gfound, g.d_pmeta[pp] + offp_gfound, 8 * from.size(), hipMemcpyDeviceToHost, c->compute));
          HIPCHK(hipStreamSynchronize(c->compute));
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
          for (size_t q = 0; q < st.size(); ++q) pieces.emplace_back(st[q], q + 1 < st.size() ? st[q + 1] : gp.second);
        }
      }
      if (n_seg + pieces.size() > max_seg) return kFallbackToHost;
      const uint32_t first = n_seg;
      for (auto& gp : pieces) {
        GzSeg& sg = h_segs[n_seg];
        sg.start_bit = gp.first;
        sg.stop_bit = gp.second;
        const uint64_t cap = std::min<uint64_t>(seg_cap(gp.first, gp.second), 0x3F000000u);
        if (pool_used + kGzWindow + cap > pool_syms) return kFallbackToHost;
        sg.sym_off = pool_used;
        sg.cap = (uint32_t)cap;
        sg.reserved = 0;
        pool_used += kGzWindow + cap;
        ++n_seg;
      }
      if (n_seg == first) return kFallbackToHost;
      if (verbose) std::fprintf(stderr, "scfq gzdev:   batch %u round %d: %u gap segments\n", k, round + 1, n_seg - first);
      ++n_gap_rounds;
      HIPCHK(hipMemcpyAsync(sl.d_meta + off_segs + sizeof(GzSeg) * first, h_segs + first, sizeof(GzSeg) * (n_seg - first), hipMemcpyHostToDevice, c->compute));
      hipLaunchKernelGGL(gz_segment_decode, dim3((n_seg - first + kWavesPerWg - 1) / kWavesPerWg), dim3(64 * kWavesPerWg), kWavesPerWg * kWaveLdsBytes, c->compute,
                         vbase, copy_end_of(k), reinterpret_cast<const GzSeg*>(sl.d_meta + off_segs) + first, n_seg - first, sl.d_sym,
                         reinterpret_cast<GzSegOut*>(sl.d_meta + off_outs) + first, inflate_serial_loop());
      HIPCHK(hipGetLastError());
      HIPCHK(hipMemcpyAsync(h_outs + first, sl.d_meta + off_outs + sizeof(GzSegOut) * first, sizeof(GzSegOut) * (n_seg - first), hipMemcpyDeviceToHost, c->compute));
      HIPCHK(hipStreamSynchronize(c->compute));
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
    struct PartHere { uint32_t member; uint64_t off, len; };
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
        if (n_work + nt > max_work) return kFallbackToHost;
        for (uint32_t t = 0; t < nt; ++t) { h_we[n_work] = q; h_wt[n_work] = t; ++n_work; }
      }
      h_first[n_chains] = n_chain;
      if (n_chain) valid_end = (!finished_end && chain[n_chain - 1].member == member_end_no) ? v : 0u;
    }
    if (batch_out > out_max) return kFallbackToHost;
    walk_ms += std::chrono::duration<double, std::milli>(clk::now() - tw).count();
    if (n_chain) {
      const bool carry_in = chain[0].member == member_no && valid > 0;                                     // the first chain goes on inside a member begun earlier
      const bool carry_out = !finished_end && chain[n_chain - 1].member == member_end_no;               // the last chain's member goes on in the next batch
      // the byte in front of this batch's output is the last byte of the batch before it: parked below the buffer before that is overwritten
      if (have_prev_out) HIPCHK(hipMemcpyAsync(d_out - 1, d_out + prev_out_bytes - 1, 1, hipMemcpyDeviceToDevice, c->compute));
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
                           sl.d_sym, g.d_win, w_in, w_out, (const uint8_t*)nullptr);
      } else {
        // groups of `group` entries inside every member's chain: maps, windows in front of the groups, windows in front of the entries
        GzChain* h_gchain = reinterpret_cast<GzChain*>(g.h_pmeta[pp] + offp_gchain);
        uint32_t* h_gfirst = reinterpret_cast<uint32_t*>(g.h_pmeta[pp] + offp_gfirst);
        uint32_t* h_mfirst = reinterpret_cast<uint32_t*>(g.h_pmeta[pp] + offp_mfirst);
        uint32_t n_groups = 0;
        for (uint32_t ch = 0; ch < n_chains; ++ch) {
          h_mfirst[ch] = n_groups;
          for (uint32_t q = h_first[ch]; q < h_first[ch + 1]; q += group) {
            if (n_groups >= max_groups) return kFallbackToHost;
            h_gfirst[n_groups] = q;
            h_gchain[n_groups] = GzChain{};
            h_gchain[n_groups].sym_off = (uint64_t)n_groups * kGzWindow;       // (map g lies one window further: the form of a segment's symbols)
            h_gchain[n_groups].n_sym = kGzWindow;
            ++n_groups;
          }
        }
        h_mfirst[n_chains] = n_groups;
        h_gfirst[n_groups] = n_chain;
        HIPCHK(hipMemcpyAsync(g.d_pmeta[pp] + offp_gchain, h_gchain, sizeof(GzChain) * n_groups, hipMemcpyHostToDevice, c->compute));
        HIPCHK(hipMemcpyAsync(g.d_pmeta[pp] + offp_gfirst, h_gfirst, 4ull * (n_groups + 1), hipMemcpyHostToDevice, c->compute));
        HIPCHK(hipMemcpyAsync(g.d_pmeta[pp] + offp_mfirst, h_mfirst, 4ull * (n_chains + 1), hipMemcpyHostToDevice, c->compute));
        const uint32_t* d_gfirst = reinterpret_cast<const uint32_t*>(g.d_pmeta[pp] + offp_gfirst);
        hipLaunchKernelGGL(gz_window_maps, dim3(n_groups), dim3(1024), 0, c->compute, d_chain, d_gfirst, sl.d_sym, reinterpret_cast<uint16_t*>(g.d_maps));
        hipLaunchKernelGGL(gz_window_chain, dim3(n_chains), dim3(1024), 0, c->compute, reinterpret_cast<const GzChain*>(g.d_pmeta[pp] + offp_gchain),
                           reinterpret_cast<const uint32_t*>(g.d_pmeta[pp] + offp_mfirst), reinterpret_cast<const uint16_t*>(g.d_maps), g.d_gwin, w_in, w_out,
                           (const uint8_t*)nullptr);
        hipLaunchKernelGGL(gz_window_chain, dim3(n_groups), dim3(1024), 0, c->compute, d_chain, d_gfirst, sl.d_sym, g.d_win, (const uint8_t*)nullptr,
                           (uint8_t*)nullptr, (const uint8_t*)g.d_gwin);
      }
      HIPCHK(hipGetLastError());
      span_end(sp_chain, c->compute);
      if (carry_out) wcarry ^= 1;
      if (n_work) {
        span_begin(sp_resolve, c->compute);
        hipLaunchKernelGGL(gz_resolve, dim3((unsigned)n_work), dim3(256), 0, c->compute, reinterpret_cast<const GzChain*>(g.d_pmeta[pp] + offp_chain),
                           reinterpret_cast<const uint32_t*>(g.d_pmeta[pp] + offp_we), reinterpret_cast<const uint32_t*>(g.d_pmeta[pp] + offp_wt), sl.d_sym,
                           g.d_win, d_out, reinterpret_cast<uint32_t*>(g.d_crc));
        HIPCHK(hipGetLastError());
        span_end(sp_resolve, c->compute);
      }
      span_begin(sp_crc, c->compute);
      for (const PartHere& ph : parts_here) {
        const uint64_t nt = (ph.len + kCrcTile - 1) / kCrcTile;
        if (tiles_used + nt > crc_tiles_max) return kFallbackToHost;
        if (nt) {
          hipLaunchKernelGGL(gz_crc32_tiles, dim3((unsigned)nt), dim3(256), 0, c->compute, d_out + ph.off, ph.len, nt * kCrcTile - ph.len,
                             reinterpret_cast<uint32_t*>(g.d_crc) + 16 + tiles_used);
          HIPCHK(hipGetLastError());
        }
        parts.push_back(GzPart{ph.member, ph.len, tiles_used, nt});
        tiles_used += nt;
      }
      span_end(sp_crc, c->compute);
      if (batch_out) {
        span_begin(sp_scan, c->compute);
        rc = scan_async(c, d_out, batch_out, have_prev_out ? -2 : -1, flags & ~SCFQ_PREV_IN_MEMORY, timing);
        if (rc) return rc;
        span_end(sp_scan, c->compute);
        have_prev_out = true;
        prev_out_bytes = batch_out;
      }
    }
    HIPCHK(hipEventRecord(g.ev_post[pp], c->compute));
    for (const GzMemberEnd& me : ends_here) member_ends.push_back(me);
    total_out += batch_out;
    pos = pos_end;
    member_no = member_end_no;
    valid = valid_end;
    finished = finished_end;
    return SCFQ_OK;
  };

  // ---- the pipeline -----------------------------------------------------------------------------------------------------------
  for (uint32_t it = 0; it < nb + 2; ++it) {
    if (it < nb && (rc = stage_a(it))) return rc;
    if (it >= 1 && it - 1 < nb && (rc = stage_b(it - 1))) return rc;
    if (it >= 2) {
      if ((rc = stage_c(it - 2))) return rc;
      if (finished) break;              // (trailing garbage may leave batches behind the last member: nothing in them counts)
    }
  }
  if (!finished) return kFallbackToHost;               // the data ended inside a member

  // ---- every member's ISIZE and CRC-32 ------------------------------------------------------------------------------------------
  HIPCHK(hipMemcpyAsync(g.h_crc, g.d_crc, 4 * (16 + tiles_used), hipMemcpyDeviceToHost, c->compute));
  HIPCHK(hipStreamSynchronize(c->compute));
  {
    const uint32_t* hc = reinterpret_cast<const uint32_t*>(g.h_crc);
    if (hc[0]) return kFallbackToHost;                 // a reference before a member's start
    const uint32_t x_tile = gz_xpow8n(kCrcTile);
    size_t pi = 0;
    for (const GzMemberEnd& me : member_ends) {
      uint32_t raw = 0;
      uint64_t len = 0;
      for (; pi < parts.size() && parts[pi].member == me.member; ++pi) {
        uint32_t r = 0;
        for (uint64_t t = 0; t < parts[pi].n_tiles; ++t) r = gz_mulmod(x_tile, r) ^ hc[16 + parts[pi].tile_at + t];
        raw = gz_mulmod(gz_xpow8n(parts[pi].len), raw) ^ r;
        len += parts[pi].len;
      }
      const uint32_t crc = raw ^ gz_mulmod(gz_xpow8n(len), 0xFFFFFFFFu) ^ 0xFFFFFFFFu;
      if ((uint32_t)len != me.isize || crc != me.crc) {
        if (verbose) std::fprintf(stderr, "scfq gzdev: member %u: %llu bytes, CRC-32 %08x; its trailer says %u, %08x: host path\n", me.member,
                                  (unsigned long long)len, crc, me.isize, me.crc);
        return kFallbackToHost;
      }
    }
    if (pi != parts.size()) return kFallbackToHost;
  }
  c->timing.host_fill_ms += fill_ms;
  c->timing.ingest_wall_ms += std::chrono::duration<double, std::milli>(clk::now() - t_begin).count();
  if (verbose) {
    for (hipStream_t st : {c->copy, g.s_search, g.s_decode[0], g.s_decode[1]}) (void)hipStreamSynchronize(st);
    auto sum = [](const std::vector<Span>& v) { double t = 0; for (const Span& s : v) { float ms = 0; if (s.a && s.b && hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) t += ms; } return t; };
    std::fprintf(stderr, "scfq gzdev: %-28s %8.2f ms\n", "copy to HBM", sum(sp_copy));
    std::fprintf(stderr, "scfq gzdev: %-28s %8.2f ms\n", "block-start search", sum(sp_search));
    std::fprintf(stderr, "scfq gzdev: %-28s %8.2f ms\n", "segment decode", sum(sp_decode));
    std::fprintf(stderr, "scfq gzdev: %-28s %8.2f ms\n", "chain walk", walk_ms);
    std::fprintf(stderr, "scfq gzdev: %-28s %8.2f ms\n", "window chain", sum(sp_chain));
    std::fprintf(stderr, "scfq gzdev: %-28s %8.2f ms\n", "resolve", sum(sp_resolve));
    std::fprintf(stderr, "scfq gzdev: %-28s %8.2f ms\n", "crc tiles", sum(sp_crc));
    std::fprintf(stderr, "scfq gzdev: %-28s %8.2f ms\n", "scan", sum(sp_scan));
    std::fprintf(stderr, "scfq gzdev: host: pinned-buffer waits %.1f, memcpy %.1f, copy enqueue %.1f, search waits %.1f, decode waits %.1f, post waits %.1f ms\n", h_evsync_ms,
                 h_memcpy_ms, h_enqueue_ms, h_search_wait_ms, h_dec_wait_ms, h_post_wait_ms);
    std::fprintf(stderr, "scfq gzdev: %-28s %8.2f ms\n", "wall", std::chrono::duration<double, std::milli>(clk::now() - t_begin).count());
    std::fprintf(stderr, "scfq gzdev: %u batch(es), %u segments planned, %u decoded (%u gap rounds), %u on the chain, %zu member(s), %llu bytes inflated\n", nb,
                 n_planned_total, n_decoded_total, n_gap_rounds, n_chain_total, member_ends.size(), (unsigned long long)total_out);
  }
  return SCFQ_OK;
}

// One file at a time per device (gz_shared); every stream of the path is idle when this returns, whatever the outcome — the
// scan of the last batch included, which reads the shared output buffer: the buffers belong to the next caller.
int ingest_gz_device(Ctx* c, const uint8_t* img, uint64_t fsize, uint32_t flags, bool timing) {
  GzShared& gs = gz_shared(c->dev);
  std::lock_guard<std::mutex> lk(gs.mu);
  const int rc = ingest_gz_device_batches(c, img, fsize, flags, timing);
  GzDevBuffers& g = gs.buf;
  if (c->copy) (void)hipStreamSynchronize(c->copy);
  if (g.s_search) (void)hipStreamSynchronize(g.s_search);
  for (int b = 0; b < 2; ++b) if (g.s_decode[b]) (void)hipStreamSynchronize(g.s_decode[b]);
  if (c->compute) (void)hipStreamSynchronize(c->compute);
  return rc;
}
