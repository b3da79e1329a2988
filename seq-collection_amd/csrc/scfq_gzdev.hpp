// scfq_gzdev.hpp — host side of the device gzip inflate (kernels: gz_inflate_kernels.hpp).  Included by scfq_api.hip
// inside its anonymous namespace (uses Ctx, HIPCHK, scan_async, parallel_pieces, kFallbackToHost).
//
// ingest_gz_device(): COMPRESSED file -> pinned ring -> HBM, then on the device: block-start search, segment decode to
// 16-bit symbols, window chain, resolve, CRC-32 tiles; on the host in between: the chain walk.  A decoded segment is only
// believed when the walk reaches its start bit EXACTLY — from the member's first block, through every segment's end bit,
// over member trailers and headers — so a false sync, a corrupt block or a truncated file can never contribute a byte:
// the walk stops there, gaps it can prove (a false sync skipped, the first block of a further member) are decoded in a
// further round, and anything else hands the whole file to the host readers (kFallbackToHost), which reproduce gzread's
// behaviour for it byte by byte (partial output, SCFQ_EGZ).  CRC-32 and ISIZE of every member are checked before the
// inflated bytes are scanned.

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
  if (g->d_comp) (void)hipFree(g->d_comp);
  if (g->d_sym) (void)hipFree(g->d_sym);
  if (g->d_out) (void)hipFree(g->d_out);
  if (g->d_win) (void)hipFree(g->d_win);
  if (g->d_meta) (void)hipFree(g->d_meta);
  if (g->h_meta) (void)hipHostFree(g->h_meta);
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

struct GzMember {
  uint64_t out_off = 0, out_len = 0;     // in the inflated stream
  uint32_t crc = 0, isize = 0;           // trailer
  uint32_t first_chain = 0;              // index of its first chain entry
};

int ingest_gz_device(Ctx* c, const uint8_t* img, uint64_t fsize, uint32_t flags, bool timing) {
  using namespace scfq_dinflate;
  using clk = std::chrono::steady_clock;
  const auto t_begin = clk::now();
  static const bool verbose = std::getenv("SCFQ_VERBOSE") != nullptr;
  auto lap_t = clk::now();
  auto lap = [&](const char* what) {
    if (!verbose) return;
    (void)hipStreamSynchronize(c->copy);
    (void)hipStreamSynchronize(c->compute);
    const auto now = clk::now();
    std::fprintf(stderr, "scfq gzdev: %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - lap_t).count());
    lap_t = now;
  };
  const long h0 = scfq_gzfast::member_header(img, (size_t)fsize);
  if (h0 <= 0 || fsize < 64) return kFallbackToHost;
  GzDevBuffers& g = c->gz;
  int rc;

  // ---- compressed bytes -> HBM (pinned double buffer, copy stream), zero padding behind them -------------------------
  const uint64_t comp_pad = 256;
  if ((rc = gz_grow(&g.d_comp, &g.comp_cap, fsize + comp_pad))) return rc;
  {
    const uint64_t chunk = 64ull << 20;
    rc = ensure_staging(c, chunk, true);
    if (rc) return rc;
    for (uint64_t off = 0, it = 0; off < fsize; off += chunk, ++it) {
      const int b = (int)(it & 1);
      const uint64_t len = std::min(chunk, fsize - off);
      if (it >= 2) HIPCHK(hipEventSynchronize(c->ev_copied[b]));
      const uint8_t* src = img + off;
      parallel_pieces(len, [&](uint64_t o, uint64_t l) { std::memcpy(c->h_pin[b] + o, src + o, l); return 0; });
      HIPCHK(hipMemcpyAsync(g.d_comp + off, c->h_pin[b], (size_t)len, hipMemcpyHostToDevice, c->copy));
      HIPCHK(hipEventRecord(c->ev_copied[b], c->copy));
    }
    HIPCHK(hipMemsetAsync(g.d_comp + fsize, 0, comp_pad, c->copy));
    HIPCHK(hipEventRecord(c->ev_copied[0], c->copy));
    HIPCHK(hipStreamWaitEvent(c->compute, c->ev_copied[0], 0));
    c->timing.h2d_bytes += fsize;
  }
  const double fill_ms = std::chrono::duration<double, std::milli>(clk::now() - t_begin).count();
  lap("compressed bytes to HBM");

  // ---- plan: segment s starts at the first block header found at or after data + s * seg_bytes ------------------------
  const uint64_t data0 = (uint64_t)h0;
  const uint64_t comp = fsize - data0;
  static const uint64_t seg_kb = (uint64_t)std::max(32, env_int("SCFQ_GZ_DEVICE_SEGMENT_KB", 128));
  static const uint64_t max_segs = (uint64_t)std::max(2, env_int("SCFQ_GZ_DEVICE_MAX_SEGMENTS", 8192));
  static const uint64_t ratio = (uint64_t)std::max(2, env_int("SCFQ_GZ_DEVICE_MAX_RATIO", 7));   // output symbols a segment may produce per compressed byte
  uint64_t seg_bytes = std::max<uint64_t>(seg_kb << 10, (comp + max_segs - 1) / max_segs);
  seg_bytes = (seg_bytes + 4095) & ~4095ull;
  const uint32_t n_plan = (uint32_t)std::max<uint64_t>(1, (comp + seg_bytes - 1) / seg_bytes);
  const uint32_t spare = 64 + n_plan / 16;                               // room for gap segments of later rounds
  const uint32_t max_seg = n_plan + spare;
  // output room of a segment: `ratio` symbols per compressed byte it spans + 256 Ki (it runs on to the end of a block), behind its 32768 markers; the pool holds
  // every planned segment plus `spare` gap segments of up to four slots each
  auto seg_cap = [&](uint64_t start_bit, uint64_t stop_bit) { return (((stop_bit - start_bit + 7) / 8) * ratio + 262144 + 7) & ~7ull; };
  const uint64_t pool_syms = comp * ratio + (uint64_t)max_seg * (kGzWindow + 262144 + 8) + (uint64_t)spare * 4 * seg_bytes * ratio;
  uint64_t pool_used = 0;
  // meta layout (device and pinned mirror, same offsets)
  const uint64_t off_from = 0, off_found = off_from + 8ull * max_seg, off_segs = off_found + 8ull * max_seg,
                 off_outs = off_segs + sizeof(GzSeg) * max_seg, off_status = off_outs + sizeof(GzSegOut) * max_seg, meta_fixed = off_status + 64;
  if ((rc = gz_grow(&g.d_meta, &g.meta_cap, meta_fixed)) || (rc = gz_grow(&g.h_meta, &g.hmeta_cap, meta_fixed, true))) return rc;
  if ((rc = gz_grow(&g.d_sym, &g.sym_cap, 2ull * pool_syms))) return rc;
  uint64_t* h_from = reinterpret_cast<uint64_t*>(g.h_meta + off_from);
  uint64_t* h_found = reinterpret_cast<uint64_t*>(g.h_meta + off_found);
  GzSeg* h_segs = reinterpret_cast<GzSeg*>(g.h_meta + off_segs);
  GzSegOut* h_outs = reinterpret_cast<GzSegOut*>(g.h_meta + off_outs);
  const uint64_t end_bit = fsize * 8;
  for (uint32_t s = 0; s < n_plan; ++s) h_from[s] = (data0 + (uint64_t)s * seg_bytes) * 8;
  if (n_plan > 1) {
    HIPCHK(hipMemcpyAsync(g.d_meta + off_from, h_from, 8ull * n_plan, hipMemcpyHostToDevice, c->compute));
    hipLaunchKernelGGL(gz_sync_search, dim3(n_plan - 1), dim3(kSyncThreads), 0, c->compute, reinterpret_cast<const uint64_t*>(g.d_comp), end_bit,
                       reinterpret_cast<const uint64_t*>(g.d_meta + off_from) + 1, n_plan - 1, seg_bytes * 8,
                       reinterpret_cast<uint64_t*>(g.d_meta + off_found) + 1);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(h_found + 1, g.d_meta + off_found + 8, 8ull * (n_plan - 1), hipMemcpyDeviceToHost, c->compute));
    HIPCHK(hipStreamSynchronize(c->compute));
  }
  lap("block-start search");
  if (verbose && n_plan > 1) {
    uint64_t sum = 0, mx = 0, none = 0;
    for (uint32_t s = 1; s < n_plan; ++s) { if (h_found[s] == ~0ull) { ++none; continue; } const uint64_t d = h_found[s] - h_from[s]; sum += d; mx = std::max(mx, d); }
    std::fprintf(stderr, "scfq gzdev:   block starts: mean distance %.0f bits, max %llu, none found for %llu of %u\n", (double)sum / std::max<uint64_t>(1, n_plan - 1 - none),
                 (unsigned long long)mx, (unsigned long long)none, n_plan - 1);
  }
  // segments in stream order: the exact start, then every found start (strictly increasing)
  std::vector<uint64_t> starts;
  starts.reserve(n_plan);
  starts.push_back(data0 * 8);
  for (uint32_t s = 1; s < n_plan; ++s)
    if (h_found[s] != ~0ull && h_found[s] > starts.back() && h_found[s] + 64 < end_bit) starts.push_back(h_found[s]);
  uint32_t n_seg = (uint32_t)starts.size();
  for (uint32_t s = 0; s < n_seg; ++s) {
    h_segs[s].start_bit = starts[s];
    h_segs[s].stop_bit = (s + 1 < n_seg) ? starts[s + 1] : end_bit;
    const uint64_t cap = std::min<uint64_t>(seg_cap(h_segs[s].start_bit, h_segs[s].stop_bit), 0x7F000000u);
    h_segs[s].sym_off = pool_used;
    h_segs[s].cap = (uint32_t)cap;       // better compression than `ratio`: overflow status, and the file goes to the host path
    h_segs[s].reserved = 0;
    pool_used += kGzWindow + cap;
  }
  if (pool_used > pool_syms) return kFallbackToHost;
  auto decode = [&](uint32_t first, uint32_t count) -> int {
    HIPCHK(hipMemcpyAsync(g.d_meta + off_segs + sizeof(GzSeg) * first, h_segs + first, sizeof(GzSeg) * count, hipMemcpyHostToDevice, c->compute));
    hipLaunchKernelGGL(gz_segment_decode, dim3((count + kWavesPerWg - 1) / kWavesPerWg), dim3(64 * kWavesPerWg), kWavesPerWg * kWaveLdsBytes, c->compute,
                       g.d_comp, fsize, reinterpret_cast<const GzSeg*>(g.d_meta + off_segs) + first, count, g.d_sym,
                       reinterpret_cast<GzSegOut*>(g.d_meta + off_outs) + first);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(h_outs + first, g.d_meta + off_outs + sizeof(GzSegOut) * first, sizeof(GzSegOut) * count, hipMemcpyDeviceToHost, c->compute));
    HIPCHK(hipStreamSynchronize(c->compute));
    return SCFQ_OK;
  };
  if ((rc = decode(0, n_seg))) return rc;
  lap("segment decode");

  // ---- the chain walk --------------------------------------------------------------------------------------------------
  std::vector<uint32_t> chain;           // segment indices in stream order
  std::vector<GzMember> members;
  std::vector<uint32_t> chain_member;
  for (int round = 0;; ++round) {
    std::map<uint64_t, uint32_t> by_start;
    for (uint32_t s = 0; s < n_seg; ++s) by_start.emplace(h_segs[s].start_bit, s);      // (the first decode of a start bit wins; later ones are identical)
    chain.clear(); members.clear(); chain_member.clear();
    std::vector<std::pair<uint64_t, uint64_t>> gaps;
    bool tentative = false, finished = false;
    uint64_t pos = data0 * 8, out_off = 0;
    GzMember m;
    m.first_chain = 0;
    while (!finished) {
      auto it = by_start.find(pos);
      if (it == by_start.end()) {
        // nothing was decoded from this exact bit (the first block of a further member, or the segment planned here started
        // at a false sync and the one before ran over it): decode [pos, next known start) in the next round
        auto nx = by_start.upper_bound(pos);
        const uint64_t stop = nx != by_start.end() ? nx->first : end_bit;
        gaps.emplace_back(pos, stop);
        if (verbose) {
          auto pv = by_start.lower_bound(pos);
          const uint64_t prev_start = pv != by_start.begin() ? std::prev(pv)->first : 0;
          std::fprintf(stderr, "scfq gzdev:   gap at bit %llu (%.1f %% of the file) up to %llu; the start before it is %llu bits back\n", (unsigned long long)pos,
                       100.0 * (double)pos / (double)end_bit, (unsigned long long)stop, (unsigned long long)(pos - prev_start));
        }
        tentative = true;
        if (nx == by_start.end()) break;
        pos = nx->first;                 // (assume the gap segment arrives exactly there; the next round's walk checks it)
        continue;
      }
      const GzSegOut& r = h_outs[it->second];
      if (r.status != kGzOk && r.status != kGzMemberEnd) {
        if (tentative) break;            // may not even be on the real path: decide after the gaps are decoded
        if (verbose) std::fprintf(stderr, "scfq gzdev: segment at bit %llu ended with status %u: host path\n", (unsigned long long)pos, r.status);
        return kFallbackToHost;
      }
      if (r.end_bit <= pos) return kFallbackToHost;
      chain.push_back(it->second);
      chain_member.push_back((uint32_t)members.size());
      out_off += r.n_sym;
      if (r.status == kGzMemberEnd) {
        const uint64_t q = (r.end_bit + 7) >> 3;             // the trailer starts on the next byte boundary
        if (q + 8 > fsize) return kFallbackToHost;           // truncated trailer: gzread's error, from the host path
        m.crc = (uint32_t)img[q] | ((uint32_t)img[q + 1] << 8) | ((uint32_t)img[q + 2] << 16) | ((uint32_t)img[q + 3] << 24);
        m.isize = (uint32_t)img[q + 4] | ((uint32_t)img[q + 5] << 8) | ((uint32_t)img[q + 6] << 16) | ((uint32_t)img[q + 7] << 24);
        m.out_len = out_off - m.out_off;
        members.push_back(m);
        const long h = scfq_gzfast::member_header(img + q + 8, (size_t)(fsize - (q + 8)));
        if (h < 0) return kFallbackToHost;                   // a damaged further header: the host path decides
        if (h == 0) { finished = true; break; }              // end of file, or trailing garbage (ignored, as gzread does)
        pos = (q + 8 + (uint64_t)h) * 8;
        m = GzMember{};
        m.out_off = out_off;
        m.first_chain = (uint32_t)chain.size();
      } else {
        pos = r.end_bit;
        if (pos + 8 >= end_bit) return kFallbackToHost;      // the data ends inside a member: truncated file
      }
    }
    if (finished && !tentative) break;
    if (gaps.empty() || round >= 4 || n_seg + gaps.size() > max_seg || members.size() > 4096) {
      if (verbose) std::fprintf(stderr, "scfq gzdev: chain not closed after %d rounds (%zu gaps): host path\n", round, gaps.size());
      return kFallbackToHost;
    }
    const uint32_t first = n_seg;
    // A gap is as long as a planned segment and one wave would take a whole decode phase for it: it is cut into pieces
    // the same way the file was — its exact start, then block starts searched at equal steps inside it
    std::vector<std::pair<uint64_t, uint64_t>> pieces;       // (start, stop)
    {
      const uint32_t kSub = 8;
      std::vector<uint64_t> from, owner_stop;
      for (auto& gp : gaps) {
        if (by_start.count(gp.first)) continue;
        const uint64_t span = gp.second - gp.first;
        const uint32_t parts = span >= (uint64_t)kSub * 8 * 4096 ? kSub : 1;
        for (uint32_t j = 1; j < parts; ++j) { from.push_back(gp.first + span * j / parts); owner_stop.push_back(gp.second); }
      }
      std::vector<uint64_t> found(from.size(), ~0ull);
      if (!from.empty() && from.size() <= max_seg) {
        std::memcpy(h_from, from.data(), 8 * from.size());
        HIPCHK(hipMemcpyAsync(g.d_meta + off_from, h_from, 8 * from.size(), hipMemcpyHostToDevice, c->compute));
        hipLaunchKernelGGL(gz_sync_search, dim3((unsigned)from.size()), dim3(kSyncThreads), 0, c->compute, reinterpret_cast<const uint64_t*>(g.d_comp), end_bit,
                           reinterpret_cast<const uint64_t*>(g.d_meta + off_from), (uint32_t)from.size(), seg_bytes * 8,
                           reinterpret_cast<uint64_t*>(g.d_meta + off_found));
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(h_found, g.d_meta + off_found, 8 * from.size(), hipMemcpyDeviceToHost, c->compute));
        HIPCHK(hipStreamSynchronize(c->compute));
        for (size_t k = 0; k < from.size(); ++k) found[k] = h_found[k];
      }
      size_t fk = 0;
      for (auto& gp : gaps) {
        if (by_start.count(gp.first)) continue;
        const uint64_t span = gp.second - gp.first;
        const uint32_t parts = span >= (uint64_t)kSub * 8 * 4096 ? kSub : 1;
        std::vector<uint64_t> st{gp.first};
        for (uint32_t j = 1; j < parts; ++j, ++fk)
          if (fk < found.size() && found[fk] != ~0ull && found[fk] > st.back() && found[fk] < gp.second && !by_start.count(found[fk])) st.push_back(found[fk]);
        for (size_t k = 0; k < st.size(); ++k) pieces.emplace_back(st[k], k + 1 < st.size() ? st[k + 1] : gp.second);
      }
    }
    if (n_seg + pieces.size() > max_seg) return kFallbackToHost;
    for (auto& gp : pieces) {
      GzSeg& sg = h_segs[n_seg];
      sg.start_bit = gp.first;
      sg.stop_bit = gp.second;
      const uint64_t cap = std::min<uint64_t>(seg_cap(gp.first, gp.second), 0x7F000000u);
      if (pool_used + kGzWindow + cap > pool_syms) return kFallbackToHost;
      sg.sym_off = pool_used;
      sg.cap = (uint32_t)cap;
      sg.reserved = 0;
      pool_used += kGzWindow + cap;
      ++n_seg;
    }
    if (n_seg == first) return kFallbackToHost;
    if (verbose) std::fprintf(stderr, "scfq gzdev: round %d: %u gap segments\n", round + 1, n_seg - first);
    if ((rc = decode(first, n_seg - first))) return rc;
  }
  if (members.empty() || members.size() > 1024) return kFallbackToHost;     // (files of very many small members: the host's serial reader)
  const uint64_t total = members.back().out_off + members.back().out_len;
  for (const GzMember& mm : members)
    if ((uint32_t)mm.out_len != mm.isize) return kFallbackToHost;            // ISIZE mismatch: corrupt; the host path reports it
  lap("chain walk");

  // ---- windows, bytes, CRC --------------------------------------------------------------------------------------------
  const uint32_t n_chain = (uint32_t)chain.size();
  const uint32_t n_mem = (uint32_t)members.size();
  std::vector<uint32_t> work_entry, work_tile;
  uint64_t n_tiles_crc = 0;
  for (const GzMember& mm : members) n_tiles_crc += (mm.out_len + kCrcTile - 1) / kCrcTile;
  for (uint32_t k = 0; k < n_chain; ++k) {
    const uint32_t nt = (h_outs[chain[k]].n_sym + kResolveTile - 1) / kResolveTile;
    for (uint32_t t = 0; t < nt; ++t) { work_entry.push_back(k); work_tile.push_back(t); }
  }
  const uint64_t off_chain = meta_fixed, off_first = off_chain + sizeof(GzChain) * n_chain, off_we = off_first + 4ull * (n_mem + 1),
                 off_wt = off_we + 4ull * work_entry.size(), off_crc = off_wt + 4ull * work_tile.size(), meta_all = off_crc + 4ull * n_tiles_crc + 64;
  {
    // (growing the meta buffers drops their contents: everything the device still needs from them is re-sent below)
    const bool regrow = g.meta_cap < meta_all || g.hmeta_cap < meta_all;
    std::vector<uint8_t> keep;
    if (regrow) { keep.assign(g.h_meta, g.h_meta + meta_fixed); }
    if ((rc = gz_grow(&g.d_meta, &g.meta_cap, meta_all)) || (rc = gz_grow(&g.h_meta, &g.hmeta_cap, meta_all, true))) return rc;
    if (regrow) std::memcpy(g.h_meta, keep.data(), keep.size());
    h_outs = reinterpret_cast<GzSegOut*>(g.h_meta + off_outs);
    h_segs = reinterpret_cast<GzSeg*>(g.h_meta + off_segs);
  }
  GzChain* h_chain = reinterpret_cast<GzChain*>(g.h_meta + off_chain);
  uint32_t* h_first = reinterpret_cast<uint32_t*>(g.h_meta + off_first);
  {
    uint64_t oo = 0;
    uint32_t valid = 0, cur_m = 0xFFFFFFFFu;
    for (uint32_t k = 0; k < n_chain; ++k) {
      if (chain_member[k] != cur_m) { cur_m = chain_member[k]; valid = 0; h_first[cur_m] = k; }
      const uint32_t s = chain[k];
      h_chain[k].sym_off = h_segs[s].sym_off;
      h_chain[k].out_off = oo;
      h_chain[k].n_sym = h_outs[s].n_sym;
      h_chain[k].valid_before = valid;
      h_chain[k].chain_id = cur_m;
      h_chain[k].reserved = 0;
      oo += h_outs[s].n_sym;
      valid = (uint32_t)std::min<uint64_t>(kGzWindow, (uint64_t)valid + h_outs[s].n_sym);
    }
    h_first[n_mem] = n_chain;
  }
  std::memcpy(g.h_meta + off_we, work_entry.data(), 4ull * work_entry.size());
  std::memcpy(g.h_meta + off_wt, work_tile.data(), 4ull * work_tile.size());
  if ((rc = gz_grow(&g.d_win, &g.win_cap, (uint64_t)kGzWindow * n_chain))) return rc;
  if ((rc = gz_grow(&g.d_out, &g.out_cap, total + 2 * kStagePad))) return rc;
  uint8_t* d_out = g.d_out + kStagePad;
  HIPCHK(hipMemcpyAsync(g.d_meta + off_chain, g.h_meta + off_chain, off_crc - off_chain, hipMemcpyHostToDevice, c->compute));
  HIPCHK(hipMemsetAsync(g.d_meta + off_status, 0, 64, c->compute));
  hipLaunchKernelGGL(gz_window_chain, dim3(n_mem), dim3(1024), 0, c->compute, reinterpret_cast<const GzChain*>(g.d_meta + off_chain),
                     reinterpret_cast<const uint32_t*>(g.d_meta + off_first), g.d_sym, g.d_win);
  HIPCHK(hipGetLastError());
  lap("window chain");
  if (!work_entry.empty()) {
    hipLaunchKernelGGL(gz_resolve, dim3((unsigned)work_entry.size()), dim3(256), 0, c->compute, reinterpret_cast<const GzChain*>(g.d_meta + off_chain),
                       reinterpret_cast<const uint32_t*>(g.d_meta + off_we), reinterpret_cast<const uint32_t*>(g.d_meta + off_wt), g.d_sym, g.d_win, d_out,
                       reinterpret_cast<uint32_t*>(g.d_meta + off_status));
    HIPCHK(hipGetLastError());
  }
  lap("resolve");
  {
    uint64_t tile_at = 0;
    for (const GzMember& mm : members) {
      const uint64_t nt = (mm.out_len + kCrcTile - 1) / kCrcTile;
      if (nt) {
        const uint64_t pad = nt * kCrcTile - mm.out_len;
        hipLaunchKernelGGL(gz_crc32_tiles, dim3((unsigned)nt), dim3(256), 0, c->compute, d_out + mm.out_off, mm.out_len, pad,
                           reinterpret_cast<uint32_t*>(g.d_meta + off_crc) + tile_at);
        HIPCHK(hipGetLastError());
      }
      tile_at += nt;
    }
  }
  HIPCHK(hipMemcpyAsync(g.h_meta + off_crc, g.d_meta + off_crc, 4ull * n_tiles_crc, hipMemcpyDeviceToHost, c->compute));
  HIPCHK(hipMemcpyAsync(g.h_meta + off_status, g.d_meta + off_status, 4, hipMemcpyDeviceToHost, c->compute));
  HIPCHK(hipStreamSynchronize(c->compute));
  lap("crc tiles");
  if (*reinterpret_cast<const uint32_t*>(g.h_meta + off_status)) return kFallbackToHost;   // a reference before a member's start
  {
    const uint32_t x_tile = gz_xpow8n(kCrcTile);
    const uint32_t* tc = reinterpret_cast<const uint32_t*>(g.h_meta + off_crc);
    uint64_t tile_at = 0;
    for (const GzMember& mm : members) {
      const uint64_t nt = (mm.out_len + kCrcTile - 1) / kCrcTile;
      uint32_t r = 0;
      for (uint64_t t = 0; t < nt; ++t) r = gz_mulmod(x_tile, r) ^ tc[tile_at + t];
      const uint32_t crc = r ^ gz_mulmod(gz_xpow8n(mm.out_len), 0xFFFFFFFFu) ^ 0xFFFFFFFFu;
      if (crc != mm.crc) {
        if (verbose) std::fprintf(stderr, "scfq gzdev: CRC-32 of a member is %08x, its trailer says %08x: host path\n", crc, mm.crc);
        return kFallbackToHost;
      }
      tile_at += nt;
    }
  }
  // ---- the inflated stream is in HBM and proven: scan it ------------------------------------------------------------------
  if (total) {
    rc = scan_async(c, d_out, total, -1, flags & ~SCFQ_PREV_IN_MEMORY, timing);
    if (rc) return rc;
  }
  c->timing.host_fill_ms += fill_ms;
  c->timing.ingest_wall_ms += std::chrono::duration<double, std::milli>(clk::now() - t_begin).count();
  if (verbose) std::fprintf(stderr, "scfq gzdev: %u segments planned, %u decoded, %u on the chain, %u member(s), %llu bytes inflated\n", n_plan, n_seg, n_chain,
                            n_mem, (unsigned long long)total);
  return SCFQ_OK;
}
