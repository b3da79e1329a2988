// scfq_sources.hpp — host-side byte sources of the chunked ingest (included by scfq_api.hip inside its anonymous
// namespace): plain pread, memory, serial zlib gzread, the library's own gzip reader, block-parallel BGZF inflate.
// Every source yields exactly the bytes zlib's gzread yields for the same file (gzip_stream.nim:16-17 semantics).
// No device code here; the device-side BGZF inflate is bgzf_inflate_kernel.hpp + ingest_bgzf_device() in scfq_api.hip.
#pragma once

// A source of bytes for the chunked ingest loop: fills dst with up to cap bytes, returns count,
// 0 at end, negative SCFQ_* on error.
int env_int(const char* name, int dflt);

struct Source {
  virtual ~Source() {}
  virtual int64_t fill(uint8_t* dst, uint64_t cap) = 0;
};

// Host-side copies into the pinned ring are split over a few threads: one core moves ~10-15 GB/s out of the page
// cache / pageable memory, well below what PCIe Gen5 x16 takes (SCFQ_IO_THREADS, default 8).
int io_threads() {
  static const int n = std::max(1, std::min(64, env_int("SCFQ_IO_THREADS", 8)));
  return n;
}

template <typename F>
int parallel_pieces(uint64_t total, F&& piece /* int(uint64_t off, uint64_t len) */, int max_threads = 0, uint64_t kMinPiece = 4ull << 20) {
  int nt = (int)std::min<uint64_t>((uint64_t)(max_threads > 0 ? max_threads : io_threads()), (total + kMinPiece - 1) / kMinPiece);
  if (nt <= 1) return piece(0, total);
  std::vector<int> rcs(nt, 0);
  std::vector<std::thread> th;
  for (int t = 0; t < nt; ++t) {
    const uint64_t lo = total * (uint64_t)t / nt, hi = total * (uint64_t)(t + 1) / nt;
    th.emplace_back([&, t, lo, hi] { rcs[t] = piece(lo, hi - lo); });
  }
  for (auto& x : th) x.join();
  for (int r : rcs) if (r) return r;
  return 0;
}

// Compressed bytes of a file into a pinned buffer, for the device inflate paths (nothing else for the host's cores to do meanwhile:
// SCFQ_COPY_THREADS, default 12).  With the file's descriptor: pread — the kernel copies out of the page cache without a page fault per
// 4 KiB, which is what a memcpy out of the mapping pays (r3: 30 GB/s through the ring with 8 memcpy threads, against 42 GB/s for the
// plain-file path's preads); without one (a caller that only has the bytes): memcpy.
int copy_threads() {
  static const int n = std::max(1, std::min(64, env_int("SCFQ_COPY_THREADS", 12)));
  return n;
}
struct FileBytes {
  const uint8_t* img = nullptr;      // the bytes, mapped
  int fd = -1;                       // ... and, when there is one, the file they are mapped from,
  uint64_t fd_off = 0;               // img[0] being the byte at this offset of it
};
inline int copy_file_bytes(const FileBytes& fb, uint64_t off, uint8_t* dst, uint64_t len, bool always_pread = false) {
  // (pread for the big pieces of big files only — 128 MiB pieces: 49 ms per 2.43 GB against 80 with memcpy; with the 16 MiB pieces of
  // files up to 1 GiB memcpy out of the mapping is the faster by 3 - 8 %: profiles/r04/copy_variants.jsonl.  SCFQ_COPY_PREAD = 0 / 2:
  // never / always)
  static const int pread_mode = env_int("SCFQ_COPY_PREAD", 1);
  const bool use_pread = pread_mode == 2 || (pread_mode == 1 && (always_pread || len >= (64ull << 20)));
  return parallel_pieces(len, [&](uint64_t o, uint64_t l) {
    if (fb.fd >= 0 && use_pread) {
      uint64_t got = 0;
      while (got < l) {
        const ssize_t r = pread(fb.fd, dst + o + got, (size_t)(l - got), (off_t)(fb.fd_off + off + o + got));
        if (r <= 0) break;          // (the file shrank under the mapping: the mapped bytes decide, below)
        got += (uint64_t)r;
      }
      if (got == l) return 0;
    }
    std::memcpy(dst + o, fb.img + off + o, l);
    return 0;
  }, copy_threads(), 1ull << 20);
}

struct MemSource : Source {
  const uint8_t* p; uint64_t n, off = 0;
  MemSource(const uint8_t* p_, uint64_t n_) : p(p_), n(n_) {}
  int64_t fill(uint8_t* dst, uint64_t cap) override {
    const uint64_t k = std::min(cap, n - off);
    const uint8_t* src = p + off;
    parallel_pieces(k, [&](uint64_t o, uint64_t len) { std::memcpy(dst + o, src + o, len); return 0; });
    off += k;
    return (int64_t)k;
  }
};

struct FdSource : Source {
  int fd; uint64_t off, end;
  FdSource(int fd_, uint64_t off_, uint64_t end_) : fd(fd_), off(off_), end(end_) {}
  int64_t fill(uint8_t* dst, uint64_t cap) override {
    const uint64_t want = std::min(cap, end - off);
    const uint64_t base = off;
    std::vector<uint64_t> short_at;   // a piece that hit EOF early (file shrank): report the contiguous prefix
    std::mutex mu;
    const int rc = parallel_pieces(want, [&](uint64_t o, uint64_t len) {
      uint64_t got = 0;
      while (got < len) {
        ssize_t r = pread(fd, dst + o + got, len - got, (off_t)(base + o + got));
        if (r < 0) return (int)SCFQ_EIO;
        if (r == 0) { std::lock_guard<std::mutex> lk(mu); short_at.push_back(o + got); break; }
        got += (uint64_t)r;
      }
      return 0;
    });
    if (rc) return rc;
    uint64_t got = want;
    for (uint64_t v : short_at) got = std::min(got, v);
    off += got;
    return (int64_t)got;
  }
};

struct GzSource : Source {
  gzFile f;
  explicit GzSource(gzFile f_) : f(f_) {}
  int64_t fill(uint8_t* dst, uint64_t cap) override {
    uint64_t got = 0;
    while (got < cap) {
      unsigned want = (unsigned)std::min<uint64_t>(cap - got, 1u << 30);
      int r = gzread(f, dst + got, want);   // gzip_stream.nim:16-17 fsReadData == gzread
      if (r < 0) return SCFQ_EGZ;
      if (r == 0) break;
      got += (unsigned)r;
    }
    return (int64_t)got;
  }
};

// Regular gzip files: the library's own inflate (scfq_inflate.hpp) on a decoder thread, CRC + copy on this one; same
// bytes and the same accept / reject decisions as gzread.  SCFQ_INFLATE=zlib keeps everything on zlib.
struct FastGzSource : Source {
  scfq_gzfast::Stream st;
  int64_t fill(uint8_t* dst, uint64_t cap) override {
    const int64_t r = st.next_chunk(dst, cap);
    return r < 0 ? (int64_t)SCFQ_EGZ : r;
  }
};

// the same stream inflated by many threads (scfq_pgz.hpp); files of 8 MiB and more, SCFQ_PGZ=0 keeps the serial reader
struct ParallelGzSource : Source {
  scfq_pgz::Stream st;
  int64_t fill(uint8_t* dst, uint64_t cap) override {
    const int64_t r = st.next_chunk(dst, cap);
    return r < 0 ? (int64_t)SCFQ_EGZ : r;
  }
};

bool use_own_inflate() {
  static const bool v = [] { const char* e = std::getenv("SCFQ_INFLATE"); return !(e && e[0] == 'z'); }();
  return v;
}

// the source for a ".gz" path that is not BGZF: own decoder when the file is a mappable regular file that starts with a
// gzip member and the staging chunk is big enough for the decoder's write slack, zlib otherwise (transparent
// pass-through of non-gzip bytes, FIFOs, tiny chunks)
std::unique_ptr<Source> open_gz_source(const char* path, uint64_t chunk, gzFile* gz_out) {
  *gz_out = nullptr;
  if (use_own_inflate() && chunk >= (1u << 16)) {
    static const bool pgz = [] { const char* e = std::getenv("SCFQ_PGZ"); return !(e && e[0] == '0'); }();
    if (pgz) {
      auto pz = std::make_unique<ParallelGzSource>();
      if (pz->st.open(path)) return pz;
    }
    auto f = std::make_unique<FastGzSource>();
    if (f->st.open(path)) return f;
  }
  gzFile gz = gzopen(path, "rb");
  if (!gz) return nullptr;
  gzbuffer(gz, 1u << 20);
  *gz_out = gz;
  return std::make_unique<GzSource>(gz);
}

// BGZF input: block-parallel inflate straight into the pinned chunk (same bytes as gzread would produce).
struct BgzfSource : Source {
  int fd;
  uint64_t pos = 0, fsize;
  std::vector<uint8_t> cbuf, carry;
  uint64_t carry_off = 0;
  gzFile fallback = nullptr;   // serial zlib from the first non-BGZF member on
  bool done = false;
  const uint8_t* map = nullptr;   // the compressed file, mapped: block headers are parsed and blocks inflated in place
  BgzfSource(int fd_, uint64_t size) : fd(fd_), fsize(size) {
    void* m = size ? mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd_, 0) : MAP_FAILED;
    if (m != MAP_FAILED) { map = static_cast<const uint8_t*>(m); (void)madvise(m, size, MADV_SEQUENTIAL); }
  }
  ~BgzfSource() override { if (fallback) gzclose(fallback); if (map) munmap(const_cast<uint8_t*>(map), fsize); }

  int64_t serial_fill(uint8_t* dst, uint64_t cap) {
    uint64_t got = 0;
    while (got < cap) {
      int r = gzread(fallback, dst + got, (unsigned)std::min<uint64_t>(cap - got, 1u << 30));
      if (r < 0) return SCFQ_EGZ;
      if (r == 0) break;
      got += (unsigned)r;
    }
    return (int64_t)got;
  }

  int64_t fill(uint8_t* dst, uint64_t cap) override {
    if (fallback) return serial_fill(dst, cap);
    if (carry_off < carry.size()) {   // leftover of a block larger than an earlier (tiny) chunk
      const uint64_t k = std::min<uint64_t>(cap, carry.size() - carry_off);
      std::memcpy(dst, carry.data() + carry_off, k);
      carry_off += k;
      return (int64_t)k;
    }
    if (done || pos >= fsize) return 0;
    // compressed window: BGZF blocks are <= 64 KiB in and out, so `cap` compressed bytes cover >= `cap` output in
    // all but pathological (stored) cases; a short window only means a shorter chunk
    const uint64_t want = std::min<uint64_t>(fsize - pos, std::max<uint64_t>(cap, 1u << 20));
    uint64_t avail = 0;
    const uint8_t* cb;
    if (map) {
      cb = map + pos;
      avail = want;
    } else {      // not mappable: copy the window out of the file
      cbuf.resize(want);
      while (avail < want) {
        ssize_t r = pread(fd, cbuf.data() + avail, want - avail, (off_t)(pos + avail));
        if (r < 0) return SCFQ_EIO;
        if (r == 0) break;
        avail += (uint64_t)r;
      }
      cb = cbuf.data();
    }
    std::vector<scfq_bgzf::Block> blocks;
    uint64_t p = 0, out = 0;
    bool to_serial = false;
    while (p < avail) {
      uint32_t hl = 0;
      const uint32_t bs = scfq_bgzf::block_size(cb + p, avail - p, &hl);
      if (!bs) {
        const bool tail_short = (avail - p < 18) && (pos + avail < fsize);
        if (tail_short) break;                                   // header split by the window: next fill
        if (avail - p >= 2 && cb[p] == 0x1f && cb[p + 1] == 0x8b) to_serial = true;   // ordinary gzip member
        else done = true;                                        // trailing garbage after a gzip stream: ignored, as zlib does
        break;
      }
      if (p + bs > avail) {
        if (pos + avail >= fsize) return SCFQ_EGZ;               // truncated final block
        break;                                                   // block split by the window: next fill
      }
      const uint32_t isize = scfq_bgzf::rd32(cb + p + bs - 4);
      if (out + isize > cap) {
        if (!blocks.empty()) break;
        // a single block larger than the chunk: inflate it aside and serve it in pieces
        std::vector<scfq_bgzf::Block> one{{p, bs, hl, isize, scfq_bgzf::rd32(cb + p + bs - 8), 0}};
        carry.assign(isize, 0);
        if (scfq_bgzf::inflate_blocks(cb, one, 0, 1, carry.data())) return SCFQ_EGZ;
        pos += p + bs;
        carry_off = std::min<uint64_t>(cap, carry.size());
        std::memcpy(dst, carry.data(), carry_off);
        return (int64_t)carry_off;
      }
      blocks.push_back({p, bs, hl, isize, scfq_bgzf::rd32(cb + p + bs - 8), out});
      out += isize;
      p += bs;
    }
    if (!blocks.empty()) {
      static const int nthr = std::max(1, std::min(64, env_int("SCFQ_INFLATE_THREADS",
                                                               (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency())))));
      const int nt = (int)std::min<size_t>((size_t)nthr, (blocks.size() + 7) / 8);
      std::vector<int> rcs(nt, 0);
      std::vector<std::thread> th;
      for (int t = 1; t < nt; ++t)
        th.emplace_back([&, t] { rcs[t] = scfq_bgzf::inflate_blocks(cb, blocks, blocks.size() * t / nt, blocks.size() * (t + 1) / nt, dst); });
      rcs[0] = scfq_bgzf::inflate_blocks(cb, blocks, 0, blocks.size() / nt, dst);
      for (auto& x : th) x.join();
      for (int r : rcs) if (r) return SCFQ_EGZ;
    }
    pos += p;
    if (to_serial) {
      const int fd2 = dup(fd);
      if (fd2 < 0 || lseek(fd2, (off_t)pos, SEEK_SET) < 0) return SCFQ_EIO;
      fallback = gzdopen(fd2, "rb");
      if (!fallback) { close(fd2); return SCFQ_EGZ; }
      gzbuffer(fallback, 1u << 20);
      if (out == 0) return serial_fill(dst, cap);
    }
    if (out == 0 && !done && pos < fsize && blocks.empty() && !to_serial) return SCFQ_EGZ;   // no progress possible
    return (int64_t)out;
  }
};

