// scfq_pgz.hpp — ONE gzip member inflated by many host threads (BASELINE config 4: "10 GB gzip-compressed FASTQ").
//
// A deflate stream is serial in two ways: the bit position of a block is only known once the previous block has been
// decoded, and a back-reference may reach into the 32 KiB before the block.  Both are worked around the way pugz /
// rapidgzip do it:
//   1. sync: a thread that starts at an arbitrary byte offset scans bit positions for something that parses as the
//      header of a dynamic-Huffman block (complete code-length code, complete literal/length and distance codes, an
//      end-of-block code) AND decodes cleanly for a few thousand symbols.  Random bits pass with negligible probability,
//      and a wrong sync is caught anyway: the previous segment must arrive at exactly that bit position as a block
//      boundary, and the member's CRC-32 has the last word.
//   2. unknown window: the segment is decoded into 16-bit symbols behind 32768 marker symbols (0x8000 | k = "byte k of
//      the window I do not have yet").  Back-references copy symbols, known or not.  When the segment before has been
//      resolved, its last 32 KiB are the window and every marker becomes a byte (window tails sequentially, 32 KiB per
//      segment; everything else in parallel, fused with the CRC-32 of the piece).
// A batch = T segments of the compressed file; batch k+1 is decoded while batch k is served.  Anything unexpected (no sync
// found, a segment that overshoots the next sync, a member that ends) ends the batch at the last good segment: the next
// batch starts there with an exact position and window, so the worst case is the serial decoder's speed, never a wrong byte.
// Semantics are gzread's (scfq_gzfast.hpp): concatenated members, trailing garbage ignored, CRC-32 / ISIZE checked.
#pragma once
#include "scfq_gzfast.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <new>
#include <string>

namespace scfq_pgz {

using scfq_gzfast::kWindow;
using scfq_gzfast::member_header;

constexpr uint64_t kTrialSymbols = 4096;

// does a dynamic-Huffman block start at bit `bit` of [base, end)?  (header parses, and the block decodes cleanly for a while)
inline bool plausible_block(scfq_inflate::Decoder& d, const uint8_t* base, const uint8_t* end, uint64_t bit, std::vector<uint16_t>& scratch) {
  d.begin_at_bit(base, end, bit);
  if (d.bitcnt < 3) return false;
  const uint32_t hdr = d.peek(3);
  if ((hdr >> 1) != 2) return false;                 // BTYPE 2 only (and either BFINAL)
  d.drop(3);
  d.last_block = hdr & 1;
  if (!d.read_dynamic_header()) return false;
  d.state = scfq_inflate::Decoder::kHuff;
  d.total_out = kWindow;
  uint16_t* out = scratch.data() + kWindow;
  uint16_t* const lim = scratch.data() + scratch.size();
  d.stop_bit = 0;                                    // stop at the first block boundary
  const int r = d.run16(out, lim);
  if (r < 0) return false;
  return r == scfq_inflate::kAtBoundary || r == scfq_inflate::kStreamEnd || (r == scfq_inflate::kNeedOutput && (uint64_t)(out - (scratch.data() + kWindow)) >= kTrialSymbols);
}

// Large scratch buffers: 2 MiB-aligned and advised as huge pages, never value-initialised.  A dozen threads first-touching
// 80 MB each through 4 KiB faults serialise on the address-space lock (the first batch took 3x as long as later ones).
struct BigFree { void operator()(void* p) const { std::free(p); } };
template <typename T> using BigBuf = std::unique_ptr<T[], BigFree>;
template <typename T> inline BigBuf<T> big_alloc(size_t n) {
  void* p = nullptr;
  const size_t bytes = ((n * sizeof(T) + (2u << 20) - 1) / (2u << 20)) * (2u << 20);
  if (posix_memalign(&p, 2u << 20, bytes) != 0) throw std::bad_alloc();
  (void)madvise(p, bytes, MADV_HUGEPAGE);
  return BigBuf<T>(static_cast<T*>(p));
}

struct Segment {
  uint64_t start_bit = 0, end_bit = 0;               // relative to the member's deflate data
  BigBuf<uint16_t> sym;                              // [kWindow markers or bytes | symbols], never value-initialised
  size_t cap = 0;
  uint64_t n = 0;                                    // symbols produced
  bool synced = false, ok = false, member_end = false;
};

// streams open at the same time in this process (sc fq-count --jobs=N): they share the CPU budget
inline std::atomic<int>& active_streams() { static std::atomic<int> n{0}; return n; }

class Stream {
 public:
  ~Stream() { close(); }

  bool open(const char* path) {
    // fewer than 4 threads per stream left: the serial reader (one decoder thread + the ingest thread) is the better use
    if ((active_streams().load() + 1) * 4 > n_threads()) return false;
    fd_ = ::open(path, O_RDONLY);
    if (fd_ < 0) return false;
    struct stat sb;
    if (fstat(fd_, &sb) != 0 || !S_ISREG(sb.st_mode) || sb.st_size < ((long)std::max(0, env_mb("SCFQ_PGZ_MIN_MB", 8)) << 20) || sb.st_size < 18) { close(); return false; }   // small files: serial reader
    n_ = (size_t)sb.st_size;
    void* m = mmap(nullptr, n_, PROT_READ, MAP_PRIVATE, fd_, 0);
    if (m == MAP_FAILED) { close(); return false; }
    map_ = static_cast<const uint8_t*>(m);
    if (member_header(map_, n_) <= 0) { close(); return false; }
    path_ = path;
    counted_ = true;
    active_streams()++;
    return true;
  }

  void close() {
    if (counted_) { active_streams()--; counted_ = false; }
    if (th_.joinable()) {
      { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
      cv_.notify_all();
      th_.join();
    }
    if (map_) { munmap(const_cast<uint8_t*>(map_), n_); map_ = nullptr; }
    if (fd_ >= 0) { ::close(fd_); fd_ = -1; }
  }

  // up to cap bytes of the inflated stream; 0 at the end, -1 on a corrupt stream
  int64_t next_chunk(uint8_t* dst, uint64_t cap) {
    if (serial_) return serial_->next_chunk(dst, cap);
    if (!started_) { started_ = true; th_ = std::thread([this] { produce(); }); }
    for (;;) {
      if (cur_ && cur_off_ < cur_->n) {
        const uint64_t k = std::min<uint64_t>(cap, cur_->n - cur_off_);
        copy_pieces(dst, cur_->buf.get() + cur_off_, k);
        cur_off_ += k;
        return (int64_t)k;
      }
      if (cur_) {                                      // batch consumed: hand the buffer back
        const int st = cur_->status;
        { std::lock_guard<std::mutex> lk(mu_); cur_->state = Batch::kFree; }
        cv_.notify_all();
        cur_ = nullptr;
        if (st == 2) {
          // a file of many small members (cat of small .gz files, BGZF read as plain gzip): one member is not enough work to
          // split, so the rest goes through the serial reader, which starts at the next member
          serial_.reset(new scfq_gzfast::Stream());
          if (!serial_->open(path_.c_str(), handoff_)) { serial_.reset(); finished_ = failed_ = true; return -1; }
          return serial_->next_chunk(dst, cap);
        }
        if (st != 0) { finished_ = true; failed_ = st < 0; }
      }
      if (finished_) return failed_ ? -1 : 0;
      std::unique_lock<std::mutex> lk(mu_);
      cv_.wait(lk, [&] { return batches_[take_].state == Batch::kReady; });
      cur_ = &batches_[take_];
      cur_off_ = 0;
      take_ ^= 1;
    }
  }

 private:
  struct Batch {
    enum State { kFree, kReady } state = kFree;
    BigBuf<uint8_t> buf;                               // never value-initialised (a vector resize zero-fills half a GB per batch)
    size_t cap = 0;
    size_t n = 0;                                      // valid bytes
    int status = 0;                                    // 0 more, 1 end of stream, -1 error
  };

  // CPUs this process may really use: the cgroup's CPU quota when there is one (a box shared between GPUs hands each
  // tenant a slice of its cores; running more threads than the quota gets the whole group throttled for the rest of the
  // scheduler period, measured here as 3x slower phases), else the hardware count
  static int usable_cpus() {
    int n = (int)std::max(1u, std::thread::hardware_concurrency());
    if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
      long long quota = 0, period = 0;
      char q[32] = {0};
      if (std::fscanf(f, "%31s %lld", q, &period) == 2 && q[0] != 'm' && period > 0) { quota = std::atoll(q); if (quota > 0) n = std::min<int>(n, (int)((quota + period - 1) / period)); }
      std::fclose(f);
    }
    return n;
  }
  static int n_threads() {
    static const int n = [] { const char* e = std::getenv("SCFQ_INFLATE_THREADS"); int v = e ? std::atoi(e) : 0;
                              if (v <= 0) v = std::max(1, std::min(16, usable_cpus()) - 4);     // room for the ingest thread and its copies
                              return std::min(v, 64); }();
    return n;
  }

  // runs fn(k) for k = 0..n-1 on n threads
  template <typename F> static void parallel(int n, F&& fn) {
    std::vector<std::thread> th;
    for (int k = 1; k < n; ++k) th.emplace_back([&, k] { fn(k); });
    fn(0);
    for (auto& t : th) t.join();
  }

  void produce() {
    const uint8_t* const file_end = map_ + n_;
    const uint8_t* member = map_;                      // current member header
    int put = 0;
    std::vector<uint8_t> window(kWindow, 0);           // last 32 KiB of the current member's output so far
    uint64_t window_valid = 0;                         // how many of them exist (member start: 0)
    uint32_t crc = 0, isize = 0;
    long h = member_header(member, (size_t)(file_end - member));
    const uint8_t* data = member + h;                  // deflate data of the current member
    uint64_t bit = 0;                                  // exact position of the next block header, relative to data
    const int T_max = n_threads();
    const uint64_t seg_default = (uint64_t)std::max(1, env_mb("SCFQ_PGZ_SEGMENT_MB", 4)) << 20;
    uint64_t seg_bytes = seg_default;       // shrinks when members turn out to be smaller than a batch (see the end of a member)
    std::vector<BigBuf<uint16_t>> symbuf((size_t)T_max);            // reused from batch to batch
    std::vector<size_t> symcap((size_t)T_max, 0);
    for (;;) {
      Batch* B = &batches_[put];
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return stop_ || B->state == Batch::kFree; });
        if (stop_) return;
      }
      B->n = 0;
      B->status = 0;
      const uint64_t search_bytes = std::min<uint64_t>(seg_bytes, 1u << 20);   // a dynamic block starts every few 10 KB in practice
      const int T = std::min(T_max, std::max(2, T_max / std::max(1, active_streams().load())));   // this batch's share of the CPU budget
      const bool verbose = std::getenv("SCFQ_VERBOSE") != nullptr;
      auto t_mark = std::chrono::steady_clock::now();
      auto lap = [&](const char* what) {
        if (!verbose) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "scfq pgz:   %-22s %7.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_mark).count());
        t_mark = now;
      };
      // ---- plan: segment k starts at the first plausible block at or after data + bit/8 + k * seg_bytes ----------------
      std::vector<Segment> seg((size_t)T);
      const uint64_t base_byte = bit >> 3;
      const uint64_t avail = (uint64_t)(file_end - data);
      seg[0].start_bit = bit;
      seg[0].synced = true;
      parallel(T, [&](int k) {
        if (k == 0) return;
        const uint64_t from = base_byte + (uint64_t)k * seg_bytes;
        const uint64_t to = std::min<uint64_t>(from + search_bytes, avail > 16 ? avail - 16 : 0);
        if (from >= to) return;
        auto d = std::unique_ptr<scfq_inflate::Decoder>(new scfq_inflate::Decoder());
        std::vector<uint16_t> scratch(kWindow + kTrialSymbols + 2 * scfq_inflate::kOutSlack);
        for (uint32_t i = 0; i < kWindow; ++i) scratch[i] = (uint16_t)(0x8000u | i);
        for (uint64_t b = from * 8; b < to * 8; ++b) {
          if (plausible_block(*d, data, file_end, b, scratch)) { seg[k].start_bit = b; seg[k].synced = true; return; }
        }
      });
      lap("sync search");
      // segments are only usable as an unbroken chain from segment 0
      int n_seg = 1;
      while (n_seg < T && seg[n_seg].synced) ++n_seg;
      // ---- decode: segment k until it stands on segment k+1's start (the last one: first boundary past its span) ------
      parallel(n_seg, [&](int k) {
        Segment& S = seg[k];
        auto d = std::unique_ptr<scfq_inflate::Decoder>(new scfq_inflate::Decoder());
        d->begin_at_bit(data, file_end, S.start_bit);
        d->total_out = (k == 0) ? window_valid : kWindow;       // segment 0 knows how much history really exists
        d->stop_bit = (k + 1 < n_seg) ? seg[k + 1].start_bit : (base_byte + (uint64_t)(k + 1) * seg_bytes) * 8;
        S.sym.swap(symbuf[(size_t)k]);
        S.cap = symcap[(size_t)k];
        if (S.cap < kWindow + (size_t)seg_bytes * 5 + scfq_inflate::kOutSlack) {
          S.cap = kWindow + (size_t)seg_bytes * 5 + scfq_inflate::kOutSlack;
          S.sym = big_alloc<uint16_t>(S.cap);
        }
        if (k == 0) {
          for (uint32_t i = 0; i < kWindow; ++i) S.sym[i] = window[i];                    // exact window: plain bytes
        } else {
          for (uint32_t i = 0; i < kWindow; ++i) S.sym[i] = (uint16_t)(0x8000u | i);      // markers
        }
        uint16_t* out = S.sym.get() + kWindow;
        for (;;) {
          const int r = d->run16(out, S.sym.get() + S.cap);
          if (r == scfq_inflate::kNeedOutput) {              // better compression than planned for: a bigger buffer
            // ... and past 64 M symbols the segment ends at its next block boundary instead of at the planned one
            // (highly compressible input, e.g. runs of one byte at 1000:1): the chain is cut there, memory stays bounded
            // by one block more, the next batch continues from that exact position
            if (S.cap >= (64u << 20)) d->stop_bit = 0;
            const size_t used = (size_t)(out - S.sym.get());
            BigBuf<uint16_t> bigger = big_alloc<uint16_t>(S.cap * 2);
            std::memcpy(bigger.get(), S.sym.get(), used * sizeof(uint16_t));
            S.sym.swap(bigger);
            S.cap *= 2;
            out = S.sym.get() + used;
            continue;
          }
          S.n = (uint64_t)(out - (S.sym.get() + kWindow));
          S.end_bit = d->bitpos();
          if (r == scfq_inflate::kAtBoundary) S.ok = true;
          else if (r == scfq_inflate::kStreamEnd) { S.ok = true; S.member_end = true; end_ptr_[k] = d->end_of_stream(); }
          else S.ok = false;
          return;
        }
      });
      lap("decode to symbols");
      // ---- keep the unbroken, consistent prefix of the chain ------------------------------------------------------------
      int good = 0;
      bool member_end = false;
      const uint8_t* end_ptr = nullptr;
      for (int k = 0; k < n_seg; ++k) {
        // segment k >= 1 counts only if the segment before it arrived exactly at its sync point (a wrong sync never does)
        if (!seg[k].ok || (k > 0 && seg[k - 1].end_bit != seg[k].start_bit)) break;
        ++good;
        if (seg[k].member_end) { member_end = true; end_ptr = end_ptr_[k]; break; }
      }
      if (std::getenv("SCFQ_VERBOSE")) std::fprintf(stderr, "scfq pgz: batch of %d segments, %d synced, %d chained\n", T, n_seg, good);
      if (good == 0) { B->status = -1; publish(B); return; }            // segment 0 starts exactly: its failure is a corrupt stream
      // ---- windows: the last 32 KiB after each good segment, sequentially -------------------------------------------------
      std::vector<std::vector<uint8_t>> win((size_t)good + 1, std::vector<uint8_t>(kWindow, 0));
      win[0] = window;
      std::vector<uint64_t> off((size_t)good + 1, 0);
      for (int k = 0; k < good; ++k) {
        const Segment& S = seg[k];
        off[k + 1] = off[k] + S.n;
        // tail of (window_k ++ symbols_k), resolved through window_k
        const uint16_t* all = S.sym.get();                 // [kWindow | n]: position p of the concatenation
        const uint64_t total = kWindow + S.n;
        for (uint32_t i = 0; i < kWindow; ++i) {
          const uint16_t v = all[total - kWindow + i];
          win[k + 1][i] = (v & 0x8000u) ? win[k][v & 0x7FFFu] : (uint8_t)v;
        }
        if (k == 0) continue;
      }
      lap("window tails");
      // a marker that points into a part of the window that does not exist (before the member's start) is corrupt data
      // (the serial decoder's "distance too far back"): checked per segment below with window_valid
      // ---- resolve + CRC, in parallel ---------------------------------------------------------------------------------------
      if (B->cap < off[good]) { B->cap = (size_t)off[good] + (size_t)(off[good] >> 3); B->buf = big_alloc<uint8_t>(B->cap); }
      B->n = (size_t)off[good];
      std::vector<uint32_t> part_crc((size_t)good, 0);
      std::atomic<int> bad{0};
      std::vector<uint64_t> valid((size_t)good + 1, 0);
      valid[0] = window_valid;
      for (int k = 0; k < good; ++k) valid[k + 1] = std::min<uint64_t>(kWindow, valid[k] + seg[k].n);
      parallel(good, [&](int k) {
        const Segment& S = seg[k];
        uint8_t* o = B->buf.get() + off[k];
        const uint16_t* s = S.sym.get() + kWindow;
        const uint8_t* w = win[k].data();
        const uint32_t missing = (uint32_t)(kWindow - valid[k]);       // window slots [0, missing) do not exist
        uint32_t or_bad = 0;
        uint64_t i = 0;
        // markers are rare once the first 32 KiB of a segment have been written: 16 symbols at a time, the scalar path only
        // for groups that contain one
        for (; i + 16 <= S.n; i += 16) {
          uint64_t w0, w1, w2, w3;
          std::memcpy(&w0, s + i, 8); std::memcpy(&w1, s + i + 4, 8); std::memcpy(&w2, s + i + 8, 8); std::memcpy(&w3, s + i + 12, 8);
          if (((w0 | w1 | w2 | w3) & 0x8000800080008000ull) == 0) {
            // pack the low bytes of 4 x u16 into 4 bytes, four times
            auto pack = [](uint64_t x) { x = (x | (x >> 8)) & 0x0000FFFF0000FFFFull; return (uint32_t)((x | (x >> 16)) & 0xFFFFFFFFull); };
            const uint32_t p0 = pack(w0), p1 = pack(w1), p2 = pack(w2), p3 = pack(w3);
            std::memcpy(o + i, &p0, 4); std::memcpy(o + i + 4, &p1, 4); std::memcpy(o + i + 8, &p2, 4); std::memcpy(o + i + 12, &p3, 4);
          } else {
            for (uint64_t j = i; j < i + 16; ++j) {
              const uint16_t v = s[j];
              if (v & 0x8000u) { const uint32_t idx = v & 0x7FFFu; or_bad |= (idx < missing); o[j] = w[idx]; }
              else o[j] = (uint8_t)v;
            }
          }
        }
        for (; i < S.n; ++i) {
          const uint16_t v = s[i];
          if (v & 0x8000u) { const uint32_t idx = v & 0x7FFFu; or_bad |= (idx < missing); o[i] = w[idx]; }
          else o[i] = (uint8_t)v;
        }
        if (or_bad) bad = 1;
        const auto tc0 = std::chrono::steady_clock::now();
        part_crc[k] = scfq_crc::crc32(0u, o, (size_t)S.n);
        if (verbose && k == 1) std::fprintf(stderr, "scfq pgz:     (segment 1: %llu bytes, crc32 alone %.1f ms)\n", (unsigned long long)S.n,
                                            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tc0).count());
      });
      lap("resolve + crc");
      if (bad) { B->status = -1; publish(B); return; }
      for (int k = 0; k < good; ++k) {
        crc = (uint32_t)crc32_combine(crc, part_crc[k], (z_off_t)seg[k].n);
        isize += (uint32_t)seg[k].n;
      }
      window = win[good];
      window_valid = valid[good];
      bit = seg[good - 1].end_bit;
      for (int k = 0; k < n_seg; ++k) { symbuf[(size_t)k].swap(seg[(size_t)k].sym); symcap[(size_t)k] = seg[(size_t)k].cap; }
      if (member_end) {
        // trailer, then another member or the end (gzread: trailing garbage is ignored)
        if (!end_ptr || end_ptr > file_end || file_end - end_ptr < 8) { B->status = -1; publish(B); return; }
        const uint32_t want_crc = (uint32_t)end_ptr[0] | ((uint32_t)end_ptr[1] << 8) | ((uint32_t)end_ptr[2] << 16) | ((uint32_t)end_ptr[3] << 24);
        const uint32_t want_len = (uint32_t)end_ptr[4] | ((uint32_t)end_ptr[5] << 8) | ((uint32_t)end_ptr[6] << 16) | ((uint32_t)end_ptr[7] << 24);
        if (want_crc != crc || want_len != isize) { B->status = -1; publish(B); return; }
        const uint8_t* const member_start = member;
        member = end_ptr + 8;
        h = member_header(member, (size_t)(file_end - member));
        if (h == 0) { B->status = 1; publish(B); return; }
        if (h < 0) { B->status = -1; publish(B); return; }
        const uint64_t member_size = (uint64_t)(member - member_start);
        if (member_size < (2u << 20)) {                                   // small members, and more follow: serial from here
          handoff_ = (size_t)(member - map_);
          B->status = 2;
          publish(B);
          return;
        }
        // members smaller than a batch (files concatenated from many mid-sized .gz): one member per batch, cut so that
        // every thread gets a segment of it
        seg_bytes = std::min<uint64_t>(seg_default, std::max<uint64_t>(512u << 10, member_size / (uint64_t)T_max + 1));
        data = member + h;
        bit = 0;
        crc = 0; isize = 0;
        std::fill(window.begin(), window.end(), 0);
        window_valid = 0;
      }
      publish(B);
      put ^= 1;
    }
  }

  void publish(Batch* B) {
    { std::lock_guard<std::mutex> lk(mu_); B->state = Batch::kReady; }
    cv_.notify_all();
  }

  // the consumer's copy into the staging buffer, split over a few threads (one core moves ~10 GB/s)
  static void copy_pieces(uint8_t* dst, const uint8_t* src, uint64_t n) {
    if (n < (8u << 20)) { std::memcpy(dst, src, n); return; }
    parallel(2, [&](int k) { const uint64_t lo = n * (uint64_t)k / 2, hi = n * (uint64_t)(k + 1) / 2; std::memcpy(dst + lo, src + lo, hi - lo); });
  }

  static int env_mb(const char* name, int dflt) { const char* e = std::getenv(name); return e ? std::atoi(e) : dflt; }

  int fd_ = -1;
  const uint8_t* map_ = nullptr;
  size_t n_ = 0, handoff_ = 0;
  std::string path_;
  std::unique_ptr<scfq_gzfast::Stream> serial_;
  Batch batches_[2];
  Batch* cur_ = nullptr;
  uint64_t cur_off_ = 0;
  const uint8_t* end_ptr_[64] = {nullptr};
  std::thread th_;
  std::mutex mu_;
  std::condition_variable cv_;
  bool stop_ = false, started_ = false, finished_ = false, failed_ = false, counted_ = false;
  int take_ = 0;
};

}  // namespace scfq_pgz
