// scfq_gzfast.hpp — gzip (RFC 1952) stream reader for regular files on top of scfq_inflate.hpp, with the semantics of
// zlib's gzread that the reference relies on (src/fq_count.nim:32, gzip_stream.nim:16-17): concatenated members are
// decoded one after the other, bytes after the last member that do not start another member are ignored, a wrong
// CRC-32 / ISIZE trailer or a corrupt deflate stream is an error.
//
// Shape: the compressed file is mmap'd; a decoder thread inflates into a small ring of chunk buffers (each preceded by
// the 32 KiB window of the previous chunk), the consumer (the ingest thread, otherwise idle while it waits for bytes)
// checks the CRC-32 of what it takes and copies it into the pinned staging buffer: inflate, CRC and the H2D copy of
// three consecutive chunks overlap.  Files that do not start with a gzip member (zlib would pass them through
// unchanged), FIFOs and anything that cannot be mmap'd are left to the zlib path by the caller (open() returns false).
#pragma once
#include "scfq_crc32.hpp"
#include "scfq_inflate.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace scfq_gzfast {

constexpr size_t kWindow = 32768;

// p[0..n): returns the offset of the deflate data of the member that starts at p, 0 when p does not start a gzip member,
// -1 when it does but the header is malformed / truncated (zlib: "unknown compression method", "unknown header flags set")
inline long member_header(const uint8_t* p, size_t n) {
  if (n < 2 || p[0] != 0x1f || p[1] != 0x8b) return 0;
  if (n < 10) return -1;
  if (p[2] != 8 || (p[3] & 0xE0)) return -1;
  const uint8_t flg = p[3];
  size_t q = 10;
  if (flg & 4) { if (q + 2 > n) return -1; q += 2 + ((size_t)p[q] | ((size_t)p[q + 1] << 8)); if (q > n) return -1; }
  if (flg & 8) { while (q < n && p[q]) ++q; if (q >= n) return -1; ++q; }
  if (flg & 16) { while (q < n && p[q]) ++q; if (q >= n) return -1; ++q; }
  if (flg & 2) { q += 2; if (q > n) return -1; }
  return (long)q;
}

class Stream {
 public:
  ~Stream() { close(); }

  // true: `path` is a regular file that has a gzip member at byte `start` (0: the usual case; scfq_pgz.hpp hands the rest of
  // a many-member file over at a member boundary) and is now mapped
  bool open(const char* path, size_t start = 0) {
    start_ = start;
    fd_ = ::open(path, O_RDONLY);
    if (fd_ < 0) return false;
    struct stat sb;
    if (fstat(fd_, &sb) != 0 || !S_ISREG(sb.st_mode) || sb.st_size < 18 || (size_t)sb.st_size <= start) { close(); return false; }
    n_ = (size_t)sb.st_size;
    void* m = mmap(nullptr, n_, PROT_READ, MAP_PRIVATE, fd_, 0);
    if (m == MAP_FAILED) { close(); return false; }
    map_ = static_cast<const uint8_t*>(m);
    (void)madvise(m, n_, MADV_SEQUENTIAL);
    if (member_header(map_ + start_, n_ - start_) <= 0) { close(); return false; }
    return true;
  }

  void close() {
    if (th_.joinable()) {
      { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
      cv_.notify_all();
      th_.join();
    }
    if (map_) { munmap(const_cast<uint8_t*>(map_), n_); map_ = nullptr; }
    if (fd_ >= 0) { ::close(fd_); fd_ = -1; }
  }

  // up to cap bytes of the inflated stream into dst; 0 at the end, -1 on a corrupt stream
  int64_t next_chunk(uint8_t* dst, uint64_t cap) {
    if (!started_) {
      cap_ = (size_t)cap;
      for (auto& s : slots_) s.buf.resize(kWindow + cap_);
      started_ = true;
      th_ = std::thread([this] { decode_loop(); });
    }
    if (finished_) return failed_ ? -1 : 0;
    Slot* s;
    {
      std::unique_lock<std::mutex> lk(mu_);
      cv_.wait(lk, [&] { return slots_[take_].state == Slot::kReady; });
      s = &slots_[take_];
    }
    int64_t got = (int64_t)s->n;
    const uint8_t* data = s->buf.data() + kWindow;
    if (s->n > cap) { failed_ = finished_ = true; return -1; }
    // CRC-32 / ISIZE of every member that ends inside this chunk (zlib: "incorrect data check" / "incorrect length check").
    // zlib's crc32 runs at about the speed of the decoder, so a chunk is checked (and copied out) in kPieces parallel pieces
    // whose CRCs are stitched with crc32_combine.
    size_t from = 0;
    bool bad = s->status < 0;
    for (const MemberEnd& e : s->ends) {
      crc_ = crc_and_copy(crc_, data + from, dst + from, e.off - from);
      isize_ += (uint32_t)(e.off - from);
      if (crc_ != e.crc || isize_ != e.isize) bad = true;
      crc_ = (uint32_t)crc32_z(0L, Z_NULL, 0);
      isize_ = 0;
      from = e.off;
    }
    crc_ = crc_and_copy(crc_, data + from, dst + from, s->n - from);
    isize_ += (uint32_t)(s->n - from);
    const bool last = s->status != 0;
    {
      std::lock_guard<std::mutex> lk(mu_);
      s->state = Slot::kFree;
    }
    cv_.notify_all();
    take_ = (take_ + 1) % kSlots;
    if (bad) { failed_ = finished_ = true; return -1; }
    if (last) finished_ = true;
    return got;
  }

 private:
  static constexpr int kPieces = 4;
  static uint32_t crc_and_copy(uint32_t crc, const uint8_t* src, uint8_t* dst, size_t n) {
    if (n < (4u << 20)) {
      std::memcpy(dst, src, n);
      return scfq_crc::crc32(crc, src, n);
    }
    uint32_t part[kPieces];
    size_t lo[kPieces + 1];
    for (int k = 0; k <= kPieces; ++k) lo[k] = n * (size_t)k / kPieces;
    std::thread th[kPieces - 1];
    auto work = [&](int k) {
      std::memcpy(dst + lo[k], src + lo[k], lo[k + 1] - lo[k]);
      part[k] = scfq_crc::crc32(k == 0 ? crc : 0u, src + lo[k], lo[k + 1] - lo[k]);
    };
    for (int k = 1; k < kPieces; ++k) th[k - 1] = std::thread(work, k);
    work(0);
    for (auto& t : th) t.join();
    uint32_t c = part[0];
    for (int k = 1; k < kPieces; ++k) c = (uint32_t)crc32_combine(c, part[k], (z_off_t)(lo[k + 1] - lo[k]));
    return c;
  }

  struct MemberEnd { size_t off; uint32_t crc, isize; };
  struct Slot {
    enum State { kFree, kReady } state = kFree;
    std::vector<uint8_t> buf;       // [window of the previous chunk | chunk]
    size_t n = 0;
    int status = 0;                 // 0 more to come, 1 end of stream, -1 error
    std::vector<MemberEnd> ends;
  };
  static constexpr int kSlots = 3;

  void decode_loop() {
    auto dec = std::unique_ptr<scfq_inflate::Decoder>(new scfq_inflate::Decoder());
    const uint8_t* end = map_ + n_;
    const long h = member_header(map_ + start_, n_ - start_);
    dec->begin(map_ + start_ + h, end);
    std::vector<uint8_t> window(kWindow, 0);
    int put = 0;
    for (;;) {
      Slot* s = &slots_[put];
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return stop_ || s->state == Slot::kFree; });
        if (stop_) return;
      }
      std::memcpy(s->buf.data(), window.data(), kWindow);
      uint8_t* const base = s->buf.data() + kWindow;
      uint8_t* out = base;
      uint8_t* const out_end = base + cap_;
      s->ends.clear();
      s->status = 0;
      for (;;) {
        const int r = dec->run(out, out_end);
        if (r == scfq_inflate::kNeedOutput) break;
        if (r < 0) { s->status = -1; break; }
        // end of a member: trailer, then another member, or the end (trailing garbage is ignored, as zlib does)
        const uint8_t* t = dec->end_of_stream();
        if (t > end || end - t < 8) { s->status = -1; break; }
        MemberEnd me;
        me.off = (size_t)(out - base);
        me.crc = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
        me.isize = (uint32_t)t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
        s->ends.push_back(me);
        t += 8;
        const long nh = member_header(t, (size_t)(end - t));
        if (nh == 0) { s->status = 1; break; }
        if (nh < 0) { s->status = -1; break; }
        dec->begin(t + nh, end);
      }
      s->n = (size_t)(out - base);
      // window for the next chunk: the last 32 KiB produced so far
      if (s->n >= kWindow) std::memcpy(window.data(), out - kWindow, kWindow);
      else {
        std::memmove(window.data(), window.data() + s->n, kWindow - s->n);
        std::memcpy(window.data() + kWindow - s->n, base, s->n);
      }
      const bool done = s->status != 0;
      {
        std::lock_guard<std::mutex> lk(mu_);
        s->state = Slot::kReady;
      }
      cv_.notify_all();
      if (done) return;
      put = (put + 1) % kSlots;
    }
  }

  int fd_ = -1;
  const uint8_t* map_ = nullptr;
  size_t n_ = 0, cap_ = 0, start_ = 0;
  Slot slots_[kSlots];
  std::thread th_;
  std::mutex mu_;
  std::condition_variable cv_;
  bool stop_ = false, started_ = false, finished_ = false, failed_ = false;
  int take_ = 0;
  uint32_t crc_ = 0, isize_ = 0;
};

// The REST of a gzip stream from a position in the middle of a member: the device inflate path (scfq_gzdev.hpp) hands over
// here when a batch is more than it can take — what its earlier batches folded stays, the host carries on from the exact bit
// the device's chain had reached, with the 32 KiB of history in front of it and the CRC-32 / length of the member so far.
// Same bytes and the same accept / reject decisions as Stream (and zlib's gzread): the current member's trailer is checked
// against prefix + rest, further members follow, trailing garbage is ignored.  One thread, the serial decoder: this is the
// rare path.
class Resume {
 public:
  // img[0 .. n): the whole file.  bit: a block header inside a member, or the first block of a member (then valid = 0, the
  // prefix is empty).  window: 32 KiB whose last `valid` bytes are the member's output in front of `bit`.
  // crc_prefix / len_prefix: zlib crc32 and length of the member's output in front of `bit`.
  void open(const uint8_t* img, size_t n, uint64_t bit, const uint8_t* window, uint32_t valid, uint32_t crc_prefix, uint64_t len_prefix) {
    img_ = img; n_ = n;
    window_.assign(window, window + kWindow);
    dec_.reset(new scfq_inflate::Decoder());
    dec_->begin_at_bit(img, img + n, bit);
    dec_->total_out = valid;
    crc_ = crc_prefix;
    len_ = len_prefix;
  }
  // up to cap bytes of the inflated stream into dst; 0 at the end, -1 on a corrupt stream
  int64_t next_chunk(uint8_t* dst, uint64_t cap) {
    if (finished_) return failed_ ? -1 : 0;
    if (buf_.size() < kWindow + cap + 512) buf_.resize(kWindow + (size_t)cap + 512);
    std::memcpy(buf_.data(), window_.data(), kWindow);
    uint8_t* const base = buf_.data() + kWindow;
    uint8_t* out = base;
    uint8_t* const out_end = base + cap;
    size_t from = 0;
    bool bad = false, last = false;
    const uint8_t* const end = img_ + n_;
    for (;;) {
      const int r = dec_->run(out, out_end);
      if (r == scfq_inflate::kNeedOutput) break;
      if (r < 0) { bad = true; break; }
      const uint8_t* t = dec_->end_of_stream();
      if (t > end || end - t < 8) { bad = true; break; }
      const size_t off = (size_t)(out - base);
      crc_ = scfq_crc::crc32(crc_, base + from, off - from);
      len_ += off - from;
      from = off;
      const uint32_t crc = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
      const uint32_t isize = (uint32_t)t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
      if (crc != crc_ || isize != (uint32_t)len_) { bad = true; break; }
      crc_ = 0; len_ = 0;
      t += 8;
      const long nh = member_header(t, (size_t)(end - t));
      if (nh == 0) { last = true; end_off_ = (size_t)(t - img_); break; }
      if (nh < 0) { bad = true; break; }
      dec_->begin(t + nh, end);
    }
    const size_t n_out = (size_t)(out - base);
    if (n_out > cap) { failed_ = finished_ = true; return -1; }
    crc_ = scfq_crc::crc32(crc_, base + from, n_out - from);
    len_ += n_out - from;
    std::memcpy(dst, base, n_out);
    if (n_out >= kWindow) std::memcpy(window_.data(), out - kWindow, kWindow);
    else {
      std::memmove(window_.data(), window_.data() + n_out, kWindow - n_out);
      std::memcpy(window_.data() + kWindow - n_out, base, n_out);
    }
    if (bad) { failed_ = finished_ = true; return -1; }
    if (last) finished_ = true;
    return (int64_t)n_out;
  }

 private:
  const uint8_t* img_ = nullptr;
  size_t n_ = 0;
  std::vector<uint8_t> window_, buf_;
  std::unique_ptr<scfq_inflate::Decoder> dec_;
  uint32_t crc_ = 0;
  uint64_t len_ = 0;
  bool finished_ = false, failed_ = false;
  size_t end_off_ = 0;

 public:
  // once next_chunk() has returned 0: the offset just behind the last member's trailer (the end of the image, or where trailing
  // bytes that are not a gzip member begin)
  size_t end_offset() const { return end_off_; }
};

}  // namespace scfq_gzfast
