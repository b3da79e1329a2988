// scfq_inflate.hpp — raw DEFLATE (RFC 1951) decoder for the host side of the gzip ingest path.
//
// Why: BASELINE config 4 (gzip-compressed FASTQ) is bound by the host inflate, which the reference does with zlib
// gzread (src/fq_count.nim:32 -> zip/gzipfiles; the same calls as gzip_stream.nim:16-17).  zlib 1.2.11's inflate yields
// ≈0.5 GB/s on FASTQ; this decoder keeps zlib's semantics (same bytes out, corrupt streams rejected) at roughly twice the
// rate: 64-bit bit buffer refilled with one unaligned load, an 11-bit first-level literal/length table whose entries carry
// base value, extra-bit count and code length in one word (one lookup per symbol, second level only for codes longer than
// 11 bits), word-wise match copies.  Resumable between symbols, so a stream can be produced in chunks (the pinned staging
// buffers of the ingest) with the 32 KiB window simply being the bytes before the write position.
// Byte-for-byte equality with zlib is tested on the CPU (tests/test_inflate_host.py), including corrupt input.
#pragma once
#include <cstdint>
#include <cstring>

namespace scfq_inflate {

constexpr int kLitRoot = 11, kDistRoot = 8;
constexpr int kLitSize = (1 << kLitRoot) + 2048, kDistSize = (1 << kDistRoot) + 512;   // first level + second-level space
constexpr uint32_t kOutSlack = 258 + 16;     // the fast loop may write this far past the position it was entered with

// table entry (u32):  bits 0..7 code length (bits to drop), 8..12 extra-bit count or second-level index bits, 13..15 flags,
//                     16..31 literal byte / length base / distance base / second-level start
constexpr uint32_t F_LITERAL = 1u << 13, F_EOB = 1u << 14, F_SUB = 1u << 15;
constexpr uint32_t kValShift = 16;
inline uint32_t e_len(uint32_t e) { return e & 0xFFu; }
inline uint32_t e_extra(uint32_t e) { return (e >> 8) & 0x1Fu; }
inline uint32_t e_val(uint32_t e) { return e >> kValShift; }

static const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
static const uint8_t kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

enum Status { kNeedOutput = 0, kStreamEnd = 1, kAtBoundary = 3, kErrData = -1, kErrTruncated = -2 };

struct Decoder {
  // ---- input: the whole compressed stream is addressable (mmap / buffer); `in_safe` = last address from which an 8-byte
  //      load stays inside it
  const uint8_t* in_next = nullptr;
  const uint8_t* in_end = nullptr;
  uint64_t bitbuf = 0;
  uint32_t bitcnt = 0;
  // ---- block state
  enum { kHeader, kStored, kHuff, kDone } state = kHeader;
  bool last_block = false;
  uint32_t stored_left = 0;
  uint64_t total_out = 0;            // bytes produced since the stream began (bounds match distances)
  uint32_t lit[kLitSize];
  uint32_t dist[kDistSize];

  // (tests: every block header's bit position and the bytes produced before it, as pairs — scfq_debug_gz_resume cuts a stream there)
  uint64_t* boundary_log = nullptr;
  size_t boundary_cap = 0, boundary_n = 0;
  const uint8_t* in_base = nullptr;   // bit positions are counted from here
  uint64_t stop_bit = ~0ull;          // run16: return kAtBoundary at the first block boundary at or after this bit

  void begin(const uint8_t* p, const uint8_t* end) {
    in_base = p; in_next = p; in_end = end; bitbuf = 0; bitcnt = 0; state = kHeader; last_block = false; stored_left = 0; total_out = 0;
    stop_bit = ~0ull;
  }
  // start in the middle of a stream: `bit` is the position, relative to base, of a block header
  void begin_at_bit(const uint8_t* base, const uint8_t* end, uint64_t bit) {
    begin(base, end);
    in_next = base + (bit >> 3);
    refill();
    drop((int)(bit & 7));
  }
  uint64_t bitpos() const { return (uint64_t)(in_next - in_base) * 8 - bitcnt; }
  // first byte after the deflate stream (valid after kStreamEnd): whole unread bytes of the bit buffer are given back
  const uint8_t* end_of_stream() const { return in_next - (bitcnt >> 3); }

  inline void refill() {
    if (in_end - in_next >= 8) {
      uint64_t w;
      std::memcpy(&w, in_next, 8);
      bitbuf |= w << bitcnt;
      in_next += (63 - bitcnt) >> 3;
      bitcnt |= 56;
    } else {
      while (bitcnt <= 56 && in_next < in_end) { bitbuf |= (uint64_t)*in_next++ << bitcnt; bitcnt += 8; }
    }
  }
  inline uint32_t peek(int n) const { return (uint32_t)(bitbuf & ((1ull << n) - 1)); }
  inline void drop(int n) { bitbuf >>= n; bitcnt -= (uint32_t)n; }

  // Canonical Huffman code -> two-level decode table (the construction of zlib's inftrees, restated).  kind: 0 = code
  // lengths alphabet (values are the symbols), 1 = literal/length, 2 = distance.  Returns false for an over-subscribed
  // code or an incomplete one that zlib rejects.
  static bool build(const uint8_t* lens, int n, int kind, uint32_t* tab, int root, int tab_cap) {
    uint16_t count[16] = {0}, offs[16], sorted[320];
    for (int s = 0; s < n; ++s) count[lens[s]]++;
    int max = 15;
    while (max >= 1 && !count[max]) --max;
    const int first = 1 << root;
    if (max == 0) {                       // no codes at all: zlib builds a table that fails on use
      for (int k = 0; k < first; ++k) tab[k] = 0;
      return kind == 2 || kind == 0 ? true : false;
    }
    int left = 1;
    for (int len = 1; len <= 15; ++len) { left <<= 1; left -= count[len]; if (left < 0) return false; }
    if (left > 0 && (kind == 0 || max != 1)) return false;       // incomplete set (zlib: only a single 1-bit code may be)
    offs[1] = 0;
    for (int len = 1; len < 15; ++len) offs[len + 1] = (uint16_t)(offs[len] + count[len]);
    for (int s = 0; s < n; ++s) if (lens[s]) sorted[offs[lens[s]]++] = (uint16_t)s;
    for (int k = 0; k < first; ++k) tab[k] = 0;                  // unassigned patterns decode as invalid
    int next_free = first;
    uint32_t code = 0;                     // MSB-first canonical code of the current symbol
    int sym_i = 0;
    uint32_t cur_prefix = ~0u;
    int sub_start = 0, sub_bits = 0;
    for (int len = 1; len <= max; ++len) {
      for (int c = 0; c < count[len]; ++c, ++sym_i) {
        const int sym = sorted[sym_i];
        uint32_t e;
        if (kind == 0) e = ((uint32_t)sym << kValShift) | (uint32_t)len;
        else if (kind == 1) {
          if (sym < 256) e = F_LITERAL | ((uint32_t)sym << kValShift) | (uint32_t)len;
          else if (sym == 256) e = F_EOB | (uint32_t)len;
          else if (sym <= 285) e = ((uint32_t)kLenBase[sym - 257] << kValShift) | ((uint32_t)kLenExtra[sym - 257] << 8) | (uint32_t)len;
          else e = 0;                      // 286, 287: never valid in a stream
        } else {
          if (sym < 30) e = ((uint32_t)kDistBase[sym] << kValShift) | ((uint32_t)kDistExtra[sym] << 8) | (uint32_t)len;
          else e = 0;
        }
        // bit-reverse the code: the stream is read least-significant bit first
        uint32_t rev = 0;
        for (int b = 0; b < len; ++b) rev |= ((code >> b) & 1u) << (len - 1 - b);
        if (len <= root) {
          for (uint32_t k = rev; k < (uint32_t)first; k += 1u << len) tab[k] = e;
        } else {
          const uint32_t prefix = rev & (uint32_t)(first - 1);
          if (prefix != cur_prefix) {
            // new second-level table: wide enough for the longest code that shares these first `root` bits
            int curr = len - root, l2 = 1 << curr;
            int cnt_left = l2 - (count[len] - c);
            int ll = len;
            while (cnt_left > 0 && ll < max) { ++ll; ++curr; cnt_left = (cnt_left << 1) - count[ll]; }
            sub_bits = curr;
            sub_start = next_free;
            next_free += 1 << sub_bits;
            if (next_free > tab_cap) return false;
            for (int k = sub_start; k < next_free; ++k) tab[k] = 0;
            tab[prefix] = F_SUB | ((uint32_t)sub_start << kValShift) | ((uint32_t)sub_bits << 8) | (uint32_t)root;
            cur_prefix = prefix;
          }
          const uint32_t hi = rev >> root;
          const int hl = len - root;
          e = (e & ~0xFFu) | (uint32_t)hl;       // second-level entries drop only the bits beyond the first level
          for (uint32_t k = hi; k < (1u << sub_bits); k += 1u << hl) tab[sub_start + k] = e;
        }
        ++code;
      }
      code <<= 1;
    }
    return true;
  }

  bool read_dynamic_header() {
    refill();
    if (bitcnt < 14) return false;
    const int hlit = (int)peek(5) + 257; drop(5);
    const int hdist = (int)peek(5) + 1; drop(5);
    const int hclen = (int)peek(4) + 4; drop(4);
    if (hlit > 286 || hdist > 30) return false;
    uint8_t cl[19] = {0};
    for (int k = 0; k < hclen; ++k) {
      refill();
      if (bitcnt < 3) return false;
      cl[kClOrder[k]] = (uint8_t)peek(3); drop(3);
    }
    uint32_t cltab[1 << 7];
    if (!build(cl, 19, 0, cltab, 7, 1 << 7)) return false;
    uint8_t lens[320];
    int k = 0;
    while (k < hlit + hdist) {
      refill();
      const uint32_t e = cltab[peek(7)];
      const int len = (int)e_len(e);
      if (!len || (uint32_t)len > bitcnt) return false;
      drop(len);
      const int sym = (int)e_val(e);
      if (sym < 16) { lens[k++] = (uint8_t)sym; continue; }
      int rep, val = 0;
      if (sym == 16) { if (k == 0 || bitcnt < 2) return false; val = lens[k - 1]; rep = 3 + (int)peek(2); drop(2); }
      else if (sym == 17) { if (bitcnt < 3) return false; rep = 3 + (int)peek(3); drop(3); }
      else { if (bitcnt < 7) return false; rep = 11 + (int)peek(7); drop(7); }
      if (k + rep > hlit + hdist) return false;
      while (rep--) lens[k++] = (uint8_t)val;
    }
    if (lens[256] == 0) return false;                              // no end-of-block code
    return build(lens, hlit, 1, lit, kLitRoot, kLitSize) && build(lens + hlit, hdist, 2, dist, kDistRoot, kDistSize);
  }

  void load_fixed() {
    uint8_t lens[320];
    int k = 0;
    for (; k < 144; ++k) lens[k] = 8;
    for (; k < 256; ++k) lens[k] = 9;
    for (; k < 280; ++k) lens[k] = 7;
    for (; k < 288; ++k) lens[k] = 8;
    build(lens, 288, 1, lit, kLitRoot, kLitSize);
    for (k = 0; k < 32; ++k) lens[k] = 5;
    build(lens, 32, 2, dist, kDistRoot, kDistSize);               // 30, 31 map to invalid entries
  }

  // Produce output at [out_next, out_end); the bytes before out_next (total_out of them, at most 32 KiB are needed) are
  // the window.  Returns kNeedOutput when fewer than kOutSlack bytes remain (call again with a new buffer whose preceding
  // 32 KiB hold the previous output), kStreamEnd after the final block, or an error.
  int run(uint8_t*& out_next, uint8_t* out_end) {
    for (;;) {
      if (state == kDone) return kStreamEnd;
      if (state == kHeader) {
        if (boundary_log && boundary_n < boundary_cap) { boundary_log[2 * boundary_n] = bitpos(); boundary_log[2 * boundary_n + 1] = total_out; ++boundary_n; }
        refill();
        if (bitcnt < 3) return kErrTruncated;
        last_block = peek(1); drop(1);
        const uint32_t type = peek(2); drop(2);
        if (type == 0) {
          drop((int)(bitcnt & 7));                                  // to the byte boundary
          refill();
          if (bitcnt < 32) return kErrTruncated;
          const uint32_t len = peek(16), nlen = (uint32_t)((bitbuf >> 16) & 0xFFFF);
          if ((len ^ nlen) != 0xFFFF) return kErrData;
          drop(32);
          in_next -= bitcnt >> 3;                                   // whole bytes back to the byte stream
          bitbuf = 0; bitcnt = 0;
          stored_left = len;
          state = kStored;
        } else if (type == 1) { load_fixed(); state = kHuff; }
        else if (type == 2) { if (!read_dynamic_header()) return kErrData; state = kHuff; }
        else return kErrData;
      }
      if (state == kStored) {
        const uint64_t room = (uint64_t)(out_end - out_next), avail = (uint64_t)(in_end - in_next);
        uint64_t k = stored_left;
        if (k > room) k = room;
        if (k > avail) k = avail;
        std::memcpy(out_next, in_next, k);
        out_next += k; in_next += k; stored_left -= (uint32_t)k; total_out += k;
        if (stored_left) {
          if (in_next == in_end) return kErrTruncated;
          return kNeedOutput;
        }
        state = last_block ? kDone : kHeader;
        continue;
      }
      // ---- Huffman block: one table lookup per literal / length, one per distance --------------------------------
      // The bit reader and the write position live in locals for the duration of the loop: byte stores may alias any
      // member (char aliasing), which would otherwise force a reload of every field after every literal.
      {
        const int r = huffman_block(out_next, out_end);
        if (r != kBlockEnd) return r;
      }
    }
  }

  static constexpr int kBlockEnd = 2;

  // T = uint8_t (bytes) or uint16_t (symbols for the parallel reader: literals and window markers)
  template <typename T>
  int huffman_block_t(T*& out_ref, T* const out_end) {
    constexpr int kPer = 8 / (int)sizeof(T);          // symbols per 64-bit word
    T* out = out_ref;
    T* const run_start = out;
    const uint64_t total_at_entry = total_out;
    const uint8_t* in = in_next;
    const uint8_t* const iend = in_end;
    uint64_t bb = bitbuf;
    uint32_t bc = bitcnt;
    const uint32_t* const lt = lit;
    const uint32_t* const dt = dist;
    int result;
#define SCFQ_REFILL()                                                                                    \
    do {                                                                                                 \
      if (iend - in >= 8) {                                                                              \
        uint64_t w_;                                                                                     \
        std::memcpy(&w_, in, 8);                                                                         \
        bb |= w_ << bc;                                                                                  \
        in += (63 - bc) >> 3;                                                                            \
        bc |= 56;                                                                                        \
      } else {                                                                                           \
        while (bc <= 56 && in < iend) { bb |= (uint64_t)*in++ << bc; bc += 8; }                          \
      }                                                                                                  \
    } while (0)
#define SCFQ_DROP(n) do { bb >>= (n); bc -= (uint32_t)(n); } while (0)
#define SCFQ_FAIL() do { result = (in >= iend) ? (int)kErrTruncated : (int)kErrData; goto done; } while (0)
    // ---- fast loop: while at least 8 input bytes and kOutSlack output bytes remain, a refill is one unconditional load
    //      and leaves >= 56 bits: enough for a whole literal/length + distance pair (15 + 5 + 15 + 13 = 48), so none of
    //      the "do I have the bits" checks of the careful loop below are needed
    if (iend - in >= 16) {
      const uint8_t* const in_fast = iend - 8;
      T* const out_fast = out_end - kOutSlack;
      while (in <= in_fast && out <= out_fast) {
        {
          uint64_t w_;
          std::memcpy(&w_, in, 8);
          bb |= w_ << bc;
          in += (63 - bc) >> 3;
          bc |= 56;
        }
        uint32_t e = lt[bb & ((1u << kLitRoot) - 1)];
        if (e & F_LITERAL) {
          SCFQ_DROP(e_len(e));
          *out++ = (T)e_val(e);
          e = lt[bb & ((1u << kLitRoot) - 1)];
          if (e & F_LITERAL) {
            SCFQ_DROP(e_len(e));
            *out++ = (T)e_val(e);
            e = lt[bb & ((1u << kLitRoot) - 1)];
            if (e & F_LITERAL) {
              SCFQ_DROP(e_len(e));
              *out++ = (T)e_val(e);
              e = lt[bb & ((1u << kLitRoot) - 1)];
              if (e & F_LITERAL) {                 // a fourth: 4 x 11 = 44 <= 56 bits, first-level codes only
                SCFQ_DROP(e_len(e));
                *out++ = (T)e_val(e);
              }
            }
          }
          continue;
        }
        if (e & F_SUB) {
          e = lt[e_val(e) + ((bb >> kLitRoot) & ((1u << e_extra(e)) - 1))];
          SCFQ_DROP(kLitRoot);
          if (e & F_LITERAL) { SCFQ_DROP(e_len(e)); *out++ = (T)e_val(e); continue; }
        }
        if (e_len(e) == 0) { result = kErrData; goto done; }
        SCFQ_DROP(e_len(e));
        if (e & F_EOB) { state = last_block ? kDone : kHeader; result = kBlockEnd; goto done; }
        const uint32_t lx = e_extra(e);
        const uint32_t mlen = e_val(e) + (uint32_t)(bb & ((1ull << lx) - 1));
        SCFQ_DROP(lx);
        uint32_t d = dt[bb & ((1u << kDistRoot) - 1)];
        if (d & F_SUB) {
          d = dt[e_val(d) + ((bb >> kDistRoot) & ((1u << e_extra(d)) - 1))];
          SCFQ_DROP(kDistRoot);
        }
        if (e_len(d) == 0) { result = kErrData; goto done; }
        SCFQ_DROP(e_len(d));
        const uint32_t dx = e_extra(d);
        const uint32_t off = e_val(d) + (uint32_t)(bb & ((1ull << dx) - 1));
        SCFQ_DROP(dx);
        if (off > total_at_entry + (uint64_t)(out - run_start)) { result = kErrData; goto done; }
        const T* src = out - off;
        T* dst = out;
        out += mlen;
        if (off >= (uint32_t)kPer) {
          // most matches are short: two unconditional words, then the rest
          uint64_t w0, w1;
          std::memcpy(&w0, src, 8); std::memcpy(dst, &w0, 8);
          std::memcpy(&w1, src + kPer, 8); std::memcpy(dst + kPer, &w1, 8);
          if (mlen > 2u * kPer) {
            T* const stop = dst + mlen;
            src += 2 * kPer; dst += 2 * kPer;
            do { uint64_t w; std::memcpy(&w, src, 8); std::memcpy(dst, &w, 8); src += kPer; dst += kPer; } while (dst < stop);
          }
        } else if (off == 1) {
          const T v = *src;
          for (uint32_t k = 0; k < mlen; ++k) dst[k] = v;
        } else {
          for (uint32_t k = 0; k < mlen; ++k) dst[k] = src[k];
        }
      }
    }
    // ---- careful loop: the last bytes of the input / of the output buffer -----------------------------------------
    for (;;) {
      if ((uint64_t)(out_end - out) < kOutSlack) { result = kNeedOutput; goto done; }
      SCFQ_REFILL();
      uint32_t e = lt[bb & ((1u << kLitRoot) - 1)];
      if (e & F_SUB) {
        e = lt[e_val(e) + ((bb >> kLitRoot) & ((1u << e_extra(e)) - 1))];
        SCFQ_DROP(kLitRoot);
      }
      uint32_t len = e_len(e);
      if (len > bc) SCFQ_FAIL();
      if (e & F_LITERAL) {
        SCFQ_DROP(len);
        *out++ = (T)e_val(e);
        // a second and third literal usually fit in the bits already loaded
        e = lt[bb & ((1u << kLitRoot) - 1)];
        if ((e & F_LITERAL) && e_len(e) <= bc) {
          SCFQ_DROP(e_len(e));
          *out++ = (T)e_val(e);
          e = lt[bb & ((1u << kLitRoot) - 1)];
          if ((e & F_LITERAL) && e_len(e) <= bc) {
            SCFQ_DROP(e_len(e));
            *out++ = (T)e_val(e);
          }
        }
        continue;
      }
      if (len == 0) { result = kErrData; goto done; }           // unassigned code
      SCFQ_DROP(len);
      if (e & F_EOB) { state = last_block ? kDone : kHeader; result = kBlockEnd; goto done; }
      {
        const uint32_t lx = e_extra(e);
        uint32_t mlen = e_val(e);
        if (lx > bc) SCFQ_FAIL();
        mlen += (uint32_t)(bb & ((1ull << lx) - 1));
        SCFQ_DROP(lx);
        if (bc < 15 + 13) SCFQ_REFILL();
        uint32_t d = dt[bb & ((1u << kDistRoot) - 1)];
        if (d & F_SUB) {
          d = dt[e_val(d) + ((bb >> kDistRoot) & ((1u << e_extra(d)) - 1))];
          SCFQ_DROP(kDistRoot);
        }
        const uint32_t dl = e_len(d);
        if (dl == 0) { result = kErrData; goto done; }
        if (dl > bc) SCFQ_FAIL();
        SCFQ_DROP(dl);
        const uint32_t dx = e_extra(d);
        if (dx > bc) SCFQ_FAIL();
        const uint32_t off = e_val(d) + (uint32_t)(bb & ((1ull << dx) - 1));
        SCFQ_DROP(dx);
        if (off > total_at_entry + (uint64_t)(out - run_start)) { result = kErrData; goto done; }   // before the start of the stream
        const T* src = out - off;
        T* dst = out;
        out += mlen;
        if (off >= (uint32_t)kPer) {
          T* const stop = dst + mlen;
          do { uint64_t w; std::memcpy(&w, src, 8); std::memcpy(dst, &w, 8); src += kPer; dst += kPer; } while (dst < stop);
        } else if (off == 1) {
          const T v = *src;
          for (uint32_t k = 0; k < mlen; ++k) dst[k] = v;
        } else {
          for (uint32_t k = 0; k < mlen; ++k) dst[k] = src[k];
        }
      }
    }
  done:
#undef SCFQ_REFILL
#undef SCFQ_DROP
#undef SCFQ_FAIL
    total_out = total_at_entry + (uint64_t)(out - run_start);
    out_ref = out;
    in_next = in;
    bitbuf = bb;
    bitcnt = bc;
    return result;
  }

  int huffman_block(uint8_t*& out_ref, uint8_t* const out_end) { return huffman_block_t<uint8_t>(out_ref, out_end); }

  // ---- symbol output (scfq_pgz.hpp): the same stream decoded into 16-bit symbols.  The caller places 32768 symbols in
  // front of the output (markers 0x8000 | k for "byte k of the window I do not know yet", or the real bytes) and sets
  // total_out = 32768, so back-references copy symbols, known or not, and are resolved later.  Stops with kAtBoundary at the
  // first block boundary whose bit position is >= stop_bit.  Plain careful loop: this path is run by many threads at once.
  int run16(uint16_t*& out_next, uint16_t* out_end) {
    for (;;) {
      if (state == kDone) return kStreamEnd;
      if (state == kHeader) {
        if (bitpos() >= stop_bit) return kAtBoundary;
        refill();
        if (bitcnt < 3) return kErrTruncated;
        last_block = peek(1); drop(1);
        const uint32_t type = peek(2); drop(2);
        if (type == 0) {
          drop((int)(bitcnt & 7));
          refill();
          if (bitcnt < 32) return kErrTruncated;
          const uint32_t len = peek(16), nlen = (uint32_t)((bitbuf >> 16) & 0xFFFF);
          if ((len ^ nlen) != 0xFFFF) return kErrData;
          drop(32);
          in_next -= bitcnt >> 3;
          bitbuf = 0; bitcnt = 0;
          stored_left = len;
          state = kStored;
        } else if (type == 1) { load_fixed(); state = kHuff; }
        else if (type == 2) { if (!read_dynamic_header()) return kErrData; state = kHuff; }
        else return kErrData;
      }
      if (state == kStored) {
        const uint64_t room = (uint64_t)(out_end - out_next), avail = (uint64_t)(in_end - in_next);
        uint64_t k = stored_left;
        if (k > room) k = room;
        if (k > avail) k = avail;
        for (uint64_t i = 0; i < k; ++i) out_next[i] = in_next[i];
        out_next += k; in_next += k; stored_left -= (uint32_t)k; total_out += k;
        if (stored_left) {
          if (in_next == in_end) return kErrTruncated;
          return kNeedOutput;
        }
        state = last_block ? kDone : kHeader;
        continue;
      }
      {
        const int r = huffman_block_t<uint16_t>(out_next, out_end);
        if (r != kBlockEnd) return r;
      }
    }
  }

  int bitcnt_error() const { return in_next >= in_end ? kErrTruncated : kErrData; }
};

}  // namespace scfq_inflate
