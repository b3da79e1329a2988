// bgzf_inflate_kernel.hpp — device-side inflate of BGZF blocks (gfx950), the stretch row of SURVEY.md §8f-1.
//
// A BGZF file (what `bgzip` writes) is a chain of gzip members of at most 64 KiB, each recording its compressed size in
// the header: the host finds the member boundaries without inflating, ships the COMPRESSED bytes over PCIe (about a quarter
// of the inflated ones) and the device inflates them straight into the buffer that fq_scan_tiles then scans.  The bytes
// are what zlib's gzread yields for the file (gzip_stream.nim:16-17 semantics); CRC-32 and ISIZE of every member are
// checked on the device.
//
// One WAVE per BGZF block.  DEFLATE is serial inside a block (every code's position depends on the previous code's
// length), so the wave walks the bit stream as one: every lane holds the same bit buffer and takes the same branches
// (the input pointer is wave-uniform, so refills are one broadcast load), the Huffman tables live in the wave's own slice
// of LDS (two-level, 16-bit entries, built by lane 0 with the construction of zlib's inftrees), literals are stored by
// lane 0, and the parallelism of the wave goes into the match copies (lane k copies byte k) and into the CRC (64 slices,
// one per lane, stitched with a precomputed GF(2) shift).  Throughput comes from the number of blocks in flight
// (256 CUs x 24 waves), not from one block being fast.
// Every loop is bounded by the block's input bits or output bytes; a malformed stream sets an error status and ends
// the wave.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace scfq_dinflate {

#ifndef SCFQ_DABLATE
#define SCFQ_DABLATE 0      // measurement builds only (scripts/gpu_dinflate_ablate.sh): 1 no match copies, 2 no CRC, 4 no literal stores
#endif

struct Block {              // offsets are relative to the chunk's compressed / inflated buffers
  uint32_t in_off;          // first byte of the member's deflate data
  uint32_t in_len;          // bytes of deflate data (member size - header - 8-byte trailer)
  uint32_t out_off;
  uint32_t isize;           // ISIZE trailer: bytes this member inflates to (<= 65536 for BGZF)
  uint32_t crc;             // CRC-32 trailer
};

enum : uint32_t { kOk = 0, kErrData = 1, kErrLength = 2, kErrCrc = 3 };
constexpr uint32_t kInvalid = 0xFFFFu;        // table slot of a code that is not assigned (bit 15 set like a second-level pointer: one test for both)

constexpr int kLitRoot = 10, kDistRoot = 8;
constexpr int kLitEntries = 2048, kDistEntries = 768;                   // first level + second-level space
constexpr int kWaveLdsHalfwords = kLitEntries + kDistEntries + 160 /*lens[320] as bytes*/ + 128 /*code-length table*/ +
                                  320 /*sorted*/ + 32 /*count, offs*/;
constexpr int kWavesPerWg = 4;

// u16 entry: bits 0..3 code length, 4..12 symbol (kInvalid = unassigned); second-level pointer: bit 15, bits 0..3 index bits,
// bits 4..14 start of the second-level table
__device__ const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__device__ const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__device__ const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__device__ const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__device__ const uint8_t kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// Canonical Huffman code -> two-level table in LDS (serial; called by lane 0 only).  lens: code length per symbol.
// Returns false for an over-subscribed code or an incomplete one that zlib rejects.
__device__ inline bool build_table(const uint8_t* lens, int n, bool is_codelen, uint16_t* tab, int root, int cap,
                                   uint16_t* sorted, uint16_t* count /*[16]*/, uint16_t* offs /*[16]*/) {
  for (int k = 0; k < 16; ++k) count[k] = 0;
  for (int s = 0; s < n; ++s) count[lens[s]]++;
  int max = 15;
  while (max >= 1 && !count[max]) --max;
  const int first = 1 << root;
  for (int k = 0; k < first; ++k) tab[k] = (uint16_t)kInvalid;
  if (max == 0) return true;                         // no codes: every lookup fails (zlib: error on use)
  int left = 1;
  for (int len = 1; len <= 15; ++len) { left = (left << 1) - (int)count[len]; if (left < 0) return false; }
  if (left > 0 && (is_codelen || max != 1)) return false;
  offs[1] = 0;
  for (int len = 1; len < 15; ++len) offs[len + 1] = (uint16_t)(offs[len] + count[len]);
  for (int s = 0; s < n; ++s) if (lens[s]) sorted[offs[lens[s]]++] = (uint16_t)s;
  int next_free = first, sym_i = 0, sub_start = 0, sub_bits = 0;
  uint32_t code = 0, cur_prefix = 0xFFFFFFFFu;
  for (int len = 1; len <= max; ++len) {
    const int cnt = count[len];
    for (int c = 0; c < cnt; ++c, ++sym_i) {
      const uint32_t sym = sorted[sym_i];
      uint32_t rev = __brev(code) >> (32 - len);
      uint16_t e = (uint16_t)((sym << 4) | (uint32_t)len);
      if (len <= root) {
        for (uint32_t k = rev; k < (uint32_t)first; k += 1u << len) tab[k] = e;
      } else {
        const uint32_t prefix = rev & (uint32_t)(first - 1);
        if (prefix != cur_prefix) {
          int curr = len - root;
          int cnt_left = (1 << curr) - (cnt - c);
          int ll = len;
          while (cnt_left > 0 && ll < max) { ++ll; ++curr; cnt_left = (cnt_left << 1) - (int)count[ll]; }
          sub_bits = curr;
          sub_start = next_free;
          next_free += 1 << sub_bits;
          if (next_free > cap) return false;
          for (int k = sub_start; k < next_free; ++k) tab[k] = (uint16_t)kInvalid;
          tab[prefix] = (uint16_t)(0x8000u | ((uint32_t)sub_start << 4) | (uint32_t)sub_bits);
          cur_prefix = prefix;
        }
        const int hl = len - root;
        e = (uint16_t)((sym << 4) | (uint32_t)hl);
        for (uint32_t k = rev >> root; k < (1u << sub_bits); k += 1u << hl) tab[sub_start + k] = e;
      }
      ++code;
    }
    code <<= 1;
  }
  return true;
}

// CRC-32 (IEEE, reflected polynomial 0xEDB88320).  Every lane takes the standard CRC of one contiguous slice; slices are
// stitched the way zlib's crc32_combine does it: crc(A || B) = crc(A) * x^(8|B|) mod P  xor  crc(B), the product in
// GF(2)[x] with bit 31 the coefficient of x^0.
__device__ inline uint32_t crc_byte(uint32_t c, uint32_t b) {
  c ^= b;
#pragma unroll
  for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
  return c;
}
__device__ inline uint32_t gf2_mulmod(uint32_t a, uint32_t b) {
  uint32_t p = 0;
  for (int k = 0; k < 32; ++k) {
    p ^= b & (0u - ((a >> (31 - k)) & 1u));
    b = (b >> 1) ^ (0xEDB88320u & (0u - (b & 1u)));
  }
  return p;
}
__device__ inline uint32_t x_pow_8n(uint32_t n) {      // x^(8 n) mod P, n < 2^20
  uint32_t p = 1u << 31, sq = 1u << 30;                // x^0, x^1
  for (uint32_t bits = n << 3; bits; bits >>= 1) {
    if (bits & 1u) p = gf2_mulmod(sq, p);
    sq = gf2_mulmod(sq, sq);
  }
  return p;
}

// An LDS read whose address is wave-uniform returns the same value in every lane; saying so (v_readfirstlane) lets the
// compiler keep everything computed from it — the bit buffer, positions, every branch — on the scalar unit instead of
// doing it 64 times over on the vector ALU with exec-mask bookkeeping around every check.
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

__global__ __launch_bounds__(64 * kWavesPerWg) void bgzf_inflate(const uint8_t* __restrict__ comp, const Block* __restrict__ blocks,
                                                                uint32_t n_blocks, uint8_t* out, uint32_t* status /* one word, OR of (1 << error) */) {
  extern __shared__ uint16_t lds[];
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t b = blockIdx.x * kWavesPerWg + wave;
  uint16_t* lit = lds + wave * kWaveLdsHalfwords;
  uint16_t* dist = lit + kLitEntries;
  uint8_t* lens = reinterpret_cast<uint8_t*>(dist + kDistEntries);      // 320 bytes
  uint16_t* cltab = dist + kDistEntries + 160;
  uint16_t* sorted = cltab + 128;
  uint16_t* count = sorted + 320;
  uint16_t* offs = count + 16;
  __shared__ uint32_t build_ok[kWavesPerWg];
  // length / distance base and extra-bit tables: LDS copies (a lookup in the global-memory constants costs an L2 round
  // trip per match)
  __shared__ uint32_t s_len[32], s_dist[32];           // base | extra bits << 16
  if (threadIdx.x < 29) s_len[threadIdx.x] = kLenBase[threadIdx.x] | ((uint32_t)kLenExtra[threadIdx.x] << 16);
  if (threadIdx.x >= 30 && threadIdx.x < 32) s_dist[threadIdx.x] = 0;
  if (threadIdx.x < 30) s_dist[threadIdx.x] = kDistBase[threadIdx.x] | ((uint32_t)kDistExtra[threadIdx.x] << 16);
  __syncthreads();
  if (b >= n_blocks) return;

  const Block blk = blocks[b];
  const uint8_t* const in = comp + blk.in_off;      // wave-uniform base; the read position is the 32-bit offset `ip`
  const uint32_t ip_end = blk.in_len;
  uint32_t ip = 0;
  uint8_t* const o = out + blk.out_off;
  const uint32_t isize = blk.isize;
  // the member's output as a buffer descriptor: an offset outside [0, isize) is dropped (store) or reads 0 (load)
  const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(o, 0, (int)isize, 0x00020000);
  const uint32_t not_lane0 = lane == 0 ? 0u : 0xFFFFFFFFu;
  uint64_t bb = 0;
  int32_t bc = 0;                                      // bits in bb; negative once a malformed stream has run past its end
  uint32_t pos = 0, err = kOk;
#ifdef SCFQ_DSTATS
  uint32_t n_blk = 0, n_lit = 0, n_match = 0, n_mbytes = 0, n_overlap = 0, n_long = 0;
#endif
  bool last = false;

  // the 8-byte gzip trailer follows the deflate data, so an 8-byte load that starts inside the data stays in the chunk.
  // (A macro, not a lambda: captured-by-reference state ended up in scratch memory, which made every value derived from
  // it per-lane.)
#define SCFQ_DREFILL()                                                                         \
  do {                                                                                         \
    if (ip < ip_end) {                                                                         \
      uint64_t w_;                                                                             \
      __builtin_memcpy(&w_, in + ip, 8);                                                       \
      w_ = ((uint64_t)uni((uint32_t)(w_ >> 32)) << 32) | uni((uint32_t)w_);                    \
      const int32_t avail_ = (int32_t)(ip_end - ip);                                           \
      if (avail_ < 8) w_ &= (1ull << (8 * avail_)) - 1;                                        \
      bb |= w_ << bc;                                                                          \
      int32_t take_ = (63 - bc) >> 3;                                                          \
      if (take_ > avail_) take_ = avail_;                                                      \
      ip += (uint32_t)take_;                                                                   \
      bc += take_ * 8;                                                                         \
    }                                                                                          \
  } while (0)

  uint32_t guard = 0;                                  // every pass of every loop below consumes input bits or ends
  while (!last && err == kOk) {
    SCFQ_DREFILL();
    if (bc < 3) { err = kErrData; break; }
    last = bb & 1;
    const uint32_t type = (uint32_t)(bb >> 1) & 3u;
    bb >>= 3; bc -= 3;
    if (type == 0) {
      // stored: to the byte boundary, LEN / NLEN, then a plain copy (all lanes)
      const uint32_t drop = bc & 7; bb >>= drop; bc -= drop;
      SCFQ_DREFILL();
      if (bc < 32) { err = kErrData; break; }
      const uint32_t len = (uint32_t)bb & 0xFFFF, nlen = (uint32_t)(bb >> 16) & 0xFFFF;
      bb >>= 32; bc -= 32;
      if ((len ^ nlen) != 0xFFFF) { err = kErrData; break; }
      ip -= (uint32_t)bc >> 3;                                 // whole bytes go back to the byte stream
      bb = 0; bc = 0;
      if (ip_end - ip < len || pos + len > isize) { err = kErrData; break; }
      for (uint32_t k = lane; k < len; k += 64) o[pos + k] = in[ip + k];
      ip += len; pos += len;
      continue;
    }
    if (type == 3) { err = kErrData; break; }
    // ---- code tables into this wave's LDS (lane 0; the other lanes wait at the LDS dependency) ----------------------
    if (type == 1) {
      if (lane == 0) {
        for (int k = 0; k < 144; ++k) lens[k] = 8;
        for (int k = 144; k < 256; ++k) lens[k] = 9;
        for (int k = 256; k < 280; ++k) lens[k] = 7;
        for (int k = 280; k < 288; ++k) lens[k] = 8;
        bool ok = build_table(lens, 288, false, lit, kLitRoot, kLitEntries, sorted, count, offs);
        for (int k = 0; k < 32; ++k) lens[k] = 5;
        ok = ok && build_table(lens, 32, false, dist, kDistRoot, kDistEntries, sorted, count, offs);
        build_ok[wave] = ok ? 1u : 0u;
      }
    } else {
      SCFQ_DREFILL();
      if (bc < 14) { err = kErrData; break; }
      const uint32_t hlit = ((uint32_t)bb & 31) + 257, hdist = ((uint32_t)(bb >> 5) & 31) + 1, hclen = ((uint32_t)(bb >> 10) & 15) + 4;
      bb >>= 14; bc -= 14;
      if (hlit > 286 || hdist > 30) { err = kErrData; break; }
      // the 19 code-length code lengths, then the run-length coded literal/length + distance code lengths: all lanes
      // walk the bits together, lane 0 writes
      if (lane == 0) for (int k = 0; k < 19; ++k) lens[k] = 0;
      for (uint32_t k = 0; k < hclen; ++k) {
        SCFQ_DREFILL();
        if (bc < 3) { err = kErrData; break; }
        if (lane == 0) lens[kClOrder[k]] = (uint8_t)(bb & 7);
        bb >>= 3; bc -= 3;
      }
      if (err) break;
      if (lane == 0) build_ok[wave] = build_table(lens, 19, true, cltab, 7, 128, sorted, count, offs) ? 1u : 0u;
      __builtin_amdgcn_wave_barrier();
      if (!__builtin_amdgcn_readfirstlane((int)build_ok[wave])) { err = kErrData; break; }
      uint32_t k = 0, prev = 0;
      const uint32_t total = hlit + hdist;
      while (k < total) {
        if (++guard > (1u << 20)) { err = kErrData; break; }
        SCFQ_DREFILL();
        const uint32_t e = uni(cltab[bb & 127]);
        const uint32_t len = e & 15;
        if (e == kInvalid || (int32_t)len > bc) { err = kErrData; break; }
        bb >>= len; bc -= (int32_t)len;
        const uint32_t sym = e >> 4;
        uint32_t rep = 1, val = sym;
        if (sym == 16) { if (k == 0 || bc < 2) { err = kErrData; break; } val = prev; rep = 3 + ((uint32_t)bb & 3); bb >>= 2; bc -= 2; }
        else if (sym == 17) { if (bc < 3) { err = kErrData; break; } val = 0; rep = 3 + ((uint32_t)bb & 7); bb >>= 3; bc -= 3; }
        else if (sym == 18) { if (bc < 7) { err = kErrData; break; } val = 0; rep = 11 + ((uint32_t)bb & 127); bb >>= 7; bc -= 7; }
        if (k + rep > total) { err = kErrData; break; }
        // the lengths go to a second area (bytes 32..351 would collide with nothing: lens has 320 bytes, the code-length
        // lengths above are dead once cltab is built)
        if (lane == 0) for (uint32_t r = 0; r < rep; ++r) lens[k + r] = (uint8_t)val;
        k += rep;
        prev = val;
      }
      if (err) break;
      if (lane == 0) {
        bool ok = lens[256] != 0;
        // distance lengths follow the literal/length ones: build dist first from lens + hlit, then lit (build uses `sorted`)
        ok = ok && build_table(lens + hlit, (int)hdist, false, dist, kDistRoot, kDistEntries, sorted, count, offs);
        ok = ok && build_table(lens, (int)hlit, false, lit, kLitRoot, kLitEntries, sorted, count, offs);
        build_ok[wave] = ok ? 1u : 0u;
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (!__builtin_amdgcn_readfirstlane((int)build_ok[wave])) { err = kErrData; break; }
    // ---- symbols --------------------------------------------------------------------------------------------------
    // The loop holds no divergent branch: stores are predicated through the buffer descriptor (a lane that must not write
    // gets an out-of-range offset, which the hardware drops), so every branch below is a scalar branch on wave-uniform
    // state and the compiler emits the loop as written (with `if (lane == 0)` regions inside, the CFG structuriser wrapped
    // every exit of the loop in a state machine: about 35 scalar instructions per literal on top of the decode).
    // Checks are lazy where the hardware already bounds the access: writes beyond ISIZE are dropped by the descriptor
    // and found by `pos != isize` afterwards; bits consumed beyond the member's end make `bc` negative, which the refill
    // branch (taken on every pass once the input is exhausted) turns into an error.
    // Written with ONE exit at the bottom (`done`) and if/else instead of break/continue: a loop with several exits is
    // rewritten by the compiler's exit unification into the same kind of state machine.
    uint32_t done = 0;
#ifdef SCFQ_DSTATS
    ++n_blk;
#endif
    do {
      if (bc < 48) {
        SCFQ_DREFILL();
        if (bc < 0) { err = kErrData; done = 1; }
        if (pos > isize) { err = kErrLength; done = 1; }
      }
      uint32_t e = uni(lit[bb & ((1u << kLitRoot) - 1)]);
      if (e & 0x8000u) {                     // second level, or an unassigned code (kInvalid carries the same flag)
        if (e != kInvalid) {
          e = uni(lit[((e >> 4) & 0x7FF) + ((uint32_t)(bb >> kLitRoot) & ((1u << (e & 15)) - 1))]);
          bb >>= kLitRoot; bc -= kLitRoot;
        }
      }
      const uint32_t len = e & 15;           // kInvalid: 15 bits and "symbol" 0xFFF, which the range check below rejects
      bb >>= len; bc -= (int32_t)len;
      const uint32_t sym = e >> 4;
      if (sym < 256) {
        if (!(SCFQ_DABLATE & 4)) __builtin_amdgcn_raw_buffer_store_b8((uint8_t)sym, orsrc, pos | not_lane0, 0, 0);
        ++pos;
#ifdef SCFQ_DSTATS
        ++n_lit;
#endif
      } else if (sym == 256) {
        if (bc < 0) err = kErrData;
        done = 1;
      } else if (sym > 285) {
        err = kErrData; done = 1;
      } else {
        const uint32_t lb = uni(s_len[sym - 257]);               // base | extra bits << 16
        const uint32_t lx = lb >> 16;
        const uint32_t mlen = (lb & 0xFFFF) + ((uint32_t)bb & ((1u << lx) - 1));
        bb >>= lx; bc -= (int32_t)lx;
        if (bc < 28) SCFQ_DREFILL();         // distance code + extra bits: at most 15 + 13
        uint32_t d = uni(dist[bb & ((1u << kDistRoot) - 1)]);
        if (d & 0x8000u) {
          if (d != kInvalid) {
            d = uni(dist[((d >> 4) & 0x7FF) + ((uint32_t)(bb >> kDistRoot) & ((1u << (d & 15)) - 1))]);
            bb >>= kDistRoot; bc -= kDistRoot;
          }
        }
        const uint32_t dl = d & 15;
        bb >>= dl; bc -= (int32_t)dl;
        const uint32_t dsym = d >> 4;
        const uint32_t db = uni(s_dist[dsym & 31]);              // base | extra bits << 16 (entries 30, 31: zero)
        const uint32_t dx = db >> 16;
        const uint32_t off = (db & 0xFFFF) + ((uint32_t)bb & ((1u << dx) - 1));
        bb >>= dx; bc -= (int32_t)dx;
        if (dsym >= 30 || off > pos) {       // (a BGZF member starts with an empty window)
          err = kErrData; done = 1;
        } else {
          // The bytes the match reads were stored by this same wave (lane 0's literals, other lanes' earlier copies).
          // Vector memory instructions of one wave reach the CU's L1 in issue order and the write-through L1 serves later
          // loads of the same CU coherently (the workgroup-scope rule of the AMDGPU memory model: no cache maintenance
          // inside a CU), so no s_waitcnt vmcnt(0) is needed here.  SCFQ_DINFLATE_FENCE builds the conservative form
          // (measured: same speed); every member's CRC-32 is verified on the device either way.
#ifdef SCFQ_DINFLATE_FENCE
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#endif
          const uint32_t src0 = pos - off;
          if (SCFQ_DABLATE & 1) {
          } else if (off >= mlen) {
            for (uint32_t base = 0; base < mlen; base += 64) {
              const uint32_t k = base + lane;
              const uint8_t v = __builtin_amdgcn_raw_buffer_load_b8(orsrc, src0 + k, 0, 1 /*sc0*/);
              __builtin_amdgcn_raw_buffer_store_b8(v, orsrc, k < mlen ? pos + k : 0xFFFFFFFFu, 0, 0);
            }
          } else {                           // the match overlaps its own output: period `off`
            for (uint32_t base = 0; base < mlen; base += 64) {
              const uint32_t k = base + lane;
              const uint32_t j = off == 1 ? 0u : k % off;
              const uint8_t v = __builtin_amdgcn_raw_buffer_load_b8(orsrc, src0 + j, 0, 1 /*sc0*/);
              __builtin_amdgcn_raw_buffer_store_b8(v, orsrc, k < mlen ? pos + k : 0xFFFFFFFFu, 0, 0);
            }
          }
          pos += mlen;
#ifdef SCFQ_DSTATS
          ++n_match; n_mbytes += mlen; n_overlap += off < mlen; n_long += mlen > 64;
#endif
        }
      }
    } while (!done);
  }
  if (err == kOk && pos != isize) err = kErrLength;
  // ---- CRC-32 of the member: every lane the standard CRC of a contiguous slice, then 63 combines ------------------------
  if (err == kOk && !(SCFQ_DABLATE & 2)) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const uint32_t per = (isize + 63) / 64;
    const uint32_t lo = lane * per < isize ? lane * per : isize, hi = lo + per < isize ? lo + per : isize;
    uint32_t c = 0xFFFFFFFFu;
    for (uint32_t k = lo; k < hi; ++k) c = crc_byte(c, __hip_atomic_load(o + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    c ^= 0xFFFFFFFFu;                                   // the CRC of an empty slice is 0
    const uint32_t shift_full = x_pow_8n(per);
    uint32_t total = 0;
    for (uint32_t l = 0; l < 64; ++l) {
      const uint32_t c_l = (uint32_t)__builtin_amdgcn_readlane((int)c, (int)l);
      const uint32_t lo_l = l * per < isize ? l * per : isize, hi_l = lo_l + per < isize ? lo_l + per : isize;
      const uint32_t n_l = hi_l - lo_l;
      if (n_l == 0) continue;
      total = gf2_mulmod(n_l == per ? shift_full : x_pow_8n(n_l), total) ^ c_l;
    }
    if (total != blk.crc) err = kErrCrc;
  }
#undef SCFQ_DREFILL
#ifdef SCFQ_DSTATS
  if (lane == 0) { atomicAdd(status + 1, n_blk); atomicAdd(status + 2, n_lit); atomicAdd(status + 3, n_match); atomicAdd(status + 4, n_mbytes >> 4); atomicAdd(status + 5, n_overlap); atomicAdd(status + 6, n_long); }
#endif
  if (lane == 0 && err) atomicOr(status, 1u << err);      // one word for the whole launch: bit k = some block ended with error k
}

}  // namespace scfq_dinflate
