// bgzf_inflate_kernel.hpp — device-side inflate of BGZF blocks (gfx950), the stretch row of SURVEY.md §8f-1.
//
// A BGZF file (what `bgzip` writes) is a chain of gzip members of at most 64 KiB, each recording its compressed size in
// the header: the host finds the member boundaries without inflating, ships the COMPRESSED bytes over PCIe (about a quarter
// of the inflated ones) and the device inflates them straight into the buffer that fq_scan_tiles then scans.  The bytes
// are what zlib's gzread yields for the file (gzip_stream.nim:16-17 semantics); CRC-32 and ISIZE of every member are
// checked on the device.
//
// One WAVE per BGZF block.  DEFLATE is serial inside a block (every code's position depends on the previous code's
// length), and throughput comes from the number of blocks in flight (256 CUs x 16 waves) — but a wave does not have to walk
// the stream with 63 lanes idle.  The symbols of a Huffman block are decoded by symbol_loop_lanes below: every lane decodes
// the symbol that WOULD start at one of the next 64 bit positions (the Huffman tables live in the wave's own slice of LDS:
// two-level, 32-bit entries that already hold base value and extra-bit count, built by lane 0 with the construction of
// zlib's inftrees), the scalar unit only follows the chain of the real ones, and the output bytes of a round are produced by
// one load and one store instruction.  Block headers and code lengths are read by a branch-free bit reader on the scalar unit
// (compressed bytes through the scalar cache); stored blocks are lane-parallel copies; the member's CRC-32 runs four bytes
// per step through LDS tables, 64 lane slices stitched in a tree.  Level-6 FASTQ is 70 % matches (mean length 10), i.e.
// ~9 K symbols per 64 KiB member.  r1's serial symbol loop (one symbol per pass on the scalar unit, ~40 scalar instructions
// per literal and ~90 per match, every match a round trip to the L2) is kept as symbol_loop for A/B runs
// (SCFQ_INFLATE_LOOP=serial): 45-49 GB/s against 98 GB/s.
// Every loop is bounded by the block's input bits or output bytes; a malformed stream sets an error status and ends
// the wave.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace scfq_dinflate {

#ifndef SCFQ_DABLATE
#define SCFQ_DABLATE 0      // measurement builds only (scripts/gpu_dinflate_ablate.sh): 1 no match copies, 2 no CRC, 4 no literal stores (serial loop), 8 / 16 ten extra scalar / vector instructions per symbol (serial loop) or round (lane-parallel loop)
#endif

struct Block {              // offsets are relative to the chunk's compressed / inflated buffers
  uint32_t in_off;          // first byte of the member's deflate data
  uint32_t in_len;          // bytes of deflate data (member size - header - 8-byte trailer)
  uint32_t out_off;
  uint32_t isize;           // ISIZE trailer: bytes this member inflates to (<= 65536 for BGZF)
  uint32_t crc;             // CRC-32 trailer
};

enum : uint32_t { kOk = 0, kErrData = 1, kErrLength = 2, kErrCrc = 3 };

// Decode tables: two levels, 32-bit entries with everything a symbol needs already in them (no second lookup for the
// base / extra bits of a length or distance code):
//   bits 0..3   bits this level consumes          bits 4..7   number of extra bits (length / distance codes),
//                                                              or index bits of the second-level table (kSub)
//   bits 8..11  kind: kLit, kEob, kVal (a length, a distance, a code-length symbol), kSub; none set = unassigned code
//   bits 16..31 literal byte / base length / base distance / code-length symbol / start of the second-level table
constexpr uint32_t kLit = 1u << 8, kEob = 1u << 9, kVal = 1u << 10, kSub = 1u << 11;
constexpr int kLitRoot = 10, kDistRoot = 8;
constexpr int kLitEntries = 1344, kDistEntries = 416;   // first + second level: `enough 288 10 15` = 1334, `enough 32 8 15` = 402
// A wave's LDS.  The code-length table (128 entries, needed only until the last code length of a block has been read) lies where the distance
// table is built afterwards (r3 for gz_segment_decode, r5 for bgzf_inflate): 8064 bytes per wave, 32 256 per workgroup of four.
constexpr int kWaveLdsBytes = 4 * (kLitEntries + kDistEntries) + 320 /*lens*/ + 2 * (320 /*sorted*/ + 32 /*count, offs*/);
constexpr int kWavesPerWg = 4;      // five workgroups = 20 waves per CU (LDS: 5 x (32 256 + a few hundred bytes of static tables) <= 160 KiB)
enum : int { kKindCodeLen = 0, kKindLitLen = 1, kKindDist = 2 };

__device__ const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__device__ const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__device__ const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__device__ const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__device__ const uint8_t kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// s_len / s_dist: the tables above as LDS words, base << 16 | extra bits << 4 | kVal
__device__ inline uint32_t make_entry(int kind, uint32_t sym, uint32_t nbits, const uint32_t* s_len, const uint32_t* s_dist) {
  if (kind == kKindCodeLen) return kVal | (sym << 16) | nbits;
  if (kind == kKindLitLen) {
    if (sym < 256) return kLit | (sym << 16) | nbits;
    if (sym == 256) return kEob | nbits;
    if (sym < 286) return s_len[sym - 257] | nbits;
    return nbits;                                    // 286, 287: part of the fixed code, never valid in data
  }
  return sym < 30 ? (s_dist[sym] | nbits) : nbits;
}

// Canonical Huffman code -> two-level table in LDS (serial; called by lane 0 only).  lens: code length per symbol.
// Returns false for an over-subscribed code or an incomplete one that zlib rejects.
__device__ __attribute__((noinline)) bool build_table(const uint8_t* lens, int n, int kind, uint32_t* tab, int root, int cap,
                                   uint16_t* sorted, uint16_t* count /*[16]*/, uint16_t* offs /*[16]*/,
                                   const uint32_t* s_len, const uint32_t* s_dist) {
  for (int k = 0; k < 16; ++k) count[k] = 0;
  for (int s = 0; s < n; ++s) count[lens[s]]++;
  int max = 15;
  while (max >= 1 && !count[max]) --max;
  const int first = 1 << root;
  for (int k = 0; k < first; ++k) tab[k] = 0;
  if (max == 0) return true;                         // no codes: every lookup fails (zlib: error on use)
  int left = 1;
  for (int len = 1; len <= 15; ++len) { left = (left << 1) - (int)count[len]; if (left < 0) return false; }
  if (left > 0 && (kind == kKindCodeLen || max != 1)) return false;
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
      if (len <= root) {
        const uint32_t e = make_entry(kind, sym, (uint32_t)len, s_len, s_dist);
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
          for (int k = sub_start; k < next_free; ++k) tab[k] = 0;
          tab[prefix] = kSub | ((uint32_t)sub_start << 16) | ((uint32_t)sub_bits << 4);
          cur_prefix = prefix;
        }
        const int hl = len - root;
        const uint32_t e = make_entry(kind, sym, (uint32_t)hl, s_len, s_dist);
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

// ---- the symbol loop of one Huffman block, shared by bgzf_inflate (bytes) and gz_segment_decode (16-bit symbols) -------------
// A separate, NOT inlined function: inside the kernels the loop inherited the register pressure of everything around it
// (header parsing, table construction, CRC) — 350 spilled SGPRs, among them the prefetched input dwords, which the compiler
// then had to WAIT for right after requesting them in order to spill them (the prefetch hid nothing).  As a function the loop
// has the scalar registers to itself: its state comes in and goes out through a small struct in private memory, once per
// deflate block.
//
// The loop holds no divergent branch: stores are predicated through the buffer descriptor (a lane that must not write gets an
// out-of-range offset, which the hardware drops), so every branch below is a scalar branch on wave-uniform state and the
// compiler emits the loop as written (with `if (lane == 0)` regions inside, the CFG structuriser wrapped every exit of the
// loop in a state machine: about 35 scalar instructions per literal on top of the decode).  For the same reason it has ONE
// exit at the bottom and if/else instead of break/continue.
// One refill per pass is enough: a literal/length code, its extra bits, a distance code and its extra bits are at most
// 15 + 5 + 15 + 13 = 48 of the 56 bits a refill guarantees.
// Checks are lazy where the hardware already bounds the access: writes beyond the output's end are dropped by the descriptor
// and end the loop (pos > limit); bits taken beyond the input's end are found by the position check after the loop.
// A match's store is DEFERRED: the load is issued, the wave goes on decoding, and the store follows when the next match is
// about to load (it must be in the memory pipeline before a load that may read those bytes) or when the block ends, so the
// load's latency overlaps the next symbols instead of stalling the wave.  pend_off is out of range in every lane while nothing
// is pending.  The bytes a match reads were stored by this same wave (lane 0's literals, other lanes' earlier copies): vector
// memory instructions of one wave reach the CU's L1 in issue order and the write-through L1 serves later loads of the same CU
// coherently, so no s_waitcnt vmcnt(0) is needed; every member's CRC-32 is verified on the device either way.
// The table addresses are formed on the vector ALU (lit_v / dist_v look per-lane to the compiler): the loop is bound by the
// CU's one scalar ALU, every instruction moved off it counts.
struct SymState {
  uint64_t bb;               // bit buffer
  uint32_t bc, ip, pos, err; // bits in bb | bytes consumed | output position (symbols) | error (kErrData / kErrLength)
  uint32_t pf[4], pf_sh;     // the input dwords requested one refill ahead, and their byte shift
};

typedef uint32_t dword4_t __attribute__((ext_vector_type(4)));
typedef const dword4_t __attribute__((address_space(4), aligned(4))) const_dword4_t;
typedef const uint8_t __attribute__((address_space(4))) const_byte_t;

// SYM16: output symbols are 16 bits wide (gz_segment_decode), else bytes.  limit: output symbols the buffer holds.
template <bool SYM16>
__device__ __attribute__((noinline)) void symbol_loop(SymState* stp, const uint8_t* in_aligned /* 4-byte aligned base of the scalar loads */,
                                                     uint32_t in_off, uint32_t ip_end, void* out_base, uint32_t limit,
                                                     uint32_t lit_lds, uint32_t dist_lds /* LDS byte offsets of the two tables */) {
  const uint32_t lane = threadIdx.x & 63;
  // arguments of a device function arrive in vector registers: say that they are wave-uniform, so that the input goes through
  // the scalar cache and the descriptor is built from scalars
  const uint64_t in_u = ((uint64_t)uni((uint32_t)((uintptr_t)in_aligned >> 32)) << 32) | uni((uint32_t)(uintptr_t)in_aligned);
  const uint64_t out_u = ((uint64_t)uni((uint32_t)((uintptr_t)out_base >> 32)) << 32) | uni((uint32_t)(uintptr_t)out_base);
  const_byte_t* const in4 = (const_byte_t*)in_u;
  out_base = (void*)out_u;
  in_off = uni(in_off);
  ip_end = uni(ip_end);
  limit = uni(limit);
  uint64_t bb = ((uint64_t)uni((uint32_t)(stp->bb >> 32)) << 32) | uni((uint32_t)stp->bb);
  uint32_t bc = uni(stp->bc), ip = uni(stp->ip), pos = uni(stp->pos), err = kOk;
  dword4_t pf;
  pf.x = uni(stp->pf[0]); pf.y = uni(stp->pf[1]); pf.z = uni(stp->pf[2]); pf.w = uni(stp->pf[3]);
  uint32_t pf_sh = uni(stp->pf_sh);
  const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(out_base, 0, (int)(limit * (SYM16 ? 2u : 1u)), 0x00020000);
  const uint32_t not_lane0 = lane == 0 ? 0u : 0xFFFFFFFFu;
  uint32_t lane_zero;                                  // 0 in every lane, opaque to the compiler's uniformity analysis
  asm volatile("v_mov_b32 %0, 0" : "=v"(lane_zero));
  // (LDS address space, so that the table reads are ds_read and wait for lgkmcnt only: as generic pointers they were flat loads
  // that also waited for every outstanding store)
  typedef __attribute__((address_space(3))) uint32_t lds_u32;
  lds_u32* const lit_v = (lds_u32*)(uintptr_t)uni(lit_lds) + lane_zero;
  lds_u32* const dist_v = (lds_u32*)(uintptr_t)uni(dist_lds) + lane_zero;
#define SCFQ_LPREFETCH()                                                                       \
  do {                                                                                         \
    const uint32_t a_ = in_off + (ip < ip_end ? ip : ip_end);                                  \
    pf_sh = (a_ & 3u) << 3;                                                                    \
    pf = *(const_dword4_t*)(in4 + (a_ & ~3u));                                                 \
  } while (0)
#define SCFQ_LREFILL()                                                                         \
  do {                                                                                         \
    const uint32_t up_ = 31u - pf_sh;           /* (x << 1) << up_ == x << (32 - pf_sh), also for pf_sh == 0 */ \
    const uint32_t lo_ = (pf.x >> pf_sh) | ((pf.y << 1) << up_), hi_ = (pf.y >> pf_sh) | ((pf.z << 1) << up_); \
    bb |= (((uint64_t)hi_ << 32) | lo_) << bc;                                                 \
    ip += (63u - bc) >> 3;                                                                     \
    bc |= 56u;                                                                                 \
    SCFQ_LPREFETCH();                                                                          \
  } while (0)
#define SCFQ_LTAKE(n_) do { bb >>= (n_); bc -= (n_); } while (0)
  auto store_sym = [&](uint32_t v, uint32_t off) {
    if (SYM16) __builtin_amdgcn_raw_buffer_store_b16((uint16_t)v, orsrc, off, 0, 0);
    else __builtin_amdgcn_raw_buffer_store_b8((uint8_t)v, orsrc, off, 0, 0);
  };
  auto load_sym = [&](uint32_t off) -> uint32_t {
    if (SYM16) return __builtin_amdgcn_raw_buffer_load_b16(orsrc, off, 0, 1 /*sc0*/);
    return __builtin_amdgcn_raw_buffer_load_b8(orsrc, off, 0, 1 /*sc0*/);
  };
  constexpr uint32_t kSh = SYM16 ? 1u : 0u;            // symbol index -> byte offset
  uint32_t done = 0;
  uint32_t pend_off = 0xFFFFFFFFu;
  uint32_t pend_v = 0;
  do {
    if (bc < 48) SCFQ_LREFILL();           // (the prefetched dwords stay valid: ip only moves in a refill)
    if (SCFQ_DABLATE & 8) {                // measurement only: ten more scalar instructions per symbol
      uint32_t t_ = pos;
#pragma unroll
      for (int q_ = 0; q_ < 10; ++q_) asm volatile("s_add_u32 %0, %0, 1" : "+s"(t_) : : "scc");
      asm volatile("" : : "s"(t_));
    }
    if (SCFQ_DABLATE & 16) {               // measurement only: ten more vector instructions per symbol
      uint32_t t_ = lane;
#pragma unroll
      for (int q_ = 0; q_ < 10; ++q_) asm volatile("v_add_u32 %0, %0, 1" : "+v"(t_));
      asm volatile("" : : "v"(t_));
    }
    uint32_t e = uni(lit_v[bb & ((1u << kLitRoot) - 1)]);
    if (e & kSub) {
      SCFQ_LTAKE(kLitRoot);
      e = uni(lit_v[(e >> 16) + ((uint32_t)bb & ((1u << ((e >> 4) & 15)) - 1))]);
    }
    SCFQ_LTAKE(e & 15);
    if (e & kLit) {
      if (!(SCFQ_DABLATE & 4)) store_sym(e >> 16, (pos << kSh) | not_lane0);
      ++pos;
    } else if (e & kVal) {
      const uint32_t lx = (e >> 4) & 15;
      const uint32_t mlen = (e >> 16) + ((uint32_t)bb & ((1u << lx) - 1));
      SCFQ_LTAKE(lx);
      uint32_t d = uni(dist_v[bb & ((1u << kDistRoot) - 1)]);
      if (d & kSub) {
        SCFQ_LTAKE(kDistRoot);
        d = uni(dist_v[(d >> 16) + ((uint32_t)bb & ((1u << ((d >> 4) & 15)) - 1))]);
      }
      SCFQ_LTAKE(d & 15);
      const uint32_t dx = (d >> 4) & 15;
      const uint32_t off = (d >> 16) + ((uint32_t)bb & ((1u << dx) - 1));
      SCFQ_LTAKE(dx);
      if (off - 1u >= pos) {               // off > pos (nothing that far back), or off == 0: the entry of an unassigned
                                           // distance code has base 0 and no extra bits
        err = kErrData; done = 1;
      } else {
        const uint32_t src0 = pos - off;
        store_sym(pend_v, pend_off);
        pend_off = 0xFFFFFFFFu;
        if (SCFQ_DABLATE & 1) {
        } else if (mlen <= 64) {           // one pass of the wave (nested ifs: a combined condition costs the scalar ALU more)
          if (off >= mlen) {
            pend_v = load_sym((src0 + lane) << kSh);
            pend_off = lane < mlen ? (pos + lane) << kSh : 0xFFFFFFFFu;
          } else {                         // the match overlaps its own output: period `off`
            const uint32_t j = off == 1 ? 0u : lane % off;
            const uint32_t v = load_sym((src0 + j) << kSh);
            store_sym(v, lane < mlen ? (pos + lane) << kSh : 0xFFFFFFFFu);
          }
        } else {
          for (uint32_t base = 0; base < mlen; base += 64) {
            const uint32_t k = base + lane;
            const uint32_t j = off >= mlen ? k : (off == 1 ? 0u : k % off);
            const uint32_t v = load_sym((src0 + j) << kSh);
            store_sym(v, k < mlen ? (pos + k) << kSh : 0xFFFFFFFFu);
          }
        }
        pos += mlen;
      }
    } else {                               // end of block, or a code that is not assigned
      if (!(e & kEob)) err = kErrData;
      done = 1;
    }
    // (sign-bit arithmetic keeps both tests on the scalar unit; all four values are below 2^30, the callers see to it)
    done |= (limit - pos) >> 31;           // pos > limit: the descriptor dropped the excess, the stream is malformed (or the room too small)
    done |= (ip_end + 16u - ip) >> 31;     // a malformed stream reading (clamped) bytes far past the end of the data
  } while (!done);
  store_sym(pend_v, pend_off);
#undef SCFQ_LREFILL
#undef SCFQ_LPREFETCH
#undef SCFQ_LTAKE
  // (private memory is per lane: every lane stores the same wave-uniform values into its own copy)
  stp->bb = bb; stp->bc = bc; stp->ip = ip; stp->pos = pos; stp->err = err;
  stp->pf[0] = pf.x; stp->pf[1] = pf.y; stp->pf[2] = pf.z; stp->pf[3] = pf.w; stp->pf_sh = pf_sh;
}

// ---- the symbol loop, lane-parallel form ---------------------------------------------------------------------------------
// The serial loop above spends a wave's time in chains of dependent round trips — per symbol the LDS look-ups, per match (70 % of
// the symbols) a load of bytes this wave has just stored, which the L1 does not hold (stores write through and do not allocate):
// a round trip to the L2 with the wave waiting — and 63 lanes idle.  Here the lanes do the work, in ROUNDS over the next 64
// bits of the stream:
//  1. decode: lane i assumes that a symbol starts at bit i and decodes it completely — literal/length code, extra bits, distance
//     code, extra bits: four LDS gathers and ~50 vector instructions for all 64 positions at once;
//  2. walk: the scalar unit only FOLLOWS the chain of real symbols (0 -> next[0] -> next[next[0]] ..., two v_readlane per symbol,
//     4-5 symbols per round for level-6 FASTQ), noting each symbol at the lane of its first OUTPUT byte;
//  3. emit: lane t is output byte t of the round.  It finds its symbol (highest noted start at or below t), and is a literal
//     or byte t - start of a match, whose source is simply `distance` bytes back: ONE load and ONE store instruction for the whole
//     round.  The store waits for the load only at the NEXT round's emit, after that round's decode and walk, so the L2 round
//     trip is hidden once per round instead of paid once per match.
// A round whose matches read bytes the round itself produces (distance < start + length, e.g. runs), or that puts out more than
// 64 bytes, takes the slow path: the matches one after the other, as the serial loop does.
// A lane off the chain decodes garbage, harmlessly: every table entry is either a valid one or 0 (unassigned), so indices stay
// inside the tables and no lane consumes more than 48 bits.  The next round's window is requested from the scalar cache as soon
// as the chain has been followed.
#ifdef SCFQ_LPROF      // measurement builds only (scripts/gpu_lanes_prof.sh): where a round's cycles go, summed over all waves
__device__ unsigned long long g_lprof[24];
#define SCFQ_LP_T(var_) const uint64_t var_ = __builtin_readcyclecounter()
#define SCFQ_LP_ADD(slot_, v_) lp[slot_] += (v_)
#else
#define SCFQ_LP_T(var_) do {} while (0)
#define SCFQ_LP_ADD(slot_, v_) do {} while (0)
#endif
typedef uint32_t dword8_t __attribute__((ext_vector_type(8)));
typedef const dword8_t __attribute__((address_space(4), aligned(4))) const_dword8_t;
typedef const uint32_t __attribute__((address_space(4), aligned(4))) const_dword1_t;

// inclusive prefix sum over the 64 lanes (row_shr 1, 2, 4, 8, then row_bcast 15 and 31)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t ddpp_add(uint32_t v) {
  return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ uint32_t dscan(uint32_t v) {
  v = ddpp_add<0x111, 0xf>(v);
  v = ddpp_add<0x112, 0xf>(v);
  v = ddpp_add<0x114, 0xf>(v);
  v = ddpp_add<0x118, 0xf>(v);
  v = ddpp_add<0x142, 0xa>(v);
  v = ddpp_add<0x143, 0xc>(v);
  return v;
}

template <bool SYM16>
__device__ __attribute__((noinline)) void symbol_loop_lanes(SymState* stp, const uint8_t* in_aligned, uint32_t in_off, uint32_t ip_end, void* out_base,
                                                           uint32_t limit, uint32_t lit_lds, uint32_t dist_lds,
                                                           uint32_t scratch_lds /* 64 dwords of the wave's own LDS (the table builder's work area is free here) */) {
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t in_u = ((uint64_t)uni((uint32_t)((uintptr_t)in_aligned >> 32)) << 32) | uni((uint32_t)(uintptr_t)in_aligned);
  const uint64_t out_u = ((uint64_t)uni((uint32_t)((uintptr_t)out_base >> 32)) << 32) | uni((uint32_t)(uintptr_t)out_base);
  const_byte_t* const in4 = (const_byte_t*)in_u;
  out_base = (void*)out_u;
  in_off = uni(in_off);
  ip_end = uni(ip_end);
  limit = uni(limit);
  uint32_t pos = uni(stp->pos), err = kOk;
  uint64_t bit = (uint64_t)uni(stp->ip) * 8u - uni(stp->bc);          // the exact position in the stream, in bits from in_aligned + in_off
  const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(out_base, 0, (int)(limit * (SYM16 ? 2u : 1u)), 0x00020000);
  typedef __attribute__((address_space(3))) uint32_t lds_u32;
  lds_u32* const litp = (lds_u32*)(uintptr_t)uni(lit_lds);
  lds_u32* const distp = (lds_u32*)(uintptr_t)uni(dist_lds);
  // (volatile: the lanes talk to each other through these words — to the compiler a lane that has just stored 0 and did not store
  // anything else itself would still read 0)
  volatile lds_u32* const scr = (volatile lds_u32*)(uintptr_t)uni(scratch_lds);
  auto store_sym = [&](uint32_t v, uint32_t off) {
    if (SYM16) __builtin_amdgcn_raw_buffer_store_b16((uint16_t)v, orsrc, off, 0, 0);
    else __builtin_amdgcn_raw_buffer_store_b8((uint8_t)v, orsrc, off, 0, 0);
  };
  auto load_sym = [&](uint32_t off) -> uint32_t {
    if (SYM16) return __builtin_amdgcn_raw_buffer_load_b16(orsrc, off, 0, 1 /*sc0*/);
    return __builtin_amdgcn_raw_buffer_load_b8(orsrc, off, 0, 1 /*sc0*/);
  };
  auto store_raw = [&](auto v, uint32_t off) {           // (no widening on the way: see pend_ld)
    if constexpr (SYM16) __builtin_amdgcn_raw_buffer_store_b16(v, orsrc, off, 0, 0);
    else __builtin_amdgcn_raw_buffer_store_b8(v, orsrc, off, 0, 0);
  };
  auto load_raw = [&](uint32_t off) {
    if constexpr (SYM16) return __builtin_amdgcn_raw_buffer_load_b16(orsrc, off, 0, 1 /*sc0*/);
    else return __builtin_amdgcn_raw_buffer_load_b8(orsrc, off, 0, 1 /*sc0*/);
  };
  constexpr uint32_t kSh = SYM16 ? 1u : 0u;
  constexpr uint32_t kOob = 0xFFFFFFFFu;               // an offset the buffer descriptor drops
  const uint32_t sft = lane & 31u;
  const bool upper = lane >= 32u;
  // bits 0 .. lane of a 64-bit mask
  const uint32_t le_lo = lane >= 31u ? 0xFFFFFFFFu : (2u << lane) - 1u;
  const uint32_t le_hi = lane < 32u ? 0u : (lane == 63u ? 0xFFFFFFFFu : (2u << (lane - 32u)) - 1u);
  // eight dwords from the (clamped) byte the position lies in: five of them hold the 128 bits of a round
#define SCFQ_WBYTE(bit_) (((uint32_t)((bit_) >> 3)) < ip_end ? (uint32_t)((bit_) >> 3) : ip_end)
  // (x4 + x1, not x8: the three dwords of an x8 nobody reads are registers the compiler re-uses at once, and re-using the target of a
  // load in flight means waiting for the load right where it was issued)
#define SCFQ_WLOAD(bit_)                                                           \
  do {                                                                             \
    const uint32_t a_ = (in_off + SCFQ_WBYTE(bit_)) & ~3u;                         \
    D = *(const_dword4_t*)(in4 + a_);                                              \
    D4 = *(const_dword1_t*)(in4 + a_ + 16u);                                       \
  } while (0)
  dword4_t D;
  uint32_t D4;
  SCFQ_WLOAD(bit);
  uint32_t done = 0;
  // the previous round's output: loaded symbol | 0x100 + literal, offset.  (The loaded symbol keeps the load's own type until it is
  // stored: widened to 32 bits it would be masked — and so waited for — at the bottom of the round that loaded it.)
  typedef typename std::conditional<SYM16, uint16_t, uint8_t>::type sym_t;
  sym_t pend_ld = 0;
  uint32_t pend_sel = 0, pend_off = kOob;
#ifdef SCFQ_LPROF
  uint64_t lp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  do {
    SCFQ_LP_T(t0);
    const uint32_t sh = (((in_off + SCFQ_WBYTE(bit)) & 3u) << 3) | ((uint32_t)bit & 7u);
    const uint32_t W0 = (uint32_t)((((uint64_t)D.y << 32) | D.x) >> sh), W1 = (uint32_t)((((uint64_t)D.z << 32) | D.y) >> sh),
                   W2 = (uint32_t)((((uint64_t)D.w << 32) | D.z) >> sh), W3 = (uint32_t)((((uint64_t)D4 << 32) | D.w) >> sh);
#ifdef SCFQ_LPROF
    asm volatile("" : : "s"(W0), "s"(W1), "s"(W2), "s"(W3));
#endif
    SCFQ_LP_T(t1);
    if (SCFQ_DABLATE & 8) {                // measurement only: ten more scalar instructions per round
      uint32_t t_ = pos;
#pragma unroll
      for (int q_ = 0; q_ < 10; ++q_) asm volatile("s_add_u32 %0, %0, 1" : "+s"(t_) : : "scc");
      asm volatile("" : : "s"(t_));
    }
    if (SCFQ_DABLATE & 16) {               // measurement only: ten more vector instructions per round
      uint32_t t_ = lane;
#pragma unroll
      for (int q_ = 0; q_ < 10; ++q_) asm volatile("v_add_u32 %0, %0, 1" : "+v"(t_));
      asm volatile("" : : "v"(t_));
    }
    // ---- 1. every lane: the symbol that would start at bit `lane` ---------------------------------------------------------------
    const uint32_t wa = upper ? W1 : W0, wb = upper ? W2 : W1, wc = upper ? W3 : W2;
    const uint32_t x0 = __builtin_amdgcn_alignbit(wb, wa, sft), x1 = __builtin_amdgcn_alignbit(wc, wb, sft);      // 64 bits from bit `lane` on
    const uint32_t e1 = litp[x0 & ((1u << kLitRoot) - 1u)];
    const bool sub = (e1 & kSub) != 0u;
    const uint32_t i2 = sub ? (e1 >> 16) + __builtin_amdgcn_ubfe(x0, kLitRoot, (e1 >> 4) & 15u) : 0u;
    const uint32_t e2 = litp[i2];
    const uint32_t ef = sub ? e2 : e1;
    const uint32_t n1 = sub ? (uint32_t)kLitRoot + (e2 & 15u) : (e1 & 15u);                                        // <= 15
    const uint32_t y0 = __builtin_amdgcn_alignbit(x1, x0, n1), y1 = x1 >> n1;
    const uint32_t lx = (ef >> 4) & 15u;                                                                           // <= 5 for a length code
    const uint32_t mlen = (ef >> 16) + __builtin_amdgcn_ubfe(y0, 0, lx);
    const uint32_t z0 = __builtin_amdgcn_alignbit(y1, y0, lx);
    const uint32_t d1 = distp[z0 & ((1u << kDistRoot) - 1u)];
    const bool dsub = (d1 & kSub) != 0u;
    const uint32_t j2 = dsub ? (d1 >> 16) + __builtin_amdgcn_ubfe(z0, kDistRoot, (d1 >> 4) & 15u) : 0u;
    const uint32_t d2 = distp[j2];
    const uint32_t df = dsub ? d2 : d1;
    const uint32_t dn = dsub ? (uint32_t)kDistRoot + (d2 & 15u) : (d1 & 15u);                                      // <= 15
    const uint32_t dx = (df >> 4) & 15u;                                                                           // <= 13
    const uint32_t moff = (df >> 16) + __builtin_amdgcn_ubfe(z0 >> dn, 0, dx);
    // 0 literal, 1 match, 2 end of block, 3 a code that is not assigned (the distance of an unassigned code is 0: caught in the walk)
    const uint32_t kind = (ef & kLit) ? 0u : ((ef & kVal) ? 1u : ((ef & kEob) ? 2u : 3u));
    const uint32_t tot = kind == 1u ? n1 + lx + dn + dx : n1;                                                      // <= 48
    const uint32_t A = (lane + tot) | (kind << 7) | ((kind == 0u ? ((ef >> 16) & 0xFFu) : mlen) << 9);
    const uint32_t B = (df & kVal) ? moff : 0u;
#ifdef SCFQ_LPROF
    asm volatile("" : : "v"(A), "v"(B));
#endif
    SCFQ_LP_T(t2);
    // ---- 2. the chain of real symbols: lane 0, then wherever each one ends.  The scalar unit ONLY follows it (it is what every
    // wave of the CU shares: with positions, checks and bookkeeping in this loop it was 55 instructions per symbol and two thirds of
    // the round's time); everything else about the symbols is vector work on the chain's lane mask, below.
    // (by hand: seven scalar instructions per symbol; the compiler's form of the same loop had twelve)
    uint32_t cur, a;
    uint64_t chain;
    asm volatile(
        "s_mov_b32 %[cur], 0\n\t"
        "s_mov_b64 %[chain], 0\n"
        "1:\n\t"
        "v_readlane_b32 %[a], %[A], %[cur]\n\t"
        "s_bitcmp1_b32 %[a], 8\n\t"               // an end-of-block code or one that is not assigned: where the chain ends
        "s_cbranch_scc1 2f\n\t"
        "s_bitset1_b64 %[chain], %[cur]\n\t"
        "s_and_b32 %[cur], %[a], 0x7f\n\t"
        "s_cmp_lt_u32 %[cur], 64\n\t"
        "s_cbranch_scc1 1b\n"
        "2:\n"
        : [cur] "=&s"(cur), [a] "=&s"(a), [chain] "=&s"(chain)
        : [A] "v"(A)
        : "scc");
    const bool stopped = (a & 0x100u) != 0u;           // (then `cur` is the lane of that code and a & 127 the bit behind it)
#ifdef SCFQ_LPROF
    asm volatile("" : : "s"(chain), "s"(cur));
#endif
    SCFQ_LP_T(t3);
    // ---- 3. where every symbol's output goes: a wave scan over the chain lanes ----------------------------------------------------
    const bool on = ((chain >> lane) & 1ull) != 0ull;
    const uint32_t len = on ? (kind ? A >> 9 : 1u) : 0u;
    const uint32_t incl = dscan(len);
    const uint32_t st = incl - len;                    // output offset inside the round
    // A round ends in front of a symbol it cannot take in one go: a match that reads output of this very round (not in memory when the
    // round's one load is issued), that reaches back further than there is output (or whose distance code is not assigned), or
    // output beyond the round's 64 lanes.  The symbol starts the next round; if it IS the first, it is done alone, the serial way.
    // (bitwise, not && / ||: short-circuit evaluation became three nested exec-mask regions)
    const bool cut_here = on & (((kind == 1u) & ((st + len > B) | (B - 1u >= pos + st))) | (st + len > 64u));
    const uint64_t cuts = __builtin_amdgcn_ballot_w64(cut_here);
    uint32_t R, adv, stop = 0, alone = 0, c = 64u;
    if (cuts == 0ull) {
      R = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
      adv = stopped ? (a & 127u) : cur;
      if (stopped) { stop = 1; if (((a >> 7) & 3u) == 3u) err = kErrData; }
    } else {
      c = (uint32_t)__builtin_ctzll(cuts);
      R = (uint32_t)__builtin_amdgcn_readlane((int)st, (int)c);
      adv = c;
      alone = c == 0u ? 1u : 0u;
    }
    // ---- 4. output ------------------------------------------------------------------------------------------------------------
    store_raw((pend_sel & 0x100u) ? (sym_t)(pend_sel & 0xFFu) : pend_ld, pend_off);         // the round before: its load has had this round's decode and walk to arrive
#ifdef SCFQ_LPROF
    asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
#endif
    SCFQ_LP_T(t4);
    {
      // every lane is one output byte of the round: its symbol is the one noted at the highest start at or below it
      scr[lane] = 0u;
      __builtin_amdgcn_wave_barrier();
      if (on && lane < c) scr[st] = 0x08000000u | ((A >> 7) & 0x7FFu) | (B << 11);          // kind, literal / length, distance
      __builtin_amdgcn_wave_barrier();
      const uint64_t starts = __builtin_amdgcn_ballot_w64(scr[lane] != 0u);
      const uint32_t m_lo = (uint32_t)starts & le_lo, m_hi = (uint32_t)(starts >> 32) & le_hi;
      const uint32_t s = m_hi ? 63u - (uint32_t)__builtin_clz(m_hi) : 31u - (uint32_t)__builtin_clz(m_lo | 1u);
      const uint32_t sv = scr[s];
      const bool act = lane < R;
      const bool is_match = (sv & 3u) == 1u;
      pend_ld = load_raw((act && is_match) ? (pos + lane - ((sv >> 11) & 0xFFFFu)) << kSh : kOob);     // a match byte is `distance` symbols back
      pend_sel = is_match ? 0u : 0x100u | ((sv >> 2) & 0xFFu);
      pend_off = act ? (pos + lane) << kSh : kOob;
    }
    pos += R;
    if (alone) {
      // the round's first symbol is a match the lanes cannot do at once (it overlaps its own output, or is longer than 64): the serial way
      const uint32_t a0 = (uint32_t)__builtin_amdgcn_readlane((int)A, 0);
      const uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)B, 0);
      const uint32_t mlen_s = a0 >> 9;
      if (off - 1u >= pos) {                           // further back than there is output, or the distance code is not assigned
        err = kErrData; stop = 1;
      } else {
        const uint32_t src0 = pos - off;
        for (uint32_t base = 0; base < mlen_s; base += 64) {
          const uint32_t k = base + lane;
          const uint32_t j = off >= mlen_s ? k : (off == 1 ? 0u : k % off);    // (overlapping its own output: period `off`)
          const uint32_t v = load_sym((src0 + j) << kSh);
          store_sym(v, k < mlen_s ? (pos + k) << kSh : kOob);
        }
        pos += mlen_s;
        adv = a0 & 127u;
      }
    }
    bit += adv;                                        // (an end-of-block code's bits included)
    SCFQ_WLOAD(bit);
    done = stop;
    done |= (limit - pos) >> 31;                       // pos > limit: the descriptor dropped the excess
    done |= (ip_end + 16u - (uint32_t)(bit >> 3)) >> 31;       // a malformed stream reading (clamped) bytes far past the end of the data
#ifdef SCFQ_LPROF
    { SCFQ_LP_T(t5);
      SCFQ_LP_ADD(0, t1 - t0); SCFQ_LP_ADD(1, t2 - t1); SCFQ_LP_ADD(2, t3 - t2); SCFQ_LP_ADD(3, t4 - t3); SCFQ_LP_ADD(4, t5 - t4);
      SCFQ_LP_ADD(5, 1); SCFQ_LP_ADD(6, (uint64_t)__builtin_popcountll(c < 64u ? chain & ((1ull << c) - 1ull) : chain) + alone); SCFQ_LP_ADD(7, alone); }
#endif
  } while (!done);
#ifdef SCFQ_LPROF
  if (lane == 0) for (int q = 0; q < 8; ++q) atomicAdd(&g_lprof[q], (unsigned long long)lp[q]);
#endif
  store_raw((pend_sel & 0x100u) ? (sym_t)(pend_sel & 0xFFu) : pend_ld, pend_off);
  // back to the byte-wise reader of the caller: whole bytes consumed, the rest of the last one in the bit buffer, and its prefetch
  const uint32_t ipn = (uint32_t)((bit + 7u) >> 3);
  const uint32_t bcn = (uint32_t)((uint64_t)ipn * 8u - bit);
  uint32_t bbn = 0;
  if (bcn) {
    const uint32_t a_ = in_off + (ipn - 1u < ip_end ? ipn - 1u : ip_end);
    const uint32_t dw = *(const_dword1_t*)(in4 + (a_ & ~3u));
    bbn = ((dw >> ((a_ & 3u) << 3)) & 0xFFu) >> (8u - bcn);
  }
  const uint32_t a2 = in_off + (ipn < ip_end ? ipn : ip_end);
  const dword4_t pf = *(const_dword4_t*)(in4 + (a2 & ~3u));
#undef SCFQ_WLOAD
#undef SCFQ_WBYTE
  stp->bb = bbn; stp->bc = bcn; stp->ip = ipn; stp->pos = pos; stp->err = err;
  stp->pf[0] = pf.x; stp->pf[1] = pf.y; stp->pf[2] = pf.z; stp->pf[3] = pf.w; stp->pf_sh = (a2 & 3u) << 3;
}

// ---- the symbol loop, boundary-first form -----------------------------------------------------------------------------------
// symbol_loop_lanes decodes 64 bit positions completely (value, extra bits, distance) to find the 4-5 of them that hold symbols: 27
// vector instructions per symbol, and vector issue is what bounds the kernel.  Here the two halves are apart:
//  A. rounds over 128 bit positions, two per lane, that only find out HOW LONG the code at every position is (one gather per table, first
//     level only: ~25 vector instructions per 64 positions); the scalar unit follows the chain as before and the positions of the chain's
//     symbols are collected in LDS, one after the other, over as many rounds as it takes to have 64 of them;
//  B. then lane k decodes symbol k completely (the same arithmetic as step 1 of symbol_loop_lanes, now with every lane on a real
//     symbol), one scan gives every symbol's place in the output, and the output goes out in chunks of 64 symbols: every lane finds the
//     symbol it belongs to, and is a literal or one load `distance` back.
// A code that needs the second table level, a length code whose distance code does, and the end-of-block code end part A early ("hard":
// 1-2 % of the symbols): the position is collected as the group's last, part B decodes it like any other and says where the chain goes on.
// A match that reads output of its own group ends a SUB-GROUP in front of it: the stores of everything before are issued first (the
// memory pipeline keeps a wave's accesses in order, as in symbol_loop_lanes); one that overlaps its own output is copied alone, with
// its period.  The last store of a sub-group waits for its load only when the next sub-group begins.
// LDS scratch (256 dwords: the table builder's work area): positions [0, 192), a chunk's start flags [192, 256).
#ifndef SCFQ_DENSE_EMIT2
#define SCFQ_DENSE_EMIT2 1      // 1 (r4: the default): the output chunks of a sub-group go out two at a time; 0 keeps one at a time
#endif
template <bool SYM16>
__device__ __attribute__((noinline)) void symbol_loop_dense(SymState* stp, const uint8_t* in_aligned, uint32_t in_off, uint32_t ip_end, void* out_base,
                                                           uint32_t limit, uint32_t lit_lds, uint32_t dist_lds, uint32_t scratch_lds) {
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t in_u = ((uint64_t)uni((uint32_t)((uintptr_t)in_aligned >> 32)) << 32) | uni((uint32_t)(uintptr_t)in_aligned);
  const uint64_t out_u = ((uint64_t)uni((uint32_t)((uintptr_t)out_base >> 32)) << 32) | uni((uint32_t)(uintptr_t)out_base);
  const_byte_t* const in4 = (const_byte_t*)in_u;
  out_base = (void*)out_u;
  in_off = uni(in_off);
  ip_end = uni(ip_end);
  limit = uni(limit);
  uint32_t pos = uni(stp->pos), err = kOk;
  const uint64_t bit0 = (uint64_t)uni(stp->ip) * 8u - uni(stp->bc);   // the block's first symbol, in bits from in_aligned + in_off
  uint32_t rel = 0;                                                   // the chain's position = bit0 + rel
  const uint32_t b0 = (uint32_t)(bit0 >> 3), f0 = (uint32_t)bit0 & 7u;
  const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(out_base, 0, (int)(limit * (SYM16 ? 2u : 1u)), 0x00020000);
  typedef __attribute__((address_space(3))) uint32_t lds_u32;
  typedef const uint32_t __attribute__((address_space(1))) glob_u32;
  lds_u32* const litp = (lds_u32*)(uintptr_t)uni(lit_lds);
  lds_u32* const distp = (lds_u32*)(uintptr_t)uni(dist_lds);
  volatile lds_u32* const scr = (volatile lds_u32*)(uintptr_t)uni(scratch_lds);
  constexpr uint32_t kP = 0u, kC = 192u;          // positions collected (up to 63 + 128), start flags of a chunk
  auto store_sym = [&](uint32_t v, uint32_t off) {
    if (SYM16) __builtin_amdgcn_raw_buffer_store_b16((uint16_t)v, orsrc, off, 0, 0);
    else __builtin_amdgcn_raw_buffer_store_b8((uint8_t)v, orsrc, off, 0, 0);
  };
  auto load_sym = [&](uint32_t off) -> uint32_t {
    if (SYM16) return __builtin_amdgcn_raw_buffer_load_b16(orsrc, off, 0, 1 /*sc0*/);
    return __builtin_amdgcn_raw_buffer_load_b8(orsrc, off, 0, 1 /*sc0*/);
  };
  auto store_raw = [&](auto v, uint32_t off) {
    if constexpr (SYM16) __builtin_amdgcn_raw_buffer_store_b16(v, orsrc, off, 0, 0);
    else __builtin_amdgcn_raw_buffer_store_b8(v, orsrc, off, 0, 0);
  };
  auto load_raw = [&](uint32_t off) {
    if constexpr (SYM16) return __builtin_amdgcn_raw_buffer_load_b16(orsrc, off, 0, 1 /*sc0*/);
    else return __builtin_amdgcn_raw_buffer_load_b8(orsrc, off, 0, 1 /*sc0*/);
  };
  constexpr uint32_t kSh = SYM16 ? 1u : 0u;
  constexpr uint32_t kOob = 0xFFFFFFFFu;
  typedef typename std::conditional<SYM16, uint16_t, uint8_t>::type sym_t;
  sym_t pend_ld = 0;
  uint32_t pend_sel = 0, pend_off = kOob;
#if SCFQ_DENSE_EMIT2
  sym_t pend2_ld = 0;                                  // the second chunk of a pair (below)
  uint32_t pend2_sel = 0, pend2_off = kOob;
#define SCFQ_XFLUSH() do { store_raw((pend_sel & 0x100u) ? (sym_t)(pend_sel & 0xFFu) : pend_ld, pend_off); pend_off = kOob; \
                           store_raw((pend2_sel & 0x100u) ? (sym_t)(pend2_sel & 0xFFu) : pend2_ld, pend2_off); pend2_off = kOob; } while (0)
#else
#define SCFQ_XFLUSH() do { store_raw((pend_sel & 0x100u) ? (sym_t)(pend_sel & 0xFFu) : pend_ld, pend_off); pend_off = kOob; } while (0)
#endif
  scr[kP + lane] = 0u;                                 // (positions nobody has collected yet are read by the lanes beyond a group's end)
  scr[kP + 64u + lane] = 0u;
  scr[kP + 128u + lane] = 0u;
  uint32_t n_sym = 0;                                  // positions collected and not yet decoded
  uint32_t endk = 0;                                   // 1: the chain has ended (end of block: rel is behind its code); 2: the last position collected is a hard one
  uint32_t stop = 0;
  const uint32_t scr_base = uni(scratch_lds);
  const uint32_t in_last = (in_off + ip_end) & ~3u;     // the last dword a clamped read may start at (the callers' buffers have 20 bytes behind it)
  uint32_t sv = 0, sv_rel = 0u - 4096u;                // (nothing loaded yet)
#ifdef SCFQ_LPROF
  uint64_t lp[24] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  do {
    // ---- A. lengths only, until 64 symbols are known -----------------------------------------------------------------------------
    // The stream comes through a VECTOR register: lane i holds dword i of the 2048 bits from sv_rel's dword on, and every lane pulls the
    // dwords its positions lie in with ds_bpermute — no scalar loads, shifts or waits per round (the scalar unit is what the CU's
    // twenty waves share, and it has the chain to follow).  A round covers 128 bit positions, two per lane (lane and 64 + lane), written
    // stage by stage for both so that the waits of the dependent steps — bpermute, gather, gather — are paid once for the two.
    if (n_sym < 64u && endk == 0u) do {
      SCFQ_LP_T(t0);
      uint32_t T = rel - sv_rel;                       // the chain's position inside sv, in bits
      if (T >= 1800u) {
        const uint64_t ab = bit0 + rel + 8u * in_off;  // in bits from in_aligned
        const uint32_t dw = (uint32_t)(ab >> 5);
        T = (uint32_t)ab & 31u;
        sv_rel = rel - T;
        const uint32_t ad = 4u * (dw + lane);
        sv = *(glob_u32*)(in_u + (ad < in_last ? ad : in_last));
      }
      const uint32_t ta = T + lane;                                                  // (the second position is 64 bits = two dwords on)
      const uint32_t j4 = (ta >> 3) & 0x1FCu;
      const uint32_t w0 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)j4, (int)sv), w1 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(j4 + 4u), (int)sv),
                     w2 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(j4 + 8u), (int)sv), w3 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(j4 + 12u), (int)sv);
      const uint32_t xa = __builtin_amdgcn_alignbit(w1, w0, ta), xb = __builtin_amdgcn_alignbit(w3, w2, ta);     // 32 bits from each position on
      const uint32_t e1a = litp[xa & ((1u << kLitRoot) - 1u)], e1b = litp[xb & ((1u << kLitRoot) - 1u)];
      const uint32_t s1a = (e1a & 15u) + ((e1a >> 4) & 15u), s1b = (e1b & 15u) + ((e1b >> 4) & 15u);             // code + extra bits
      const uint32_t d1a = distp[__builtin_amdgcn_ubfe(xa, s1a, kDistRoot)], d1b = distp[__builtin_amdgcn_ubfe(xb, s1b, kDistRoot)];
      const uint32_t dda = ((d1a & 15u) + ((d1a >> 4) & 15u)) & (uint32_t)__builtin_amdgcn_sbfe(e1a, 10, 1);     // (kVal is bit 10: a length code)
      const uint32_t ddb = ((d1b & 15u) + ((d1b >> 4) & 15u)) & (uint32_t)__builtin_amdgcn_sbfe(e1b, 10, 1);
      // a literal, or a length code with a distance code of the first level: the chain steps over it (to position + bits, 127 + 36 at most).
      // Anything else ends the round's chain: 256
      const uint32_t Aa = ((((e1a << 2) | (e1a & d1a)) & kVal) != 0u) ? lane + s1a + dda : 256u;
      const uint32_t Ab = ((((e1b << 2) | (e1b & d1b)) & kVal) != 0u) ? 64u + lane + s1b + ddb : 256u;
#ifdef SCFQ_LPROF
      asm volatile("s_waitcnt lgkmcnt(0)" : : "v"(Aa), "v"(Ab) : "memory");
#endif
      SCFQ_LP_T(t1);
      uint32_t cur, nx;
      uint64_t ca, cb;                                 // the chain's lanes in the first and in the second half
      // Two steps per pass with the registers swapped, so that no s_mov is needed; the lane select of v_readlane must not be a register a
      // vector instruction has written within the last four instructions — nobody inserts wait states into inline assembly —: s_nop.
      // (Lane select and bit index take the low six bits: positions 64 .. 127 address the second half's registers as they are.)
      // Straight-line code of eight steps with no branch per symbol was slower: 346 against 300 cycles per 64 positions.
      asm volatile(
          "s_mov_b32 %[nx], 0\n\t"
          "s_mov_b64 %[ca], 0\n\t"
          "s_mov_b64 %[cb], 0\n"
          "1:\n\t"
          "v_readlane_b32 %[cur], %[Aa], %[nx]\n\t"
          "s_bitset1_b64 %[ca], %[nx]\n\t"
          "s_cmp_lt_u32 %[cur], 64\n\t"
          "s_cbranch_scc0 2f\n\t"
          "s_nop 0\n\t"
          "v_readlane_b32 %[nx], %[Aa], %[cur]\n\t"
          "s_bitset1_b64 %[ca], %[cur]\n\t"
          "s_cmp_lt_u32 %[nx], 64\n\t"
          "s_nop 0\n\t"
          "s_cbranch_scc1 1b\n\t"
          "s_mov_b32 %[cur], %[nx]\n"
          "2:\n\t"
          "s_cmp_lt_u32 %[cur], 128\n\t"
          "s_cbranch_scc0 5f\n\t"
          "s_mov_b32 %[nx], %[cur]\n"
          "3:\n\t"
          "v_readlane_b32 %[cur], %[Ab], %[nx]\n\t"
          "s_bitset1_b64 %[cb], %[nx]\n\t"
          "s_cmp_lt_u32 %[cur], 128\n\t"
          "s_cbranch_scc0 5f\n\t"
          "s_nop 0\n\t"
          "v_readlane_b32 %[nx], %[Ab], %[cur]\n\t"
          "s_bitset1_b64 %[cb], %[cur]\n\t"
          "s_cmp_lt_u32 %[nx], 128\n\t"
          "s_nop 0\n\t"
          "s_cbranch_scc1 3b\n\t"
          "s_mov_b32 %[cur], %[nx]\n"
          "5:\n"
          : [cur] "=&s"(cur), [nx] "=&s"(nx), [ca] "=&s"(ca), [cb] "=&s"(cb)
          : [Aa] "v"(Aa), [Ab] "v"(Ab)
          : "scc");
#ifdef SCFQ_LPROF
      asm volatile("" : : "s"(ca), "s"(cb), "s"(cur));
#endif
      SCFQ_LP_T(t2);
      uint32_t nxt = rel + cur;
      if (cur >= 256u) {
        // the chain has come to a position it cannot step over (the last one it noted): what is there
        uint32_t sl, es, ds;
        if (cb) {
          sl = 127u - (uint32_t)__builtin_clzll(cb);
          es = (uint32_t)__builtin_amdgcn_readlane((int)e1b, (int)(sl - 64u)); ds = (uint32_t)__builtin_amdgcn_readlane((int)d1b, (int)(sl - 64u));
        } else {
          sl = 63u - (uint32_t)__builtin_clzll(ca);
          es = (uint32_t)__builtin_amdgcn_readlane((int)e1a, (int)sl); ds = (uint32_t)__builtin_amdgcn_readlane((int)d1a, (int)sl);
        }
        nxt = rel;
        if ((es & kSub) || ((es & kVal) && !(ds & kVal))) endk = 2u;                 // hard: collected, part B says what it is
        else {
          if (cb) cb &= ~(1ull << (sl - 64u)); else ca &= ~(1ull << sl);
          endk = 1u;
          nxt = rel + sl + (es & 15u);                                               // behind an end-of-block code
          if (!(es & kEob)) err = kErrData;
        }
      }
      // the chain's lanes note their positions behind those already collected (only they execute this: five vector instructions per half)
      auto note = [&](uint64_t chain, uint32_t first, uint32_t position0) {
        uint64_t save;
        uint32_t t_, u_;
        asm volatile(
            "s_mov_b64 %[save], exec\n\t"
            "s_mov_b64 exec, %[chain]\n\t"
            "v_mbcnt_lo_u32_b32 %[t], %[clo], 0\n\t"
            "v_mbcnt_hi_u32_b32 %[t], %[chi], %[t]\n\t"
            "v_lshl_add_u32 %[t], %[t], 2, %[base]\n\t"
            "v_add_u32 %[u], %[rel], %[lane]\n\t"
            "ds_write_b32 %[t], %[u]\n\t"
            "s_mov_b64 exec, %[save]"
            : [save] "=&s"(save), [t] "=&v"(t_), [u] "=&v"(u_)
            : [chain] "s"(chain), [clo] "s"((uint32_t)chain), [chi] "s"((uint32_t)(chain >> 32)), [base] "s"(scr_base + 4u * (kP + first)), [rel] "s"(position0), [lane] "v"(lane)
            : "memory");
      };
      const uint32_t na = (uint32_t)__builtin_popcountll(ca);
      note(ca, n_sym, rel);
      if (cb) note(cb, n_sym + na, rel + 64u);
      n_sym += na + (uint32_t)__builtin_popcountll(cb);
      rel = nxt;
#ifdef SCFQ_LPROF
      { asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory"); SCFQ_LP_T(t3); SCFQ_LP_ADD(8, t1 - t0); SCFQ_LP_ADD(9, t2 - t1); SCFQ_LP_ADD(10, t3 - t2); SCFQ_LP_ADD(11, 1); }
#endif
    } while ((n_sym | (0u - endk)) < 64u);
    // (once per group — at most 64 rounds, 8 KiB of clamped reads:) a malformed stream reading far past the end of the data: over, the caller
    // sees the position; a block of 256 MiB: not here
    if ((ip_end + 16u - (b0 + ((f0 + rel) >> 3))) >> 31) stop = 1;
    if ((0x7FFFFFFFu - rel) >> 31) { stop = 1; err = kErrData; }
    // ---- B. lane k: symbol k ----------------------------------------------------------------------------------------------------
    const uint32_t m = n_sym < 64u ? n_sym : 64u;
    if (m != 0u && err == kOk) {
      __builtin_amdgcn_wave_barrier();
      SCFQ_LP_T(t4);
      const bool valid = lane < m;
      const uint32_t p = scr[kP + lane];
      const uint32_t q = f0 + p;
      const uint32_t by = b0 + (q >> 3);
      const uint32_t a_ = in_off + (by < ip_end ? by : ip_end);
      glob_u32* const w = (glob_u32*)(in_u + (a_ & ~3u));
      const uint32_t g0 = w[0], g1 = w[1], g2 = w[2];
      const uint32_t sh = ((a_ & 3u) << 3) | (q & 7u);
      const uint32_t x0 = __builtin_amdgcn_alignbit(g1, g0, sh), x1 = __builtin_amdgcn_alignbit(g2, g1, sh);      // 64 bits from the symbol's first
      const uint32_t e1 = litp[x0 & ((1u << kLitRoot) - 1u)];
      const bool sub = (e1 & kSub) != 0u;
      const uint32_t i2 = sub ? (e1 >> 16) + __builtin_amdgcn_ubfe(x0, kLitRoot, (e1 >> 4) & 15u) : 0u;
      const uint32_t e2 = litp[i2];
      const uint32_t ef = sub ? e2 : e1;
      const uint32_t n1 = sub ? (uint32_t)kLitRoot + (e2 & 15u) : (e1 & 15u);
      const uint32_t y0 = __builtin_amdgcn_alignbit(x1, x0, n1), y1 = x1 >> n1;
      const uint32_t lx = (ef >> 4) & 15u;
      const uint32_t mlen = (ef >> 16) + __builtin_amdgcn_ubfe(y0, 0, lx);
      const uint32_t z0 = __builtin_amdgcn_alignbit(y1, y0, lx);
      const uint32_t d1 = distp[z0 & ((1u << kDistRoot) - 1u)];
      const bool dsub = (d1 & kSub) != 0u;
      const uint32_t j2 = dsub ? (d1 >> 16) + __builtin_amdgcn_ubfe(z0, kDistRoot, (d1 >> 4) & 15u) : 0u;
      const uint32_t d2 = distp[j2];
      const uint32_t df = dsub ? d2 : d1;
      const uint32_t dn = dsub ? (uint32_t)kDistRoot + (d2 & 15u) : (d1 & 15u);
      const uint32_t dx = (df >> 4) & 15u;
      const uint32_t moff = (df >> 16) + __builtin_amdgcn_ubfe(z0 >> dn, 0, dx);
      const uint32_t kind = (ef & kLit) ? 0u : ((ef & kVal) ? 1u : ((ef & kEob) ? 2u : 3u));
      const uint32_t tot = kind == 1u ? n1 + lx + dn + dx : n1;
      const uint32_t B = (df & kVal) ? moff : 0u;
      const uint32_t len = valid ? (kind == 0u ? 1u : (kind == 1u ? mlen : 0u)) : 0u;
      const uint32_t incl = dscan(len);
      const uint32_t st = incl - len;                    // the symbol's place in the group's output
      const uint32_t R = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
      const bool tail = (endk == 2u) & (n_sym <= 64u);   // the group's last symbol is the hard one
      // a code that is not assigned; a distance that is not one, or reaches back further than there is output; an end-of-block code anywhere
      // but behind a hard position
      const bool bad = valid & ((kind == 3u) | ((kind == 1u) & (B - 1u >= pos + st)) | ((kind == 2u) & !(tail & (lane == m - 1u))));
      if (__builtin_amdgcn_ballot_w64(bad) != 0ull) {
        err = kErrData;
      } else {
        // what an output symbol needs to know about its symbol, pulled from the symbol's lane: a literal (bit 31) and its value, or a match's distance
        const uint32_t info = kind == 1u ? B : 0x80000000u | ((ef >> 16) & 0xFFu);
#ifdef SCFQ_LPROF
        asm volatile("s_waitcnt lgkmcnt(0)" : : "v"(st) : "memory");
#endif
        SCFQ_LP_T(t5);
        SCFQ_LP_ADD(12, t5 - t4); SCFQ_LP_ADD(14, 1); SCFQ_LP_ADD(17, m);
        // (Chunks that do not restart at sub-groups and end in front of the first LANE that reads the chunk's own output — fewer waits, 2.4
        // against 3.8 per group — took more instructions per chunk and were slower: 8200 against 6450 cycles per group.)
        uint32_t k0 = 0, s0 = 0;
        while (k0 < m) {
          // the first symbol from k0 on that reads what the sub-group starting at k0 puts out
          const uint64_t cutm = __builtin_amdgcn_ballot_w64(valid & (lane >= k0) & (kind == 1u) & (B < st + len - s0));
          const uint32_t kc = cutm ? (uint32_t)__builtin_ctzll(cutm) : m;
          if (kc == k0) {
            // it is the first itself: a match that overlaps its own output (distance < length), period `off`
            const uint32_t mlen_s = (uint32_t)__builtin_amdgcn_readlane((int)len, (int)k0);
            const uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)B, (int)k0);
            SCFQ_XFLUSH();
            SCFQ_LP_ADD(16, 1);
            const uint32_t dst0 = pos + s0, src0 = dst0 - off;
            for (uint32_t base = 0; base < mlen_s; base += 64) {
              const uint32_t k = base + lane;
              const uint32_t j = off == 1u ? 0u : k % off;
              const uint32_t v = load_sym((src0 + j) << kSh);
              store_sym(v, k < mlen_s ? (dst0 + k) << kSh : kOob);
            }
            k0 += 1u;
            s0 += mlen_s;
          } else {
            const uint32_t s1 = kc < m ? (uint32_t)__builtin_amdgcn_readlane((int)st, (int)kc) : R;
            uint32_t ka = k0;                            // symbols of the group that start in front of the chunk
            uint32_t first = 1;
            SCFQ_LP_ADD(16, 1);
#if SCFQ_DENSE_EMIT2
            // TWO chunks per pass while the sub-group has more than 64 symbols left: the start flags are bytes (128 of them in the 64
            // dwords), one clear, one scatter, and the dependent steps — flags, owner, info, load — of both chunks run side by side
            // (what the 128-position round of part A showed: a wave's time is its chain of dependent LDS steps).
            typedef __attribute__((address_space(3))) uint8_t lds_u8;
            volatile lds_u8* const c8 = (volatile lds_u8*)(uintptr_t)(scr_base + 4u * kC);
            const uint32_t stv = (valid & (lane - k0 < kc - k0) & (len != 0u)) ? st : 0xFFFFFF00u;
            uint32_t base2 = s0;
            while (base2 + 64u < s1) {
              SCFQ_LP_ADD(15, 2);
              scr[kC + (lane & 31u)] = 0u;
              __builtin_amdgcn_wave_barrier();
              if (stv - base2 < 128u) c8[stv - base2] = 1;
              __builtin_amdgcn_wave_barrier();
              const bool f0 = c8[lane] != 0, f1 = c8[64u + lane] != 0;
              const uint64_t starts0 = __builtin_amdgcn_ballot_w64(f0), starts1 = __builtin_amdgcn_ballot_w64(f1);
              const uint32_t n0 = (uint32_t)__builtin_popcountll(starts0);
              const uint32_t own0 = __builtin_amdgcn_mbcnt_hi((uint32_t)(starts0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)starts0, ka - 1u)) + (f0 ? 1u : 0u);
              const uint32_t own1 = __builtin_amdgcn_mbcnt_hi((uint32_t)(starts1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)starts1, ka + n0 - 1u)) + (f1 ? 1u : 0u);
              ka += n0 + (uint32_t)__builtin_popcountll(starts1);
              const uint32_t inf0 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(own0 << 2), (int)info), inf1 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(own1 << 2), (int)info);
              const bool act1 = base2 + 64u + lane < s1;                             // (the first chunk of a pair is whole)
              const bool m0 = (inf0 >> 31) == 0u, m1 = (inf1 >> 31) == 0u;
              if (first) SCFQ_XFLUSH();
              const sym_t ld0 = load_raw(m0 ? (pos + base2 + lane - inf0) << kSh : kOob);
              const sym_t ld1 = load_raw((act1 && m1) ? (pos + base2 + 64u + lane - inf1) << kSh : kOob);
              if (!first) SCFQ_XFLUSH();
              pend_ld = ld0; pend_sel = m0 ? 0u : 0x100u | (inf0 & 0xFFu); pend_off = (pos + base2 + lane) << kSh;
              pend2_ld = ld1; pend2_sel = m1 ? 0u : 0x100u | (inf1 & 0xFFu); pend2_off = act1 ? (pos + base2 + 64u + lane) << kSh : kOob;
              first = 0;
              base2 += 128u;
            }
            for (uint32_t base = base2; base < s1; base += 64u) {
#else
            for (uint32_t base = s0; base < s1; base += 64u) {
#endif
              // every lane is one output symbol: its owner is the symbol with the highest start at or below it
              SCFQ_LP_ADD(15, 1);
              scr[kC + lane] = 0u;
              __builtin_amdgcn_wave_barrier();
              if (valid & (lane - k0 < kc - k0) & (st - base < 64u) & (len != 0u)) scr[kC + (st - base)] = 1u;
              __builtin_amdgcn_wave_barrier();
              // symbols start in order: the owner is symbol ka - 1 plus the number of starts at or below the lane
              // (beyond the sub-group's end the sum may pass 63: those lanes put nothing out, and what they read from LDS is not used)
              const bool starts_here = scr[kC + lane] != 0u;
              const uint64_t starts = __builtin_amdgcn_ballot_w64(starts_here);
              const uint32_t own = __builtin_amdgcn_mbcnt_hi((uint32_t)(starts >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)starts, ka - 1u)) + (starts_here ? 1u : 0u);
              ka += (uint32_t)__builtin_popcountll(starts);
              const uint32_t inf = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(own << 2), (int)info);
              const bool act = base + lane < s1;
              const bool is_match = (inf >> 31) == 0u;
              if (first) SCFQ_XFLUSH();                  // what the sub-group before put out is what this one may read
              const sym_t ld = load_raw((act && is_match) ? (pos + base + lane - inf) << kSh : kOob);
              if (!first) SCFQ_XFLUSH();                 // (the chunk before: its load is older than the one just issued)
              pend_ld = ld;
              pend_sel = is_match ? 0u : 0x100u | (inf & 0xFFu);
              pend_off = act ? (pos + base + lane) << kSh : kOob;
              first = 0;
            }
            k0 = kc;
            s0 = s1;
          }
        }
        pos += R;
#ifdef SCFQ_LPROF
        { asm volatile("s_waitcnt vmcnt(0)" : : : "memory"); SCFQ_LP_T(t6); SCFQ_LP_ADD(13, t6 - t5); }
#endif
        if (n_sym <= 64u && endk == 2u) {
          // the hard symbol has been decoded: the chain goes on behind it, or the block ends with it
          const uint32_t kl = (uint32_t)__builtin_amdgcn_readlane((int)kind, (int)(m - 1u));
          rel = (uint32_t)__builtin_amdgcn_readlane((int)p, (int)(m - 1u)) + (uint32_t)__builtin_amdgcn_readlane((int)tot, (int)(m - 1u));
          if (kl == 2u) stop = 1;
          endk = 0u;
          if ((ip_end + 16u - (b0 + ((f0 + rel) >> 3))) >> 31) stop = 1;
          if ((0x7FFFFFFFu - rel) >> 31) err = kErrData;
        }
      }
    }
    if (n_sym > 64u) {
      const uint32_t v = scr[kP + 64u + lane], v2 = scr[kP + 128u + lane];
      __builtin_amdgcn_wave_barrier();
      scr[kP + lane] = v;
      scr[kP + 64u + lane] = v2;
      n_sym -= 64u;
    } else {
      n_sym = 0u;
      if (endk == 1u) stop = 1;
    }
    stop |= err != kOk ? 1u : 0u;
    stop |= (limit - pos) >> 31;                         // pos > limit: the descriptor dropped the excess
  } while (!stop);
  SCFQ_XFLUSH();
#ifdef SCFQ_LPROF
  if (lane == 0) for (int q = 8; q < 18; ++q) atomicAdd(&g_lprof[q], (unsigned long long)lp[q]);
#endif
#undef SCFQ_XFLUSH
  // back to the byte-wise reader of the caller (as symbol_loop_lanes)
  const uint64_t bit = bit0 + rel;
  const uint32_t ipn = (uint32_t)((bit + 7u) >> 3);
  const uint32_t bcn = (uint32_t)((uint64_t)ipn * 8u - bit);
  uint32_t bbn = 0;
  if (bcn) {
    const uint32_t a_ = in_off + (ipn - 1u < ip_end ? ipn - 1u : ip_end);
    const uint32_t dw = *(const_dword1_t*)(in4 + (a_ & ~3u));
    bbn = ((dw >> ((a_ & 3u) << 3)) & 0xFFu) >> (8u - bcn);
  }
  const uint32_t a2 = in_off + (ipn < ip_end ? ipn : ip_end);
  const dword4_t pf = *(const_dword4_t*)(in4 + (a2 & ~3u));
  stp->bb = bbn; stp->bc = bcn; stp->ip = ipn; stp->pos = pos; stp->err = err;
  stp->pf[0] = pf.x; stp->pf[1] = pf.y; stp->pf[2] = pf.z; stp->pf[3] = pf.w; stp->pf_sh = (a2 & 3u) << 3;
}

__global__ __launch_bounds__(64 * kWavesPerWg) void bgzf_inflate(const uint8_t* __restrict__ comp, const Block* __restrict__ blocks,
                                                                uint32_t n_blocks, uint8_t* out, uint32_t* status /* one word, OR of (1 << error) */,
                                                                uint32_t serial_loop /* 0: symbol_loop_lanes, 1: symbol_loop (A/B measurements), 2: symbol_loop_dense */) {
  extern __shared__ uint32_t lds[];
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t b = blockIdx.x * kWavesPerWg + wave;
  uint32_t* lit = lds + wave * (kWaveLdsBytes / 4);
  uint32_t* dist = lit + kLitEntries;
  uint32_t* cltab = dist;                                                // (aliased: see kWaveLdsBytes)
  uint8_t* lens = reinterpret_cast<uint8_t*>(dist + kDistEntries);       // 320 bytes
  uint16_t* sorted = reinterpret_cast<uint16_t*>(lens + 320);
  uint16_t* count = sorted + 320;
  uint16_t* offs = count + 16;
  __shared__ uint32_t build_ok[kWavesPerWg];
  __shared__ uint32_t s_len[32], s_dist[32];           // base << 16 | extra bits << 4 | kVal: table entries minus the code length
  // (r5: the member's CRC-32 is checked by a kernel of its own, bgzf_crc32_members below — its four 1 KiB tables were what kept a fifth
  // workgroup off every CU, and 20 waves per CU instead of 16 are worth 7 % on the symbol loop)
  if (threadIdx.x < 29) s_len[threadIdx.x] = ((uint32_t)kLenBase[threadIdx.x] << 16) | ((uint32_t)kLenExtra[threadIdx.x] << 4) | kVal;
  if (threadIdx.x < 30) s_dist[threadIdx.x] = ((uint32_t)kDistBase[threadIdx.x] << 16) | ((uint32_t)kDistExtra[threadIdx.x] << 4) | kVal;
  __syncthreads();
  if (b >= n_blocks) return;

  const Block blk = blocks[b];
  const uint8_t* const in = comp + blk.in_off;      // wave-uniform base; the read position is the 32-bit offset `ip`
  const uint32_t ip_end = blk.in_len;
  uint32_t ip = 0;
  uint8_t* const o = out + blk.out_off;
  const uint32_t isize = blk.isize;
  // the member's output as a buffer descriptor: an offset outside [0, isize) is dropped (store) or reads 0 (load)
  uint64_t bb = 0;
  uint32_t bc = 0;                                     // bits in bb: 56..63 after every refill
  uint32_t pos = 0, err = kOk;
#ifdef SCFQ_DSTATS
  uint32_t n_blk = 0, n_lit = 0, n_match = 0, n_mbytes = 0, n_overlap = 0, n_long = 0;
#endif
  bool last = false;

  // The bit reader.  bb holds the next bc bits of the stream, ip is the offset of the first byte not yet in bb.  A refill
  // is branch-free: OR in the 8 bytes at ip, advance ip by the whole bytes that fitted, and bc becomes bc | 56 (which is
  // bc + 8 * bytes taken).  The bytes come through the SCALAR cache (s_load: the compressed chunk is constant for the
  // kernel, `comp` is the base of a hipMalloc allocation and so dword-aligned): the aligned dwords that cover the 8
  // bytes, funnel-shifted by the byte offset.  That keeps the input stream out of vmcnt — with vector loads every refill
  // waited for the acknowledgement of the stores issued before it — and the dwords are requested one refill AHEAD
  // (pf: issued when ip is known, read at the next refill).  The load address is clamped to the member's end (the
  // 8-byte gzip trailer and at least 64 bytes of chunk padding follow the deflate data); ip itself is not, so that
  // `ip * 8 - bc`, the exact bit position, tells afterwards whether a malformed stream ran past its end.
  // (Macros, not lambdas: captured-by-reference state ended up in scratch memory, which made every derived value per-lane.)
  typedef uint32_t dword4 __attribute__((ext_vector_type(4)));
  typedef const dword4 __attribute__((address_space(4), aligned(4))) const_dword4;
  typedef const uint8_t __attribute__((address_space(4))) const_byte;
  const_byte* const in4 = (const_byte*)(uintptr_t)comp;
  const uint32_t in_off = blk.in_off;
  dword4 pf;                                           // one value, so that it stays in one register quad across the loop
  uint32_t pf_sh;
#define SCFQ_DPREFETCH()                                                                       \
  do {                                                                                         \
    const uint32_t a_ = in_off + (ip < ip_end ? ip : ip_end);                                  \
    pf_sh = (a_ & 3u) << 3;                                                                    \
    pf = *(const_dword4*)(in4 + (a_ & ~3u));                                                   \
  } while (0)
#define SCFQ_DREFILL()                                                                         \
  do {                                                                                         \
    const uint32_t up_ = 31u - pf_sh;           /* (x << 1) << up_ == x << (32 - pf_sh), also for pf_sh == 0 */ \
    const uint32_t lo_ = (pf.x >> pf_sh) | ((pf.y << 1) << up_), hi_ = (pf.y >> pf_sh) | ((pf.z << 1) << up_); \
    bb |= (((uint64_t)hi_ << 32) | lo_) << bc;                                                                          \
    ip += (63u - bc) >> 3;                                                                     \
    bc |= 56u;                                                                                 \
    SCFQ_DPREFETCH();                                                                          \
  } while (0)
#define SCFQ_DTAKE(n_) do { bb >>= (n_); bc -= (n_); } while (0)
  SCFQ_DPREFETCH();

  uint32_t guard = 0;                                  // every pass of every loop below consumes input bits or ends
  while (!last && err == kOk) {
    SCFQ_DREFILL();
    if (ip > ip_end + 16) { err = kErrData; break; }   // a malformed stream reading (clamped, harmless) bytes past its end
    last = bb & 1;
    const uint32_t type = (uint32_t)(bb >> 1) & 3u;
    SCFQ_DTAKE(3);
    if (type == 0) {
      // stored: to the byte boundary, LEN / NLEN, then a plain copy (all lanes)
      SCFQ_DTAKE(bc & 7);
      const uint32_t len = (uint32_t)bb & 0xFFFF, nlen = (uint32_t)(bb >> 16) & 0xFFFF;
      SCFQ_DTAKE(32);
      if ((len ^ nlen) != 0xFFFF) { err = kErrData; break; }
      ip -= bc >> 3;                                   // whole bytes go back to the byte stream
      bb = 0; bc = 0;
      if (ip > ip_end || ip_end - ip < len || pos + len > isize) { err = kErrData; break; }
      for (uint32_t k = lane; k < len; k += 64) o[pos + k] = in[ip + k];
      ip += len; pos += len;
      SCFQ_DPREFETCH();
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
        bool ok = build_table(lens, 288, kKindLitLen, lit, kLitRoot, kLitEntries, sorted, count, offs, s_len, s_dist);
        for (int k = 0; k < 32; ++k) lens[k] = 5;
        ok = ok && build_table(lens, 32, kKindDist, dist, kDistRoot, kDistEntries, sorted, count, offs, s_len, s_dist);
        build_ok[wave] = ok ? 1u : 0u;
      }
    } else {
      const uint32_t hlit = ((uint32_t)bb & 31) + 257, hdist = ((uint32_t)(bb >> 5) & 31) + 1, hclen = ((uint32_t)(bb >> 10) & 15) + 4;
      SCFQ_DTAKE(14);
      if (hlit > 286 || hdist > 30) { err = kErrData; break; }
      // the 19 code-length code lengths, then the run-length coded literal/length + distance code lengths: all lanes
      // walk the bits together, lane 0 writes
      if (lane == 0) for (int k = 0; k < 19; ++k) lens[k] = 0;
      for (uint32_t k = 0; k < hclen; ++k) {
        if ((k & 7) == 0) SCFQ_DREFILL();              // 8 x 3 bits per refill
        if (lane == 0) lens[kClOrder[k]] = (uint8_t)(bb & 7);
        SCFQ_DTAKE(3);
      }
      if (lane == 0) build_ok[wave] = build_table(lens, 19, kKindCodeLen, cltab, 7, 128, sorted, count, offs, s_len, s_dist) ? 1u : 0u;
      __builtin_amdgcn_wave_barrier();
      if (!__builtin_amdgcn_readfirstlane((int)build_ok[wave])) { err = kErrData; break; }
      uint32_t k = 0, prev = 0;
      const uint32_t total = hlit + hdist;
      while (k < total) {
        if (++guard > (1u << 20)) { err = kErrData; break; }
        SCFQ_DREFILL();                                // a code-length symbol and its repeat count: at most 7 + 7 bits
        const uint32_t e = uni(cltab[bb & 127]);
        if (!(e & kVal)) { err = kErrData; break; }
        SCFQ_DTAKE(e & 15);
        const uint32_t sym = e >> 16;
        uint32_t rep = 1, val = sym;
        if (sym == 16) { if (k == 0) { err = kErrData; break; } val = prev; rep = 3 + ((uint32_t)bb & 3); SCFQ_DTAKE(2); }
        else if (sym == 17) { val = 0; rep = 3 + ((uint32_t)bb & 7); SCFQ_DTAKE(3); }
        else if (sym == 18) { val = 0; rep = 11 + ((uint32_t)bb & 127); SCFQ_DTAKE(7); }
        if (k + rep > total) { err = kErrData; break; }
        if (lane == 0) for (uint32_t r = 0; r < rep; ++r) lens[k + r] = (uint8_t)val;
        k += rep;
        prev = val;
      }
      if (err) break;
      if (lane == 0) {
        bool ok = lens[256] != 0;
        // distance lengths follow the literal/length ones: build dist first from lens + hlit, then lit (build uses `sorted`)
        ok = ok && build_table(lens + hlit, (int)hdist, kKindDist, dist, kDistRoot, kDistEntries, sorted, count, offs, s_len, s_dist);
        ok = ok && build_table(lens, (int)hlit, kKindLitLen, lit, kLitRoot, kLitEntries, sorted, count, offs, s_len, s_dist);
        build_ok[wave] = ok ? 1u : 0u;
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (!__builtin_amdgcn_readfirstlane((int)build_ok[wave])) { err = kErrData; break; }
    // ---- symbols (symbol_loop above) -----------------------------------------------------------------------------------
#ifdef SCFQ_DSTATS
    ++n_blk;
#endif
    {
      SymState sst;
      sst.bb = bb; sst.bc = bc; sst.ip = ip; sst.pos = pos; sst.err = kOk;
      sst.pf[0] = pf.x; sst.pf[1] = pf.y; sst.pf[2] = pf.z; sst.pf[3] = pf.w; sst.pf_sh = pf_sh;
      if (serial_loop == 1u) symbol_loop<false>(&sst, comp + (blk.in_off & ~3u), blk.in_off & 3u, ip_end, o, isize, (uint32_t)(uintptr_t)lit, (uint32_t)(uintptr_t)dist);
      else if (serial_loop == 2u) symbol_loop_dense<false>(&sst, comp + (blk.in_off & ~3u), blk.in_off & 3u, ip_end, o, isize, (uint32_t)(uintptr_t)lit, (uint32_t)(uintptr_t)dist, (uint32_t)(uintptr_t)lens);
      else symbol_loop_lanes<false>(&sst, comp + (blk.in_off & ~3u), blk.in_off & 3u, ip_end, o, isize, (uint32_t)(uintptr_t)lit, (uint32_t)(uintptr_t)dist, (uint32_t)(uintptr_t)lens);
      // (every lane holds the same state in its private copy: read it back as wave-uniform values)
      bb = ((uint64_t)uni((uint32_t)(sst.bb >> 32)) << 32) | uni((uint32_t)sst.bb);
      bc = uni(sst.bc); ip = uni(sst.ip); pos = uni(sst.pos);
      const uint32_t e2 = uni(sst.err);
      pf.x = uni(sst.pf[0]); pf.y = uni(sst.pf[1]); pf.z = uni(sst.pf[2]); pf.w = uni(sst.pf[3]); pf_sh = uni(sst.pf_sh);
      if (e2) err = e2;
    }
    if (pos > isize && err == kOk) err = kErrLength;
  }
  // bits taken beyond the end of the deflate data (they were trailer bytes, or the clamped load's): the stream is malformed
  if (err == kOk && (uint64_t)ip * 8 - bc > (uint64_t)ip_end * 8) err = kErrData;
  if (err == kOk && pos != isize) err = kErrLength;
#undef SCFQ_DREFILL
#undef SCFQ_DPREFETCH
#undef SCFQ_DTAKE
#ifdef SCFQ_DSTATS
  if (lane == 0) { atomicAdd(status + 1, n_blk); atomicAdd(status + 2, n_lit); atomicAdd(status + 3, n_match); atomicAdd(status + 4, n_mbytes >> 4); atomicAdd(status + 5, n_overlap); atomicAdd(status + 6, n_long); }
#endif
  if (lane == 0 && err) atomicOr(status, 1u << err);      // one word for the whole launch: bit k = some block ended with error k
}

// ---- CRC-32 of every member of a launch (r5: a kernel of its own) ---------------------------------------------------------------------
// One workgroup per member.  The raw CRC (zero initial value, no final inversion: R(M) = M(x) x^32 mod P) of a virtual message "pad zero
// bytes, then the member's bytes" of 64 KiB — leading zeros do not change R —: thread t owns bytes [256 t, 256 t + 256) of it, four bytes per
// step through LDS tables, is shifted to the message's end with ONE product (x^(8 * 256 * (255 - t)), from a table) and the 256 threads are
// xor-ed; then crc32(M) = R(M) ^ 0xFFFFFFFF x^(8 |M|) ^ 0xFFFFFFFF against the member's trailer, x^(8 |M|) being the product of two table
// entries (|M| = 256 hi + lo).  The powers of x live in device memory, made once per context by bgzf_crc_consts_init: computed in the kernel —
// as gz_crc32_tiles does for its 1 MiB tiles — they were thirty 32-step products per thread for 256 bytes of data, and the kernel ran at
// 1.25 TB/s (0.6 ms per GiB of members beside 6.3 ms of inflate).
// A launch that has failed already (a member with corrupt deflate data or a wrong length) keeps its error word as it is.
constexpr uint32_t kMemberSpan = 1u << 16;
__device__ uint32_t g_crc_consts[3 * 256 + 8];      // [t]: x^(8 * 256 * (255 - t));  [256 + lo]: x^(8 lo);  [512 + hi]: x^(8 * 256 hi), hi = 0 .. 256
__global__ __launch_bounds__(256) void bgzf_crc_consts_init() {
  const uint32_t t = threadIdx.x;
  g_crc_consts[t] = x_pow_8n(256u * (255u - t));
  g_crc_consts[256u + t] = x_pow_8n(t);
  g_crc_consts[512u + t] = x_pow_8n(256u * t);
  if (t == 0) g_crc_consts[512u + 256u] = x_pow_8n(65536u);
}
__global__ __launch_bounds__(256) void bgzf_crc32_members(const uint8_t* __restrict__ out, const Block* __restrict__ blocks, uint32_t n_blocks, uint32_t* status) {
  __shared__ uint32_t red[256];
  __shared__ uint32_t tab[4][256];               // tab[0] the byte-wise table, tab[k][i] the CRC of byte i followed by k zero bytes
  const uint32_t t = threadIdx.x;
  {
    uint32_t e = t;
#pragma unroll
    for (int k = 0; k < 8; ++k) e = (e >> 1) ^ (0xEDB88320u & (0u - (e & 1u)));
    tab[0][t] = e;
    __syncthreads();
    for (int k = 1; k < 4; ++k) {
      const uint32_t q = tab[k - 1][t];
      tab[k][t] = (q >> 8) ^ tab[0][q & 255u];
      __syncthreads();
    }
  }
  if (blockIdx.x >= n_blocks) return;            // (workgroup-uniform)
  const Block blk = blocks[blockIdx.x];
  const uint32_t n = blk.isize;
  if (n > kMemberSpan) { if (t == 0) atomicOr(status, 1u << kErrLength); return; }      // (the host's planner admits members of at most 64 KiB)
  const uint8_t* data = out + blk.out_off;
  const uint32_t pad = kMemberSpan - n;
  const uint32_t v = t * 256u;                   // virtual offset of this thread's 256 bytes
  uint32_t c = 0;
  if (v + 256u > pad) {
    if (v >= pad) {
#pragma unroll 1
      for (uint32_t g = 0; g < 4u; ++g) {
        uint4 q[4];
        __builtin_memcpy(q, data + (v - pad) + 64u * g, 64);           // any alignment
        const uint32_t w[16] = {q[0].x, q[0].y, q[0].z, q[0].w, q[1].x, q[1].y, q[1].z, q[1].w, q[2].x, q[2].y, q[2].z, q[2].w, q[3].x, q[3].y, q[3].z, q[3].w};
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          c ^= w[j];
          c = tab[3][c & 255u] ^ tab[2][(c >> 8) & 255u] ^ tab[1][(c >> 16) & 255u] ^ tab[0][c >> 24];
        }
      }
    } else {
      for (uint32_t j = pad - v; j < 256u; ++j) c = crc_byte(c, (uint32_t)data[v + j - pad]);      // the thread the member's first byte falls to
    }
  }
  red[t] = gf2_mulmod(g_crc_consts[t], c);
  __syncthreads();
  for (uint32_t s2 = 128; s2 > 0; s2 >>= 1) {
    if (t < s2) red[t] ^= red[t + s2];
    __syncthreads();
  }
  if (t == 0) {
    const uint32_t xn = gf2_mulmod(g_crc_consts[512u + (n >> 8)], g_crc_consts[256u + (n & 255u)]);
    const uint32_t total = red[0] ^ gf2_mulmod(xn, 0xFFFFFFFFu) ^ 0xFFFFFFFFu;
    if (total != blk.crc && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) atomicOr(status, 1u << kErrCrc);
  }
}

}  // namespace scfq_dinflate
