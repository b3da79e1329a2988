// fq_scan_kernels.hpp — hand-written gfx950 (CDNA4, wave64) kernels for the `sc fq-count` hot path.
//
// Replaces the per-line loop of the reference (src/fq_count.nim:38-45: `for line in lines(stream)`,
// `i mod 4` classifier, count("G")+count("C"), count("N"), line.len) by a phase-agnostic byte-stream
// reduction: every byte is read from HBM exactly once and attributed to the class
// r = (number of '\n' before it in the scanned range) mod 4; which r is the sequence line is only
// decided by the ordered fold of the per-range partials (K2).
//
// K1 fq_scan_tiles : one wave = one contiguous "range" of 4 KiB tiles. Tiles are streamed
//      HBM -> LDS with non-temporal LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 B = 1 KiB coalesced
//      per instruction, 4 per tile, a 2-slot ring per wave, no VGPR staging, no barriers: a wave only
//      ever reads LDS bytes it loaded itself).  Each lane then owns 64 contiguous bytes of the tile
//      (4 x ds_read_b128), turns them into 64-bit match masks ('\n', G|C, N [, '@', '+']) with the
//      bit-plane classifier (v_perm byte transpose, masked block swaps into 8 bit planes, one 8-input AND per
//      symbol in v_bitop3), gets its line phase from a DPP wave prefix-sum of newline counts, and adds
//      per-segment popcounts into 8-bit packed per-class fields (4 classes in one VGPR; the class index is a
//      shift amount, never a register index).  No MFMA: this is an HBM-bound byte reduction.
// K2 fq_fold_fused : ordered (non-commutative) fold of the per-range partials, one launch.
// K3 / K4          : quality-byte histogram / '@','+' structure check, fused variants of K1.  K3's default form counts
//      quality bytes over small alphabets straight from the bit planes (hist_tile_planes), LDS atomics only otherwise.
// K5 fq_index_masks + fq_index_expand : line index (record-boundary detection) in one pass over the input.
//
// Byte semantics follow Nim 1.0.6 readLine as used by the reference: '\n' ends a line, a '\r'
// directly before that '\n' is not part of the line; G/C/N are case-sensitive (fq_count.nim:43-44).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "scfq_hdrhash.hpp"

#ifndef SCFQ_ABLATE
#define SCFQ_ABLATE 0   // diagnostic timing-only builds (make ablate): 1 = no classifier, 2 = no segment accounting, 3 = neither
#endif

namespace scfq {

constexpr int kTile = 4096;          // bytes per wave-iteration (64 lanes x 64 B)
constexpr int kWavesPerBlock = 4;
constexpr int kRing = 2;             // default LDS ring slots per wave (1 being consumed + 1 in flight)
constexpr int kPartialWords = 32;    // == SCFQ_PARTIAL_WORDS
constexpr uint32_t kMaxTilesPerRange = 960;  // 16-bit per-lane class fields: 64 B/tile * 960 < 65536

// word offsets inside a partial (scfq_partial layout, include/sc_fqcount.h)
enum : int { W_NL = 0, W_GC = 1, W_N = 5, W_LEN = 9, W_STARTS = 13, W_FAT = 17, W_FPLUS = 21, W_BYTES = 25, W_LAST = 26 };

constexpr uint32_t F_QUAL_HIST = 1u, F_STRUCT = 2u;

// K3: per-wave LDS histogram, replicated kHistRep x (copy = lane % kHistRep) so that the skewed byte
// distributions of FASTQ (one quality value = 90 % of a line) do not serialise 64 lanes on one address.
// Layout: u32 bin[(byte * 4 + class) * kHistRep + copy]  -> 16 KiB per wave.
constexpr int kHistRep = 4;
constexpr int kHistWords = 256 * 4 * kHistRep;
constexpr int kHistWaves = 2;   // waves per block in the histogram variant (LDS: 2 x (ring + 16 KiB))

// K3, speculative fast form (HIST == 2): every range comes with a GUESS of which relative class is the quality line
// (fq_scan_tiles<.., GUESS> reads the '@' / '+' line starts of the first tiles of the range; the guess is VERIFIED
// against the exact phases of the ordered fold afterwards and wrong / missing guesses are redone by the exact
// HIST == 1 kernel, so results never depend on it).  With the class known only quality bytes are histogrammed, into ONE
// workgroup-shared histogram of kQBins bins x kQRep lane-keyed copies (16 KiB for 8 waves instead of 16 KiB per wave):
// u32 bin[byte * kQRep + copy].
// (r4) 128 bins x 32 copies, copy = lane & 31: bank = (byte & 1) << 5 | lane & 31, so the 32 lanes the LDS serves together never
// meet in a bank, whatever the bytes are — with 256 x 16 the lanes l and l + 16 shared a copy and collided whenever their bytes
// agreed modulo 4, i.e. in nearly every atomic of a long-read quality tile.  Quality bytes are ASCII; a workgroup that meets a
// quality byte >= 128 (or cannot tell: see the callers of q_note_high) marks its histogram as not to be used, its ranges are
// counted again by the exact kernel like those of a wrong guess (fq_hist_verify), and results never depend on the assumption.
// SCFQ_QWINDOW=0 keeps the 256-bin layout (SCFQ_QREP copies).
#ifndef SCFQ_QWINDOW
#define SCFQ_QWINDOW 0      // (r4: built, bit-exact, and measured at +1 % on long reads, -0.5 % on Illumina: LDS atomics cost a CU 4 - 5 cycles per
                            // wave instruction with or without bank conflicts — profiles/r04/hist_ab.txt.  Not the default.)
#endif
#ifndef SCFQ_QREP
#define SCFQ_QREP 16
#endif
constexpr bool kQWindow = SCFQ_QWINDOW != 0;
constexpr int kQRep = kQWindow ? 32 : SCFQ_QREP;     // 4 | 8 | 16 | 32
constexpr int kQBins = kQWindow ? 128 : 256;
constexpr int kQShift = 2 + (kQRep == 32 ? 5 : kQRep == 16 ? 4 : kQRep == 8 ? 3 : 2);   // byte offset of a bin copy: byte << kQShift | copy << 2
constexpr int kQWords = kQBins * kQRep;
constexpr uint32_t kQFlagWord = 10 * kQRep;      // (window layout) the workgroup's "do not use" word: bin '\n' copy 0 — a newline is never a quality byte
constexpr uint32_t kQPoison = 0xFFFFFFFFu;       // ... and what hist_wg[wg][255] holds for such a workgroup
#ifndef SCFQ_QWAVES
#define SCFQ_QWAVES 8
#endif
constexpr int kQWaves = SCFQ_QWAVES;   // waves (= ranges) per workgroup of the speculative form: they share one histogram
#ifndef SCFQ_QOCC
#define SCFQ_QOCC 1                    // waves per SIMD the speculative form is compiled for (the register budget the compiler is given; 1 = none)
#endif
constexpr uint32_t kNoGuess = 255u;

// ------------------------------------------------------------------------------------------------
// cross-lane helpers (wave64, DPP)
// ------------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_add(uint32_t v) {
  // v + (v moved by DPP control CTRL); lanes without a source add 0
  return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}

// inclusive prefix sum over the 64 lanes of a wave (Kogge-Stone in rows of 16, then row broadcasts)
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
  v = dpp_add<0x111, 0xf>(v);  // row_shr:1
  v = dpp_add<0x112, 0xf>(v);  // row_shr:2
  v = dpp_add<0x114, 0xf>(v);  // row_shr:4
  v = dpp_add<0x118, 0xf>(v);  // row_shr:8
  v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 -> rows 1,3
  v = dpp_add<0x143, 0xc>(v);  // row_bcast:31 -> rows 2,3
  return v;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
  v = wave_inclusive_scan(v);
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// value of lane-1 (lane 0 receives `lane0`), one DPP move
__device__ __forceinline__ uint32_t wave_shr1(uint32_t v, uint32_t lane0) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)lane0, (int)v, 0x138 /*wave_shr:1*/, 0xf, 0xf, false);
}

// ------------------------------------------------------------------------------------------------
// LDS-DMA tile loads.  Issued from inline asm on purpose: hipcc (ROCm 7.2) drains vmcnt(0) before
// any ds_read while a builtin LDS-DMA is pending, which would serialise the ring; as asm the
// loads are invisible to its bookkeeping and are retired by our own counted s_waitcnt vmcnt(N).
// The leading lgkmcnt(0) retires this wave's earlier ds_reads of the slot being overwritten.
// ------------------------------------------------------------------------------------------------
template <bool NT>
__device__ __forceinline__ void glds_tile(const uint8_t* lane_src, uint32_t lds_slot_addr) {
  uint32_t keep;
  if (NT) {   // non-temporal: the stream is read exactly once
    asm volatile(
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off nt\n\t"
        "global_load_lds_dwordx4 %1, off offset:1024 nt\n\t"
        "global_load_lds_dwordx4 %1, off offset:2048 nt\n\t"
        "global_load_lds_dwordx4 %1, off offset:3072 nt\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(lane_src), "s"(lds_slot_addr)
        : "memory");
  } else {
    asm volatile(
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "global_load_lds_dwordx4 %1, off offset:1024\n\t"
        "global_load_lds_dwordx4 %1, off offset:2048\n\t"
        "global_load_lds_dwordx4 %1, off offset:3072\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(lane_src), "s"(lds_slot_addr)
        : "memory");
  }
}

// edge tiles: every piece has its own (clamped) source address; still exactly 4 VMEM ops
__device__ __forceinline__ void glds_tile_edge(const uint8_t* s0, const uint8_t* s1, const uint8_t* s2,
                                               const uint8_t* s3, uint32_t lds_slot_addr) {
  uint32_t keep;
  asm volatile(
      "s_waitcnt lgkmcnt(0)\n\t"
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %5\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "global_load_lds_dwordx4 %2, off offset:1024\n\t"
      "global_load_lds_dwordx4 %3, off offset:2048\n\t"
      "global_load_lds_dwordx4 %4, off offset:3072\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(s0), "v"(s1), "v"(s2), "v"(s3), "s"(lds_slot_addr)
      : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ------------------------------------------------------------------------------------------------
// symbol masks: 64 bytes (16 dwords, memory order) -> three 64-bit masks, bit k <=> byte k
// ------------------------------------------------------------------------------------------------
// 4x4 byte transpose with v_perm_b32: in a,b,c,e (4 dwords) -> o[j] = {a.j, b.j, c.j, e.j}
__device__ __forceinline__ void transpose4x4(uint32_t a, uint32_t b, uint32_t c, uint32_t e, uint32_t* o) {
  uint32_t t0 = __builtin_amdgcn_perm(b, a, 0x05010400u);  // a0 b0 a1 b1
  uint32_t t1 = __builtin_amdgcn_perm(b, a, 0x07030602u);  // a2 b2 a3 b3
  uint32_t t2 = __builtin_amdgcn_perm(e, c, 0x05010400u);  // c0 e0 c1 e1
  uint32_t t3 = __builtin_amdgcn_perm(e, c, 0x07030602u);  // c2 e2 c3 e3
  o[0] = __builtin_amdgcn_perm(t2, t0, 0x05040100u);
  o[1] = __builtin_amdgcn_perm(t2, t0, 0x07060302u);
  o[2] = __builtin_amdgcn_perm(t3, t1, 0x05040100u);
  o[3] = __builtin_amdgcn_perm(t3, t1, 0x07060302u);
}

// ---- bit-plane classifier (the hot path) -----------------------------------------------------------------------
// After the byte transpose, dword j of a 32-byte group holds the bytes at positions j, 8+j, 16+j, 24+j.  Three
// rounds of masked block swaps between register pairs (an 8x8 bit-matrix transpose done on all four byte lanes at
// once) turn the 8 dwords into 8 BIT PLANES: plane i bit p = bit i of byte p, p = 0..31 in memory order.  A byte
// compare against a constant c is then one 8-input AND of planes or their complements = 3 + 1 v_bitop3_b32 per 32
// bytes and symbol, exact for every byte value (no ASCII assumption, no carry tricks, no fallback path).
// Cost per 64-byte lane: 32 v_perm + 96 cheap ops for the planes, then 8 v_bitop3 per symbol — against
// 48 (xor/add, shift, merge) ops per symbol for the carry-SWAR form it replaced.
struct PlaneConsts {
  uint32_t m1, m2, m4;   // swap masks, pinned in VGPRs (an SGPR source halves the issue rate; VOP3 takes no literal on gfx9).  The
                         // shifted form of a mask is its complement, and a complemented select is the same v_bitop3 with its
                         // operands swapped: three registers, not six
  __device__ __forceinline__ void init() {
    m1 = 0x55555555u; m2 = 0x33333333u; m4 = 0x0F0F0F0Fu;
    asm volatile("" : "+v"(m1), "+v"(m2), "+v"(m4));
  }
};

// (a & mask) | (b & ~mask)
__device__ __forceinline__ uint32_t bsel(uint32_t mask, uint32_t a, uint32_t b) { return __builtin_amdgcn_bitop3_b32(mask, a, b, 0xCA); }

template <int S>
__device__ __forceinline__ void plane_swap(uint32_t& lo, uint32_t& hi, uint32_t m) {
  // exchange the bits of `lo` selected by (m << S) = ~m with the bits of `hi` selected by m
  const uint32_t nlo = bsel(m, lo, hi << S);
  const uint32_t nhi = bsel(m, lo >> S, hi);
  lo = nlo; hi = nhi;
}

// x[0..7] (byte-transposed dwords of one 32-byte group) -> bit planes in place
__device__ __forceinline__ void to_bit_planes(uint32_t* x, const PlaneConsts& pc) {
  plane_swap<1>(x[0], x[1], pc.m1); plane_swap<1>(x[2], x[3], pc.m1);
  plane_swap<1>(x[4], x[5], pc.m1); plane_swap<1>(x[6], x[7], pc.m1);
  plane_swap<2>(x[0], x[2], pc.m2); plane_swap<2>(x[1], x[3], pc.m2);
  plane_swap<2>(x[4], x[6], pc.m2); plane_swap<2>(x[5], x[7], pc.m2);
  plane_swap<4>(x[0], x[4], pc.m4); plane_swap<4>(x[1], x[5], pc.m4);
  plane_swap<4>(x[2], x[6], pc.m4); plane_swap<4>(x[3], x[7], pc.m4);
}

// minterm selectors for v_bitop3 (truth-table bit index = A<<2 | B<<1 | C)
constexpr int minterm3(int c, int i0, int i1, int i2) { return 1 << ((((c >> i0) & 1) << 2) | (((c >> i1) & 1) << 1) | ((c >> i2) & 1)); }
constexpr int minterm2(int c, int i0, int i1) { return 1 << ((((c >> i0) & 1) << 2) | (((c >> i1) & 1) << 1) | ((c >> i1) & 1)); }

// INVERTED match mask (bit set = byte != C) over the 32 positions of one group; DONTCARE2: ignore bit 2 ('C' | 'G')
template <int C, bool DONTCARE2 = false>
__device__ __forceinline__ uint32_t plane_ne(const uint32_t* w) {
  const uint32_t g1 = DONTCARE2 ? __builtin_amdgcn_bitop3_b32(w[0], w[1], w[1], minterm2(C, 0, 1))
                                : __builtin_amdgcn_bitop3_b32(w[0], w[1], w[2], minterm3(C, 0, 1, 2));
  const uint32_t g2 = __builtin_amdgcn_bitop3_b32(w[3], w[4], w[5], minterm3(C, 3, 4, 5));
  const uint32_t g3 = __builtin_amdgcn_bitop3_b32(w[6], w[7], w[7], minterm2(C, 6, 7));
  return __builtin_amdgcn_bitop3_b32(g1, g2, g3, 0x7F);   // ~(g1 & g2 & g3)
}

// x[8]: the bit planes of the group, left to the caller (the K3 plane-matching form counts quality bytes from them)
template <bool STRUCT>
__device__ __forceinline__ void masks32_planes_x(const uint32_t* d, const PlaneConsts& pc, uint32_t* x, uint32_t& wnl, uint32_t& wgc,
                                                 uint32_t& wnn, uint32_t& wat, uint32_t& wpl) {
  transpose4x4(d[0], d[2], d[4], d[6], &x[0]);
  transpose4x4(d[1], d[3], d[5], d[7], &x[4]);
  to_bit_planes(x, pc);
  wnl = plane_ne<0x0A>(x);
  wgc = plane_ne<0x43, true>(x);
  wnn = plane_ne<0x4E>(x);
  if (STRUCT) { wat = plane_ne<0x40>(x); wpl = plane_ne<0x2B>(x); }
}

template <bool STRUCT>
__device__ __forceinline__ void masks32_planes(const uint32_t* d, const PlaneConsts& pc, uint32_t& wnl, uint32_t& wgc,
                                               uint32_t& wnn, uint32_t& wat, uint32_t& wpl) {
  uint32_t x[8];
  transpose4x4(d[0], d[2], d[4], d[6], &x[0]);
  transpose4x4(d[1], d[3], d[5], d[7], &x[4]);
  to_bit_planes(x, pc);
  wnl = plane_ne<0x0A>(x);
  wgc = plane_ne<0x43, true>(x);
  wnn = plane_ne<0x4E>(x);
  if (STRUCT) { wat = plane_ne<0x40>(x); wpl = plane_ne<0x2B>(x); }
}

struct Masks {
  uint64_t nl, gc, nn, at, pl;
};

// 64 bytes -> match masks (bit k <=> byte k), via the bit-plane classifier; used by the generic (edge-tile) path
template <bool STRUCT>
__device__ __forceinline__ Masks masks64(const uint32_t* d, const PlaneConsts& pc) {
  uint32_t a[5] = {0, 0, 0, ~0u, ~0u}, b[5] = {0, 0, 0, ~0u, ~0u};
  masks32_planes<STRUCT>(d, pc, a[0], a[1], a[2], a[3], a[4]);
  masks32_planes<STRUCT>(d + 8, pc, b[0], b[1], b[2], b[3], b[4]);
  Masks m;
  m.nl = ~((uint64_t)a[0] | ((uint64_t)b[0] << 32));
  m.gc = ~((uint64_t)a[1] | ((uint64_t)b[1] << 32));
  m.nn = ~((uint64_t)a[2] | ((uint64_t)b[2] << 32));
  m.at = ~((uint64_t)a[3] | ((uint64_t)b[3] << 32));
  m.pl = ~((uint64_t)a[4] | ((uint64_t)b[4] << 32));
  return m;
}

__device__ __forceinline__ uint32_t popc64(uint64_t v) { return (uint32_t)__popcll(v); }

// per-lane accumulators: 16-bit fields, [0] holds classes 0 (bits 0..15) and 2 (bits 16..31),
// [1] holds classes 1 and 3.  Fed once per tile from 8-bit x 4-class tile fields.
struct Acc16 {
  uint32_t e, o;
  __device__ __forceinline__ void add_tile8(uint32_t t8) {
    e += t8 & 0x00FF00FFu;
    o += (t8 >> 8) & 0x00FF00FFu;
  }
};

struct WaveState {
  Acc16 gc, nn, len, crlf, fat, fplus;      // (line starts per class are not counted tile by tile: range_starts() derives them)
  uint32_t p_gc, p_nn, p_len, p_crlf;   // 8-bit x 4-class fields of up to 3 tiles, not yet widened
  uint32_t p_fat, p_fpl;                // ... K4 fields (STRUCT only)
  uint32_t pending;                     // wave-uniform: tiles accumulated in p_*
  uint32_t phase;      // wave-uniform: newlines seen so far in this range, mod 4
  uint32_t nl_total;   // wave-uniform: newlines seen so far in this range
  int32_t prev_last;   // wave-uniform: byte before the next tile (-1: none / start of input)
  uint32_t qcls;       // wave-uniform, HIST == 2: relative class guessed to be the quality line (4 = no guess: no histogram)
  uint32_t piv4;       // wave-uniform, HIST == 2: the range's pivot quality byte, replicated into all four bytes
  uint32_t piv_set;    // wave-uniform: piv4 is valid
  uint32_t piv_cnt;    // per lane: quality dwords equal to piv4 (counted here instead of four LDS atomics each)
  // HIST == 2, plane-matching form (hist_tile_planes): the byte values met so far on quality lines ("hot" values, at most
  // kHot) are counted from the bit planes into per-lane registers; anything else goes to the LDS histogram
  uint32_t hotp[2];    // wave-uniform: the hot byte values in order of discovery, four per word
  uint32_t p_hot[2];   // per lane: 8-bit fields, quality bytes equal to hot value k of up to 3 tiles (k = 4 * word + field): the value
                       // is a SHIFT AMOUNT here, never a register index — fields of 16 bits in four registers, picked by k, became an
                       // array in scratch memory (r4: a load and a store per value and tile, 7 % of the kernel)
  Acc16 hot_lo, hot_hi;// per lane: the same widened to 16-bit fields (values 0..3 / 4..7), like the class counters
  uint32_t n_hot;      // wave-uniform
  uint32_t qmode;      // wave-uniform: 0 = plane matching, 1 = the alphabet of this range is too large for it: hist_tile_q
  uint32_t qover;      // wave-uniform: tiles that held quality bytes outside a full hot set
  uint32_t qbad;       // wave-uniform (window layout): a quality byte >= 128 may have gone to the LDS histogram: the workgroup's histogram is void
};
constexpr int kHot = 8;
constexpr uint32_t kHotOverflowTiles = 2;   // after this many tiles with bytes outside a full hot set the range goes to hist_tile_q

// widen the pending 8-bit fields into the 16-bit per-lane accumulators (at most every 3rd tile: 3 x 64 < 256)
__device__ __forceinline__ void flush_pending(WaveState& st) {
  st.gc.add_tile8(st.p_gc);
  st.nn.add_tile8(st.p_nn);
  st.len.add_tile8(st.p_len);
  st.crlf.add_tile8(st.p_crlf);
  st.fat.add_tile8(st.p_fat);
  st.fplus.add_tile8(st.p_fpl);
  st.hot_lo.add_tile8(st.p_hot[0]);
  st.hot_hi.add_tile8(st.p_hot[1]);
  st.p_gc = st.p_nn = st.p_len = st.p_crlf = st.p_fat = st.p_fpl = 0;
  st.p_hot[0] = st.p_hot[1] = 0;
  st.pending = 0;
}

// ------------------------------------------------------------------------------------------------
// one 4 KiB tile, already resident in this wave's LDS slot
//   EDGE   : tile is not entirely inside [B, E): per-lane valid mask V applies
//   STRUCT : K4 line-start checks ('@' / '+')
//   HIST   : K3 byte histogram per class into this wave's LDS histogram
// ------------------------------------------------------------------------------------------------
template <bool EDGE, bool STRUCT, int HIST>
__device__ __forceinline__ void process_tile(const uint8_t* slot, int lane, uint64_t V, int64_t first_valid_pos,
                                             int32_t prev_byte_param, WaveState& st, uint32_t* hist_lds,
                                             const PlaneConsts& pc) {
  const uint4* p = reinterpret_cast<const uint4*>(slot + lane * 64);
  const uint4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
  uint32_t d[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w,
                    q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
  const Masks m = masks64<STRUCT>(d, pc);

  uint64_t NL = m.nl, GC = m.gc, NN = m.nn, VR = ~0ull;
  if (EDGE) { NL &= V; GC &= V; NN &= V; VR = V; }

  // line phase of this lane's first byte: wave prefix sum of newline counts
  const uint32_t cnt = popc64(NL);
  const uint32_t incl = wave_inclusive_scan(cnt);
  const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
  uint32_t sh = ((st.phase + incl - cnt) & 3u) * 8u;   // shift of this lane's current class field

  uint64_t LS = 0;
  if (STRUCT) {
    // line-start bytes: the byte after a '\n' (previous lane's / previous tile's last byte for bit 0)
    uint32_t last_is_nl = (uint32_t)(NL >> 63);
    uint32_t carry0 = (st.prev_last == '\n' || st.prev_last == -1) ? 1u : 0u;
    uint32_t carry = wave_shr1(last_is_nl, carry0);
    if (EDGE) {
      // the valid region may start inside this tile: its first valid byte starts a line iff the
      // caller's prev byte says so; bytes before it do not exist
      LS = (NL << 1);
      int64_t fv = first_valid_pos - (int64_t)lane * 64;   // lane-local index of first valid byte of the input
      if (fv >= 0 && fv < 64) {
        uint64_t bit = 1ull << fv;
        bool starts = (prev_byte_param == '\n' || prev_byte_param == -1);
        LS = (LS & ~bit) | (starts ? bit : 0);
      } else if (first_valid_pos < (int64_t)lane * 64) {
        LS |= carry;
      }
      LS &= V;
    } else {
      LS = (NL << 1) | carry;
    }
  }

  uint32_t t_gc = 0, t_nn = 0, t_len = 0, t_crlf = 0, t_fat = 0, t_fpl = 0;
  bool q_high = false;
  const int lane_base = lane * 64;
  uint64_t x = NL;
  for (;;) {
    const uint64_t xm1 = x - 1;
    const uint64_t below = ~x & xm1;     // bits strictly below the first remaining newline (all if none)
    const uint64_t upto = x ^ xm1;       // ... plus the newline itself
    const bool has_nl = (x != 0);
    const uint64_t seg = below & VR;
    t_len += popc64(seg) << sh;
    t_gc += popc64(GC & below) << sh;
    t_nn += popc64(NN & below) << sh;
    if (STRUCT) {
      const uint64_t ls = LS & upto;      // a line start may itself be the newline (empty line)
      t_fat += popc64(ls & m.at) << sh;
      t_fpl += popc64(ls & m.pl) << sh;
    }
    if (HIST == 2) {
      // K3 fast form (edge tiles only; interior tiles use hist_tile_q): the bytes of a quality segment, without its newline
      if ((sh >> 3) == st.qcls) {
        uint64_t s = seg;
        while (s) {
          const int k = __builtin_ctzll(s);
          s &= s - 1;
          const uint32_t byte = slot[lane_base + k];
          if (kQWindow && byte >= (uint32_t)kQBins) q_high = true;
          atomicAdd(&hist_lds[(byte & (uint32_t)(kQBins - 1)) * kQRep + (lane & (kQRep - 1))], 1u);
        }
      }
    }
    if (HIST == 1) {
      // K3 (edge tiles only; interior tiles use hist_tile_full): every valid byte of this segment INCLUDING its
      // newline goes to bin[class][byte]; newline and "\r\n" bytes are taken back once per range
      uint64_t s = upto & VR;
      const uint32_t cls = sh >> 3;
      while (s) {
        const int k = __builtin_ctzll(s);
        s &= s - 1;
        const uint32_t byte = slot[lane_base + k];
        atomicAdd(&hist_lds[(byte * 4 + cls) * kHistRep + (lane & (kHistRep - 1))], 1u);
      }
    }
    // the '\r' of a "\r\n" line end is not part of the line: look one byte behind the newline
    const int q = (int)popc64(below);                 // lane-local index of the newline (64 if none)
    int idx = lane_base + q - 1;
    // lanes without a newline read a dummy, bank-spread address (lane*4) instead of 64 B-strided ones
    int pb = slot[has_nl ? (idx < 0 ? 0 : idx) : lane * 4];
    if (idx < 0) pb = st.prev_last;
    if (EDGE) { if ((int64_t)(lane_base + q) == first_valid_pos) pb = prev_byte_param; }
    t_crlf += ((has_nl && pb == '\r') ? 1u : 0u) << sh;

    GC &= ~upto; NN &= ~upto; VR &= ~upto; x &= ~upto;
    if (STRUCT) LS &= ~upto;
    sh = (sh + 8u) & 31u;
    if (__builtin_amdgcn_ballot_w64(has_nl) == 0) break;   // every lane has consumed its last segment
  }

  st.p_gc += t_gc; st.p_nn += t_nn; st.p_len += t_len; st.p_crlf += t_crlf;
  if (HIST == 2 && kQWindow && __builtin_amdgcn_ballot_w64(q_high) != 0) st.qbad = 1u;
  if (STRUCT) { st.p_fat += t_fat; st.p_fpl += t_fpl; }
  if (++st.pending == 3) flush_pending(st);
  st.phase = (st.phase + total) & 3u;
  st.nl_total += total;
  st.prev_last = (int32_t)((uint32_t)__builtin_amdgcn_readlane((int)d[15], 63) >> 24);
}

// K3, interior tiles: all 64 bytes of the lane, fully unrolled. cls0 = class of the lane's first byte, NL = the lane's
// newline mask (a newline belongs to the line it ends, the class advances after it).  Written in cheap VOP2 ops
// only (shift / and / add with literals): the byte's bin offset is ((d >> s) & 0x3FC0) = byte << 6, the class term
// advances by the newline bit moved to bit 4.  Byte offset of a bin copy: byte << 6 | class << 4 | copy << 2.
__device__ __forceinline__ void hist_tile_full(const uint32_t* d, uint32_t* hist_lds, int lane, uint32_t cls0, uint64_t NL) {
  constexpr int kLog = (kHistRep == 4) ? 2 : (kHistRep == 2 ? 1 : 0);
  constexpr int kByteShift = 4 + kLog, kClsShift = 2 + kLog;      // byte offset = byte << kByteShift | class << kClsShift | copy << 2
  constexpr uint32_t kByteMask = 0xFFu << kByteShift, kClsMask = 3u << kClsShift, kClsOne = 1u << kClsShift;
  uint8_t* base = reinterpret_cast<uint8_t*>(hist_lds) + ((lane & (kHistRep - 1)) << 2);
  uint32_t clsterm = cls0 << kClsShift;
  const uint32_t nl_lo = (uint32_t)NL, nl_hi = (uint32_t)(NL >> 32);
#pragma unroll
  for (int k = 0; k < 64; ++k) {
    const uint32_t w = d[k >> 2];
    const int b = k & 3;
    const uint32_t off = (b == 0) ? ((w << kByteShift) & kByteMask) : ((w >> (8 * b - kByteShift)) & kByteMask);
    __hip_atomic_fetch_add(reinterpret_cast<uint32_t*>(base + off + clsterm), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const uint32_t m = (k < 32) ? nl_lo : nl_hi;
    const int j = k & 31;
    const uint32_t step = (j >= kClsShift) ? ((m >> (j - kClsShift)) & kClsOne) : ((m << (kClsShift - j)) & kClsOne);
    clsterm = (clsterm + step) & kClsMask;
  }
}

// K3 fast form, interior tiles: the class that is the quality line is known (guessed, verified later), so only the
// lane's quality bytes are counted.  Segment i of a lane (between its newlines i-1 and i) has class (cls0 + i) & 3; the
// first quality segment [a, b) is split into whole dwords [A, Bd) - four unpredicated atomics under one dword-level
// exec mask, byte -> bin offset in two cheap ops - and at most 3 head + 3 tail bytes re-read from LDS.  Further quality
// segments of the same lane (5+ newlines in 64 bytes) take a per-byte loop.
// hi7: bit 7 of the lane's 64 bytes (the classifier's plane 7).
__device__ __forceinline__ void hist_tile_q(uint32_t* hq, const uint8_t* slot, int lane, uint32_t cls0,
                                            uint64_t NL, uint32_t cnt, uint64_t hi7, WaveState& st) {
  const uint32_t qcls = st.qcls;
  const uint8_t* lane_bytes = slot + lane * 64;
  uint8_t* base = reinterpret_cast<uint8_t*>(hq) + ((lane & (kQRep - 1)) << 2);
  constexpr uint32_t kMask = (uint32_t)(kQBins - 1) << kQShift;   // byte offset of a bin copy: byte << kQShift | copy << 2
  static_assert(kQRep == 32 || kQRep == 16 || kQRep == 8 || kQRep == 4, "bin offset arithmetic");
#ifndef SCFQ_QABLATE
#define SCFQ_QABLATE 0   // timing-only diagnostic builds: 1 = addresses computed but no LDS atomics, 2 = no histogram work at all
#endif
  if (SCFQ_QABLATE == 2) return;
  auto bump = [&](uint32_t off) {
    off &= kMask;
    if (SCFQ_QABLATE == 1) { uint8_t* p = base + off; asm volatile("" :: "v"(p)); return; }
    __hip_atomic_fetch_add(reinterpret_cast<uint32_t*>(base + off), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  // bin address of the byte in bits 6..13 of t: the histogram is 16 KiB-aligned in LDS, so "base + offset" is an OR and
  // the whole address is one v_and_or_b32 on top of the shift
  const uint32_t base_bits = (uint32_t)(uintptr_t)base;
  auto bump_byte = [&](uint32_t t) {
    const uint32_t addr = (t & kMask) | base_bits;
    if (SCFQ_QABLATE == 1) { asm volatile("" :: "v"(addr)); return; }
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    __hip_atomic_fetch_add((lds_u32*)(uintptr_t)addr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  const uint32_t i0 = (qcls - cls0) & 3u;
  const uint64_t x0 = NL, x1 = x0 & (x0 - 1), x2 = x1 & (x1 - 1), x3 = x2 & (x2 - 1);
  const uint64_t xa = (i0 == 1) ? x0 : (i0 == 2) ? x1 : x2;                      // begins with newline i0-1 (i0 >= 1)
  const uint64_t xb = (i0 == 0) ? x0 : (i0 == 1) ? x1 : (i0 == 2) ? x2 : x3;    // begins with newline i0
  const uint32_t a = (i0 == 0) ? 0u : (xa ? (uint32_t)__builtin_ctzll(xa) + 1u : 65u);   // 65: the lane has no such segment
  const uint32_t b = xb ? (uint32_t)__builtin_ctzll(xb) : 64u;
  // long reads: most tiles lie entirely inside a header / sequence / separator line and hold no quality byte at all
  if (__builtin_amdgcn_ballot_w64(a < b || cnt >= i0 + 4u) == 0) return;
  // (window layout) a lane that holds quality bytes and ANY byte >= 128 voids the workgroup's histogram — conservative: the high
  // byte may lie on another line of the lane; FASTQ text is ASCII, and a file where it is not only loses the fast form here
  if (kQWindow && __builtin_amdgcn_ballot_w64(hi7 != 0 && (a < b || cnt >= i0 + 4u)) != 0) st.qbad = 1u;
  // The lane's 16 dwords come from LDS again (4 x ds_read_b128 of the slot the classifier read them from): kept in registers
  // from the top of the tile they were 16 VGPRs that every tile of the kernel paid for — with them gone the kernel fits
  // 5 waves per SIMD instead of 4, and occupancy is what this variant is short of.
  const uint4* lane_q = reinterpret_cast<const uint4*>(lane_bytes);
  if (__builtin_amdgcn_ballot_w64(cnt != 0u) == 0) {
    // long reads: a tile without any newline that lies inside a quality line (i0 == 0 for every lane, or the test above would have
    // left) is 4096 quality bytes — no segment bounds, no predicates, no head or tail: shift, v_and_or, ds_add per byte
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint4 v = lane_q[q];
      const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t w = w4[j];
        bump_byte(w << kQShift);
        bump_byte(w >> (8 - kQShift));
        bump_byte(w >> (16 - kQShift));
        bump_byte(w >> (24 - kQShift));
      }
    }
    return;
  }
  const uint32_t A = (a + 3u) >> 2, Bd = b >> 2;
  const uint32_t width = (Bd > A) ? Bd - A : 0u;
  // Pivot: quality strings are dominated by one value (one byte is ~90 % of an Illumina quality line), which would
  // serialise the lanes of every atomic on one bin.  A dword made of four pivot bytes is counted in a register instead;
  // the pivot is whatever byte starts the first whole quality dword this wave meets (any choice is exact).
  if (!st.piv_set) {
    const uint64_t have = __builtin_amdgcn_ballot_w64(width != 0);
    if (have) {
      const int L = __builtin_ctzll(have);
      const uint32_t a_l = (uint32_t)__builtin_amdgcn_readlane((int)A, L);
      const uint32_t byte = slot[L * 64 + 4 * a_l];
      st.piv4 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(byte * 0x01010101u));
      st.piv_set = 1u;
    }
  }
  const uint32_t piv4 = st.piv4;
  uint32_t pc = st.piv_cnt;
  // (four dwords at a time: the loads of the next four are not issued before these are used up)
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint4 v = lane_q[q];
    const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int j = 4 * q + jj;
      const uint32_t w = w4[jj];
      const bool in = (uint32_t)j - A < width;
      const bool is_piv = (w == piv4);
      pc += (in && is_piv) ? 1u : 0u;
      if (in && !is_piv) {
        bump_byte(w << kQShift);
        bump_byte(w >> (8 - kQShift));
        bump_byte(w >> (16 - kQShift));
        bump_byte(w >> (24 - kQShift));
      }
    }
    asm volatile("" ::: "memory");
  }
  st.piv_cnt = pc;
  const uint32_t he = (b < 4u * A) ? b : 4u * A;               // head bytes [a, he)
  const uint32_t ts = (Bd > A) ? 4u * Bd : 4u * A;             // tail bytes [ts, b)
#pragma unroll
  for (uint32_t i = 0; i < 3; ++i) {
    const uint32_t hp = a + i, tp = ts + i;
    if (hp < he) bump((uint32_t)lane_bytes[hp] << kQShift);
    if (tp < b) bump((uint32_t)lane_bytes[tp] << kQShift);
  }
  if (cnt >= i0 + 4u) {   // rare: another quality segment starts in this lane
    for (uint32_t k = b + 1; k < 64; ++k) {
      const uint64_t bit = 1ull << k;
      const uint32_t cls = (cls0 + popc64(NL & (bit - 1))) & 3u;
      if (cls == qcls && !(NL & bit)) bump((uint32_t)lane_bytes[k] << kQShift);
    }
  }
}

// ---- K3 fast form, plane matching ------------------------------------------------------------------------------
// The bit planes of the classifier are already there: the bytes of a 32-byte group that equal a value v are ONE 8-input
// AND of planes / complements (3 + 1 v_bitop3, the quality-segment mask M riding along as the ninth input), so a quality
// line over a small alphabet is counted with ~12 vector instructions per value and tile instead of one LDS atomic per
// byte: binned Illumina qualities (4 - 8 distinct values) never touch the LDS histogram at all.  The truth table of
// v_bitop3 is an immediate, the values are only known at run time: each 3-plane group is matched under a wave-uniform
// switch on the corresponding bits of v (scalar branches; the vector unit sees one instruction per case).
// Exact for any input: bytes outside the hot set are found (M minus the union of the matches), new values join the set
// while it has room, the rest goes to the LDS histogram byte by byte, and a range whose alphabet keeps overflowing the
// set switches to hist_tile_q (the dword loop) for its remaining tiles.
#include "hot_dispatch.inc"     // SCFQ_HOT_STANZAS: 256 stanzas of SCFQ_HOT_STANZA_BYTES, generated by scripts/gen_hot_dispatch.py

// Bytes equal to v (wave-uniform, 0..127) inside the segment masks.  pa0 / pa1 (positions 0..31 of the lane) and pb0 / pb1 (32..63) are
// the segment mask ANDed with "plane 7 clear, plane 6 clear / set" — made once per tile, shared by every value (r4; r2 - r3 matched
// planes 6 and 7 and the mask per value: 8 v_bitop3 per value where this form has 6).
// One computed jump into the stanza of v (six v_bitop3 with v's truth tables as immediates), one jump back: a compare
// chain over the bits of v (what a C++ switch becomes here) cost ~30 scalar instructions and ~8 branches per 3-plane
// group and saturated the CU's scalar unit.  s[90:91] hold the target address.
__device__ __forceinline__ void hot_match(const uint32_t* xa, const uint32_t* xb, uint32_t pa0, uint32_t pa1, uint32_t pb0, uint32_t pb1,
                                          uint32_t v, uint32_t& ma, uint32_t& mb) {
  uint32_t t0, t1, st;
  asm volatile(
      "s_mul_i32 %[st], %[v], %[stanza]\n\t"
      "s_getpc_b64 s[90:91]\n"
      ".Lhm_pc%=:\n\t"
      "s_add_u32 s90, s90, %[st]\n\t"
      "s_addc_u32 s91, s91, 0\n\t"
      "s_add_u32 s90, s90, .Lhm_tab%=-.Lhm_pc%=\n\t"
      "s_addc_u32 s91, s91, 0\n\t"
      "s_setpc_b64 s[90:91]\n"
      ".Lhm_tab%=:\n\t"
      SCFQ_HOT_STANZAS
      "\n.Lhm_end%=:\n\t"
      // the jump lands at .Lhm_tab + v * stanza: if a stanza's encoding ever changes size (another v_bitop3 encoding, a relaxed
      // branch, padding) this stops the build instead of jumping into the middle of an instruction
      ".if (.Lhm_end%= - .Lhm_tab%=) != 256 * %c[stanza]\n\t"
      ".error \"hot_match: the 256 stanzas of hot_dispatch.inc are not SCFQ_HOT_STANZA_BYTES each\"\n\t"
      ".endif"
      : [ma] "=&v"(ma), [mb] "=&v"(mb), [t0] "=&v"(t0), [t1] "=&v"(t1), [st] "=&s"(st)
      : [v] "s"(v), [stanza] "n"(SCFQ_HOT_STANZA_BYTES), [a0] "v"(xa[0]), [a1] "v"(xa[1]), [a2] "v"(xa[2]), [a3] "v"(xa[3]),
        [a4] "v"(xa[4]), [a5] "v"(xa[5]), [b0] "v"(xb[0]), [b1] "v"(xb[1]), [b2] "v"(xb[2]),
        [b3] "v"(xb[3]), [b4] "v"(xb[4]), [b5] "v"(xb[5]), [pa0] "v"(pa0), [pa1] "v"(pa1), [pb0] "v"(pb0), [pb1] "v"(pb1)
      : "s90", "s91", "scc");
}

// One interior tile whose lanes hold at most two newlines each (checked by the caller; every 100+ bp FASTQ shape).
// xa / xb: the bit planes of the lane's two 32-byte groups; cls0: class of the lane's first byte.
__device__ __forceinline__ void hist_tile_planes(const uint32_t* xa, const uint32_t* xb, uint32_t* hq, const uint8_t* slot, int lane,
                                                 uint32_t cls0, uint64_t NL, WaveState& st) {
  // the lane's quality segment: segment i0 = (qcls - cls0) & 3 of the lane (segments are separated by its newlines)
  const uint32_t i0 = (st.qcls - cls0) & 3u;
  const uint64_t xm1 = NL - 1, x1 = NL & xm1, x1m1 = x1 - 1;
  const uint64_t seg0 = ~NL & xm1;                       // below the first newline (everything when there is none)
  const uint64_t seg1 = (~x1 & x1m1) & ~(NL ^ xm1);      // above the first newline, below the second; none without a first
  const uint64_t seg2 = ~(x1 ^ x1m1);                    // above the second newline; none without a second
  // (r4, measured: the same selection with masks instead of ?: — 12 instructions where the compiler's nested exec-mask regions are ~25 —
  // ran no faster, 0.664–0.671 against 0.674; and the dword loop without its pivot ran 2.7 % SLOWER on long reads: profiles/r04/hist_ab.txt)
  const uint64_t M = (i0 == 0) ? seg0 : (i0 == 1) ? seg1 : (i0 == 2) ? seg2 : 0ull;
  const uint32_t Ma = (uint32_t)M, Mb = (uint32_t)(M >> 32);
  if (__builtin_amdgcn_ballot_w64((Ma | Mb) != 0) == 0) return;      // long reads: most tiles hold no quality byte at all
  // the segment mask with planes 7 and 6 folded in, for the values 0..63 and 64..127 (truth-table index: mask << 2 | plane 7 << 1 | plane 6)
  const uint32_t pa0 = __builtin_amdgcn_bitop3_b32(Ma, xa[7], xa[6], 0x10), pa1 = __builtin_amdgcn_bitop3_b32(Ma, xa[7], xa[6], 0x20);
  const uint32_t pb0 = __builtin_amdgcn_bitop3_b32(Mb, xb[7], xb[6], 0x10), pb1 = __builtin_amdgcn_bitop3_b32(Mb, xb[7], xb[6], 0x20);
  const uint32_t total = (uint32_t)__builtin_popcount(Ma) + (uint32_t)__builtin_popcount(Mb);
  // ---- the common pass: every hot value counted inside M, no remainder kept — whether the hot set covers the segment is told by the
  // COUNTS (hot values are distinct, a byte matches at most one: the sum of the counts is the number of covered bytes).  One dispatch
  // per value (wave-uniform loop; the values sit in a 64-bit scalar shift register), the counts go to 8-bit fields of two registers
  // (values 0..3 / 4..7) with one shift-add each.
  uint32_t n_hot = (uint32_t)__builtin_amdgcn_readfirstlane((int)st.n_hot);
  {
    uint64_t hv = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)st.hotp[1]) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)st.hotp[0]);
    uint32_t p0 = st.p_hot[0], p1 = st.p_hot[1], covered = 0, k = 0;
    for (; k < n_hot && k < 4u; ++k, hv >>= 8) {
      uint32_t ma, mb;
      hot_match(xa, xb, pa0, pa1, pb0, pb1, (uint32_t)hv & 0xFFu, ma, mb);
      const uint32_t c = (uint32_t)__builtin_popcount(ma) + (uint32_t)__builtin_popcount(mb);
      p0 = (c << (8u * k)) + p0;
      covered += c;
    }
    for (; k < n_hot; ++k, hv >>= 8) {
      uint32_t ma, mb;
      hot_match(xa, xb, pa0, pa1, pb0, pb1, (uint32_t)hv & 0xFFu, ma, mb);
      const uint32_t c = (uint32_t)__builtin_popcount(ma) + (uint32_t)__builtin_popcount(mb);
      p1 = (c << (8u * (k - 4u))) + p1;
      covered += c;
    }
    st.p_hot[0] = p0;
    st.p_hot[1] = p1;
    if (__builtin_amdgcn_ballot_w64(covered != total) == 0) return;      // (nearly every tile of a file with binned qualities)
  }
  // ---- a tile with bytes outside the hot set (the first tiles of a range; a file with many quality values): the remainder, by
  // matching the hot values once more — their counts are in already
  uint32_t ra = Ma, rb = Mb;
  {
    uint64_t hv = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)st.hotp[1]) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)st.hotp[0]);
    for (uint32_t k = 0; k < n_hot; ++k, hv >>= 8) {
      uint32_t ma, mb;
      hot_match(xa, xb, pa0, pa1, pb0, pb1, (uint32_t)hv & 0xFFu, ma, mb);
      ra &= ~ma;
      rb &= ~mb;
    }
  }
  uint64_t have = __builtin_amdgcn_ballot_w64((ra | rb) != 0);
  while (have != 0 && n_hot < (uint32_t)kHot) {
    // a byte outside the hot set while the set has room: its value joins and is counted at once.  (A value >= 128 cannot be matched by
    // this form — FASTQ text is ASCII — and goes the way of an overflowing set, below.)
    const int L = __builtin_ctzll(have);
    const uint32_t la = (uint32_t)__builtin_amdgcn_readlane((int)ra, L), lb = (uint32_t)__builtin_amdgcn_readlane((int)rb, L);
    const uint32_t kbit = la ? (uint32_t)__builtin_ctz(la) : 32u + (uint32_t)__builtin_ctz(lb);
    const uint32_t v = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)slot[L * 64 + kbit]);
    if (v >= 128u) break;
    if (n_hot < 4u) st.hotp[0] |= v << (n_hot * 8u); else st.hotp[1] |= v << ((n_hot - 4u) * 8u);
    uint32_t ma, mb;
    hot_match(xa, xb, pa0, pa1, pb0, pb1, v, ma, mb);
    const uint32_t c = (uint32_t)__builtin_popcount(ma) + (uint32_t)__builtin_popcount(mb);
    if (n_hot < 4u) st.p_hot[0] += c << (8u * n_hot); else st.p_hot[1] += c << (8u * (n_hot - 4u));
    ra &= ~ma;
    rb &= ~mb;
    n_hot += 1u;
    st.n_hot = n_hot;
    have = __builtin_amdgcn_ballot_w64((ra | rb) != 0);
  }
  if (have == 0) return;
  // the hot set is full (or the byte is not ASCII): the rest is counted in the workgroup's LDS histogram byte by byte; a range that
  // keeps coming here has a large alphabet (unbinned qualities) and is better served by the dword loop of hist_tile_q
  uint64_t r = (uint64_t)ra | ((uint64_t)rb << 32);
  const uint8_t* lane_bytes = slot + lane * 64;
  bool q_high = false;
  while (r) {
    const int kb = __builtin_ctzll(r);
    r &= r - 1;
    const uint32_t byte = lane_bytes[kb];
    if (kQWindow && byte >= (uint32_t)kQBins) q_high = true;
    atomicAdd(&hq[(byte & (uint32_t)(kQBins - 1)) * kQRep + (lane & (kQRep - 1))], 1u);
  }
  if (kQWindow && __builtin_amdgcn_ballot_w64(q_high) != 0) st.qbad = 1u;
  if (++st.qover > kHotOverflowTiles) st.qmode = 1u;
}

// ------------------------------------------------------------------------------------------------
// Interior tile of the default variant (no EDGE / STRUCT / HIST): the same arithmetic as
// process_tile with the segment loop restructured so that the common FASTQ shapes (0, 1 or 2
// newlines in a lane's 64 bytes) cost one straight-line "first segment" block, one "last segment"
// block and at most one pass of the middle loop.
// ------------------------------------------------------------------------------------------------
template <bool STRUCT, int HIST>
__device__ __forceinline__ void process_tile_fast(const uint8_t* slot, int lane, WaveState& st, uint32_t* hist_lds,
                                                  const PlaneConsts& pc) {
  const uint4* p = reinterpret_cast<const uint4*>(slot + lane * 64);
  const uint4 q0v = p[0], q1v = p[1], q2v = p[2], q3v = p[3];
  uint32_t d[16] = {q0v.x, q0v.y, q0v.z, q0v.w, q1v.x, q1v.y, q1v.z, q1v.w,
                    q2v.x, q2v.y, q2v.z, q2v.w, q3v.x, q3v.y, q3v.z, q3v.w};
  uint64_t WNL, WGC, WNN;   // inverted masks: bit set = byte is NOT '\n' / G|C / 'N'
  uint32_t xa[8] = {0, 0, 0, 0, 0, 0, 0, 0}, xb[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // bit planes of the two 32-byte groups (live past the classifier for HIST == 2 only)
  if (SCFQ_ABLATE == 1 || SCFQ_ABLATE == 3) {   // timing-only: masks are a cheap function of the data
    WNL = ~((uint64_t)(d[0] & d[5] & 0x01010101u) | ((uint64_t)(d[9] & d[13] & 0x00010100u) << 32));
    WGC = ((uint64_t)d[1] << 32) | d[2];
    WNN = ((uint64_t)d[3] << 32) | d[4] | d[6] | d[7] | d[8] | d[10] | d[11] | d[12] | d[14] | d[15];
  } else {
    uint32_t a0, a1, a2, a3 = 0, a4 = 0, b0, b1, b2, b3 = 0, b4 = 0;
    // (K4 takes no '@' / '+' masks from the classifier any more: it looks at the ONE byte behind each newline, below)
    masks32_planes_x<false>(d, pc, xa, a0, a1, a2, a3, a4);
    masks32_planes_x<false>(d + 8, pc, xb, b0, b1, b2, b3, b4);
    WNL = (uint64_t)a0 | ((uint64_t)b0 << 32);
    WGC = (uint64_t)a1 | ((uint64_t)b1 << 32);
    WNN = (uint64_t)a2 | ((uint64_t)b2 << 32);
  }
  const uint64_t NL = ~WNL;

  const uint32_t cnt = popc64(NL);
  const uint32_t incl = wave_inclusive_scan(cnt);
  const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
  const uint32_t sh0 = ((st.phase + incl - cnt) & 3u) * 8u;
  if (HIST == 1) hist_tile_full(d, hist_lds, lane, sh0 >> 3, NL);
  if (HIST == 2) {
    if (st.qcls < 4u) {   // wave-uniform branches
      // plane matching needs the quality segment of a lane to be one of its first three segments: at most two newlines
      // per lane (any read of 30+ bases); other tiles, and ranges with a large quality alphabet, take the dword loop
      if (st.qmode == 0u && __builtin_amdgcn_ballot_w64(cnt > 2u) == 0) hist_tile_planes(xa, xb, hist_lds, slot, lane, sh0 >> 3, NL, st);
      else hist_tile_q(hist_lds, slot, lane, sh0 >> 3, NL, cnt, (uint64_t)xa[7] | ((uint64_t)xb[7] << 32), st);
    }
  }

  // first segment: everything below the first newline (the whole lane when there is none)
  const uint64_t xm1 = NL - 1;
  const uint64_t below = WNL & xm1;
  uint32_t s_len = popc64(below);                 // running sums over the segments handled so far
  uint32_t s_gc = popc64(below & ~WGC);
  uint32_t s_nn = popc64(below & ~WNN);
  uint32_t t_len = st.p_len + (s_len << sh0);
  uint32_t t_gc = st.p_gc + (s_gc << sh0);
  uint32_t t_nn = st.p_nn + (s_nn << sh0);
  uint32_t t_crlf = st.p_crlf;
  // K4: a line is tested by its FIRST byte, and a line starts behind a newline: the byte behind each newline of the lane is read
  // from LDS next to the byte in front of it (the "\r\n" test) and compared — r3 built '@' and '+' masks of all 64 bytes with the
  // classifier (16 v_bitop3), a line-start mask and two 64-bit popcounts per segment for it: ~58 vector instructions per tile
  // against ~25 here.  Every line start is counted ONCE, by the tile that holds its first byte: the line behind a newline in the
  // tile's last byte belongs to the next tile (edge tiles, process_tile, count by the same rule), whose first byte it is.
  uint32_t t_fat = 0, t_fpl = 0;
  if (STRUCT) {
    t_fat = st.p_fat; t_fpl = st.p_fpl;
    if (st.prev_last == '\n' || st.prev_last == -1) {       // wave-uniform: this tile's first byte starts a line (class: lane 0's)
      const uint32_t b0 = slot[0];
      t_fat += ((lane == 0 && b0 == '@') ? 1u : 0u) << sh0;
      t_fpl += ((lane == 0 && b0 == '+') ? 1u : 0u) << sh0;
    }
  }

  if (total != 0 && SCFQ_ABLATE < 2) {   // wave-uniform: some lane of this tile holds a newline
    const int lane_base = lane * 64;
    const bool has1 = (NL != 0);
    // second newline of the lane (every FASTQ shape has at most two per 64 bytes: "...\n+\n...")
    const uint64_t x1 = NL & xm1, x1m1 = x1 - 1;
    const uint64_t below1 = ~x1 & x1m1;
    const bool has2 = (x1 != 0);
    // '\r' directly before a newline is not part of the line: both look-behind bytes are requested from LDS
    // before anything consumes them (newline-free lanes read bank-spread dummies)
    const int q1 = (int)popc64(below1);                         // index of the lane's second newline (64: none)
    const int idx0 = lane_base + (int)s_len - 1;
    const int idx1 = lane_base + q1 - 1;                        // >= 0: a lane's second newline is never at bit 0
    int pb0 = slot[has1 ? (idx0 < 0 ? 0 : idx0) : lane * 4];
    const int pb1 = slot[has2 ? idx1 : lane * 4 + 256];
    int na0 = 0, na1 = 0;
    bool v0 = false, v1 = false;
    if (STRUCT) {
      // the bytes BEHIND the two newlines (the next lane's first byte when the newline is this lane's last; nothing when it is the tile's)
      v0 = has1 && idx0 + 2 < kTile;
      v1 = has2 && idx1 + 2 < kTile;
      na0 = slot[v0 ? idx0 + 2 : lane * 4 + 512];
      na1 = slot[v1 ? idx1 + 2 : lane * 4 + 768];
    }
    if (idx0 < 0) pb0 = st.prev_last;
    t_crlf += ((has1 && pb0 == '\r') ? 1u : 0u) << sh0;
    uint32_t sh = (sh0 + 8u) & 31u;
    {   // the segment strictly between the first and the second newline (straight-line: no loop for FASTQ shapes)
      const uint64_t upto0 = NL ^ xm1;
      uint64_t seg = below1 & ~upto0;
      seg = has2 ? seg : 0;                        // a lane whose next segment is its last one is handled below
      const uint32_t m_len = popc64(seg), m_gc = popc64(seg & ~WGC), m_nn = popc64(seg & ~WNN);
      t_len += m_len << sh; t_gc += m_gc << sh; t_nn += m_nn << sh;
      s_len += m_len; s_gc += m_gc; s_nn += m_nn;
      if (STRUCT) {
        // the line behind the first newline has the class of this segment, the one behind the second the class after it
        const uint32_t sh2 = (sh + 8u) & 31u;
        t_fat += ((v0 && na0 == '@') ? 1u : 0u) << sh; t_fpl += ((v0 && na0 == '+') ? 1u : 0u) << sh;
        t_fat += ((v1 && na1 == '@') ? 1u : 0u) << sh2; t_fpl += ((v1 && na1 == '+') ? 1u : 0u) << sh2;
      }
      t_crlf += ((has2 && pb1 == '\r') ? 1u : 0u) << sh;
    }
    // further middle segments: three or more newlines in one lane (blank lines, very short records)
    uint64_t xc = x1, xcm1 = x1m1;
    for (;;) {
      const uint64_t x2 = xc & xcm1;               // drop the lowest remaining newline
      const bool has3 = (x2 != 0);
      if (__builtin_amdgcn_ballot_w64(has3) == 0) break;
      const uint64_t x2m1 = x2 - 1;
      const uint64_t below2 = ~x2 & x2m1;
      const uint64_t upto1 = xc ^ xcm1;
      uint64_t seg = below2 & ~upto1;
      seg = has3 ? seg : 0;
      sh = (sh + 8u) & 31u;
      const uint32_t m_len = popc64(seg), m_gc = popc64(seg & ~WGC), m_nn = popc64(seg & ~WNN);
      t_len += m_len << sh; t_gc += m_gc << sh; t_nn += m_nn << sh;
      s_len += m_len; s_gc += m_gc; s_nn += m_nn;
      const int idx = lane_base + (int)popc64(below2) - 1;
      const int pb = slot[has3 ? idx : lane * 4];
      if (STRUCT) {
        const bool v = has3 && idx + 2 < kTile;
        const int na = slot[v ? idx + 2 : lane * 4 + 512];
        const uint32_t shn = (sh + 8u) & 31u;
        t_fat += ((v && na == '@') ? 1u : 0u) << shn; t_fpl += ((v && na == '+') ? 1u : 0u) << shn;
      }
      t_crlf += ((has3 && pb == '\r') ? 1u : 0u) << sh;
      xc = x2; xcm1 = x2m1;
    }
    // last segment by complement: what lies above the highest newline = lane totals - segments counted so far
    // (newline-free lanes: totals == first segment, so this adds 0)
    const uint32_t shl = (sh0 + cnt * 8u) & 31u;
    t_len += (64u - cnt - s_len) << shl;
    t_gc += (64u - popc64(WGC) - s_gc) << shl;
    t_nn += (64u - popc64(WNN) - s_nn) << shl;
  }

  st.p_len = t_len; st.p_gc = t_gc; st.p_nn = t_nn; st.p_crlf = t_crlf;
  if (STRUCT) { st.p_fat = t_fat; st.p_fpl = t_fpl; }
  if (++st.pending == 3) flush_pending(st);
  st.phase = (st.phase + total) & 3u;
  st.nl_total += total;
  st.prev_last = (int32_t)((uint32_t)__builtin_amdgcn_readlane((int)d[15], 63) >> 24);
}

// ------------------------------------------------------------------------------------------------
// K1
// ------------------------------------------------------------------------------------------------
struct ScanArgs {
  const uint8_t* base;     // first byte of the range to scan (any alignment)
  uint64_t n;              // bytes
  int32_t prev_byte;       // byte before base[0]: 0..255, -1 = start of input, -2 = read base[-1]
  uint32_t tiles_per_range;
  uint64_t n_ranges;
  uint64_t* partials;      // [n_ranges][kPartialWords]
  uint32_t* hist_partials; // [n_ranges][4][256] u32, HIST == 1 only
  const uint8_t* todo;     // HIST == 1, optional: scan only the ranges with todo[range] != 0 (redo pass of the fast form)
  const uint8_t* guess;    // HIST == 2: per range, the relative class guessed to be the header line (0..3), kNoGuess = none
  uint32_t* hist_wg;       // HIST == 2: [n_workgroups][256] u32, quality-class histogram of the workgroup's ranges
  uint8_t* guess_out;      // GUESS: per range result
  uint32_t guess_cap_tiles;// GUESS: tiles a range may look at before giving up
};

// RING = LDS ring slots per wave (1 being consumed + RING-1 in flight); NT = non-temporal DMA loads
// HIST  = 0 none | 1 exact 4-class per-wave histogram | 2 speculative quality-class workgroup histogram (see kQRep)
// GUESS = true: no partials; look at the line starts of the first tiles of each range (K4 accounting) and report which
//         relative class is the header line: the only h with an '@' line start in class h and a '+' line start in
//         class h+2 (in a well-formed FASTQ only the header/separator pair can satisfy it: a sequence line never
//         starts with '@' or '+').  Ambiguous or not found within guess_cap_tiles: kNoGuess.
template <bool STRUCT, int HIST, int RING = kRing, bool NT = true, bool GUESS = false>
__global__ __launch_bounds__(HIST == 1 ? 64 * kHistWaves : HIST == 2 ? 64 * kQWaves : 64 * kWavesPerBlock, (HIST == 2 && !GUESS) ? SCFQ_QOCC : 1) void fq_scan_tiles(ScanArgs a) {
  constexpr int WAVES = HIST == 1 ? kHistWaves : HIST == 2 ? kQWaves : kWavesPerBlock;
  static_assert(RING >= 2 && RING <= 4, "ring depth");
  static_assert(!GUESS || (STRUCT && HIST == 0), "the guess pass is the K4 accounting without partials");
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int lane = threadIdx.x & 63;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // HIST == 2: the shared histogram comes first (its bin addressing ORs offsets into a 16 KiB-aligned base), rings after
  uint8_t* ring = smem + (HIST == 2 ? kQWords * 4 : 0) + wave * (RING * kTile);
  const uint64_t range = (uint64_t)blockIdx.x * WAVES + wave;
  bool active = range < a.n_ranges;
  if (HIST == 1 && active && a.todo) active = (a.todo[range] != 0);
  if (HIST != 2 && !active) return;     // HIST == 2: every wave reaches the workgroup barriers below
  uint32_t* hist_lds = nullptr;
  if (HIST == 1) {
    hist_lds = reinterpret_cast<uint32_t*>(smem + WAVES * RING * kTile) + wave * kHistWords;
    for (int k = lane; k < kHistWords; k += 64) hist_lds[k] = 0;
  }
  if (HIST == 2) {
    hist_lds = reinterpret_cast<uint32_t*>(smem);
    if ((uint32_t)(uintptr_t)hist_lds & (kQWords * 4 - 1)) __builtin_trap();   // a property of the build, not of the input
    for (int k = threadIdx.x; k < kQWords; k += 64 * WAVES) hist_lds[k] = 0;
    __syncthreads();
  }
  if (active) {

  // everything that steers the tile loop is wave-uniform: pin it in SGPRs so the per-tile bookkeeping runs on the
  // scalar unit instead of 64-bit VALU compares
  // Everything that steers the tile loop is wave-uniform and kept in 32-bit SGPR form (tile indices, not addresses):
  // gfx9 has no ordered 64-bit scalar compare, so 64-bit bookkeeping would run as VALU compares on every tile.
  const uint64_t B = (uint64_t)(uintptr_t)a.base, E = B + a.n;
  const uint64_t A0 = B & ~(uint64_t)(kTile - 1);
  const uint32_t n_tiles = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((E - A0 + kTile - 1) / kTile));
  const uint32_t t_begin = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(range * a.tiles_per_range));
  uint32_t t_end = t_begin + (GUESS ? a.guess_cap_tiles : a.tiles_per_range);   // a guess may look past its own range
  if (t_end > n_tiles) t_end = n_tiles;
  t_end = (uint32_t)__builtin_amdgcn_readfirstlane((int)t_end);
  // tiles [full_lo, full_hi) lie entirely inside [B, E); at most the first and the last tile of an input do not
  const uint32_t full_lo = (uint32_t)__builtin_amdgcn_readfirstlane((A0 < B) ? 1 : 0);
  const uint32_t full_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)((A0 + (uint64_t)n_tiles * kTile > E) ? n_tiles - 1 : n_tiles));

  // Both halo bytes are fetched and pinned into SGPRs BEFORE the first LDS-DMA is issued: a
  // compiler-visible load whose first use sat inside the tile loop made hipcc emit s_waitcnt vmcnt(0)
  // there, draining the whole DMA ring on every pass.
  int32_t prev_param = a.prev_byte;
  if (prev_param == -2) prev_param = a.base[-1];
  prev_param = __builtin_amdgcn_readfirstlane(prev_param);

  uint32_t qcls = 4u;
  if (HIST == 2) {
    const uint32_t g = a.guess[range];
    qcls = (g < 4u) ? ((g + 3u) & 3u) : 4u;     // header class h -> quality class h + 3
  }
  qcls = (uint32_t)__builtin_amdgcn_readfirstlane((int)qcls);   // pinned before the first DMA, like the halo bytes

  PlaneConsts pc;
  pc.init();
  WaveState st = {};
  st.qcls = qcls;
  uint32_t guessed = kNoGuess;
  // byte before this range's first tile: from memory when it belongs to the input, else the caller's halo
  st.prev_last = prev_param;
  if (A0 + (uint64_t)t_begin * kTile > B) st.prev_last = *reinterpret_cast<const uint8_t*>(A0 + (uint64_t)t_begin * kTile - 1);
  st.prev_last = __builtin_amdgcn_readfirstlane(st.prev_last);
  const int32_t range_prev = st.prev_last;      // the byte in front of the range's first byte (-1: the input starts here)

  const uint32_t ring_lds = (uint32_t)(uintptr_t)ring;   // LDS byte address of slot 0 (wave-uniform)

  auto issue = [&](uint32_t t, uint32_t slot) {
    const uint64_t ts = A0 + (uint64_t)t * kTile;
    const uint32_t dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)(ring_lds + slot * kTile));
    if (t >= full_lo && t < full_hi) {
      glds_tile<NT>(reinterpret_cast<const uint8_t*>(ts + (uint64_t)lane * 16), dst);
    } else {
      // pieces with no valid byte are redirected to a 16 B piece that is certainly readable
      const uint64_t safe = (B & ~15ull);
      uint64_t s[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint64_t ps = ts + (uint64_t)k * 1024 + (uint64_t)lane * 16;
        const bool ok = (ps + 16 > B) && (ps < E);
        s[k] = (ok ? ps : safe) - (uint64_t)k * 1024;   // the instruction re-adds offset:k*1024
      }
      glds_tile_edge(reinterpret_cast<const uint8_t*>(s[0]), reinterpret_cast<const uint8_t*>(s[1]),
                     reinterpret_cast<const uint8_t*>(s[2]), reinterpret_cast<const uint8_t*>(s[3]), dst);
    }
  };

  // prologue: RING-1 tiles in flight
#pragma unroll
  for (int k = 0; k < RING - 1; ++k)
    if (t_begin + k < t_end) issue(t_begin + k, k);

  uint32_t slot = 0;
  for (uint32_t t = t_begin; t < t_end; ++t) {
    const uint32_t s2 = (slot >= 1) ? slot - 1 : RING - 1;      // (slot + RING - 1) % RING: the slot consumed last
    if (t + (RING - 1) < t_end) issue(t + (RING - 1), s2);
    // retire tile t: everything issued after it may stay in flight (4 DMA instructions per tile)
    const uint32_t after = t_end - 1 - t;
    if (after >= (uint32_t)(RING - 1)) wait_vmcnt<4 * (RING - 1)>();
    else if (RING > 3 && after == 2) wait_vmcnt<8>();
    else if (RING > 2 && after == 1) wait_vmcnt<4>();
    else wait_vmcnt<0>();

    const uint8_t* sl = ring + slot * kTile;
    const uint64_t ts = A0 + (uint64_t)t * kTile;
    if (t >= full_lo && t < full_hi) {
      process_tile_fast<STRUCT, HIST>(sl, lane, st, hist_lds, pc);
    } else {
      // valid bytes of this lane: absolute [ts + 64*lane, +64) intersected with [B, E)
      const int64_t ls = (int64_t)(ts + (uint64_t)lane * 64);
      int64_t lo = (int64_t)B - ls, hi = (int64_t)E - ls;
      lo = lo < 0 ? 0 : (lo > 64 ? 64 : lo);
      hi = hi < 0 ? 0 : (hi > 64 ? 64 : hi);
      const uint64_t mhi = (hi >= 64) ? ~0ull : ((1ull << hi) - 1);
      const uint64_t mlo = (lo >= 64) ? ~0ull : ((1ull << lo) - 1);
      const uint64_t V = mhi & ~mlo;
      const int64_t first_valid = (B >= ts) ? (int64_t)(B - ts) : -1;   // tile-local position of input byte 0
      process_tile<true, STRUCT, HIST>(sl, lane, V, first_valid, prev_param, st, hist_lds, pc);
    }
    slot = (slot == RING - 1) ? 0 : slot + 1;
    if (GUESS) {
      // which relative classes have seen a line start with '@' / with '+' so far (any lane)
      flush_pending(st);
      auto any4 = [&](const Acc16& acc) {
        uint32_t m = 0;
        if (__builtin_amdgcn_ballot_w64((acc.e & 0xFFFFu) != 0)) m |= 1u;
        if (__builtin_amdgcn_ballot_w64((acc.o & 0xFFFFu) != 0)) m |= 2u;
        if (__builtin_amdgcn_ballot_w64((acc.e >> 16) != 0)) m |= 4u;
        if (__builtin_amdgcn_ballot_w64((acc.o >> 16) != 0)) m |= 8u;
        return m;
      };
      const uint32_t atm = any4(st.fat), plm = any4(st.fplus);
      const uint32_t cand = atm & ((plm >> 2) | (plm << 2)) & 0xFu;   // bit h: '@' in class h and '+' in class (h + 2) & 3
      if (cand) {
        guessed = (cand & (cand - 1)) ? kNoGuess : (uint32_t)__builtin_ctz(cand);
        wait_vmcnt<0>();     // tiles still in flight write this wave's LDS ring: retire them before the wave ends
        break;
      }
    }
  }
  if (GUESS) {
    if (lane == 0) a.guess_out[range] = (uint8_t)guessed;
    return;
  }

  flush_pending(st);
  // ---- range partial: reduce the per-lane 16-bit fields across the wave, lane 0 stores --------
  uint64_t* out = a.partials + range * kPartialWords;
  auto sum4 = [&](const Acc16& acc) {
    const uint32_t c0 = wave_sum(acc.e & 0xFFFFu), c2 = wave_sum(acc.e >> 16);
    const uint32_t c1 = wave_sum(acc.o & 0xFFFFu), c3 = wave_sum(acc.o >> 16);
    return make_uint4(c0, c1, c2, c3);
  };
  auto store4 = [&](int w, uint64_t v0, uint64_t v1, uint64_t v2, uint64_t v3) {
    if (lane == 0) { out[w + 0] = v0; out[w + 1] = v1; out[w + 2] = v2; out[w + 3] = v3; }
  };
  const uint4 gc4 = sum4(st.gc), nn4 = sum4(st.nn), len4 = sum4(st.len), cr4 = sum4(st.crlf);
  store4(W_GC, gc4.x, gc4.y, gc4.z, gc4.w);
  store4(W_N, nn4.x, nn4.y, nn4.z, nn4.w);
  // len excludes the '\r' of "\r\n": u64 modular (may wrap when that '\r' lies in the previous range)
  store4(W_LEN, (uint64_t)len4.x - cr4.x, (uint64_t)len4.y - cr4.y, (uint64_t)len4.z - cr4.z, (uint64_t)len4.w - cr4.w);
  if (STRUCT) {
    const uint4 a4 = sum4(st.fat), p4 = sum4(st.fplus);
    // Line starts per class need no per-tile work: a byte starts a line iff the byte before it is a '\n' (or it is the first byte of
    // the input), so the j-th newline of the range (class j mod 4) is followed by a line start of class (j + 1) mod 4 — unless it is
    // the range's last byte, whose successor is the next range's first byte (class 0 there) — and the range's own first byte starts
    // a line iff the byte before the range says so.
    const uint64_t last_at = ((A0 + (uint64_t)t_end * kTile < E) ? A0 + (uint64_t)t_end * kTile : E) - 1;
    const uint32_t last_is_nl = (*reinterpret_cast<const uint8_t*>(last_at) == (uint8_t)'\n') ? 1u : 0u;
    const uint32_t nfol = st.nl_total - last_is_nl;      // newlines of the range that are followed by a byte of the range
    const uint32_t first_is_start = (range_prev == '\n' || range_prev == -1) ? 1u : 0u;
    const uint4 s4 = make_uint4((nfol >> 2) + first_is_start, (nfol + 3u) >> 2, (nfol + 2u) >> 2, (nfol + 1u) >> 2);
    store4(W_STARTS, s4.x, s4.y, s4.z, s4.w);
    store4(W_FAT, a4.x, a4.y, a4.z, a4.w);
    store4(W_FPLUS, p4.x, p4.y, p4.z, p4.w);
  } else {
    store4(W_STARTS, 0, 0, 0, 0); store4(W_FAT, 0, 0, 0, 0); store4(W_FPLUS, 0, 0, 0, 0);
  }
  if (lane == 0) {
    out[W_NL] = st.nl_total;
    for (int k = W_BYTES; k < kPartialWords; ++k) out[k] = 0;
  }
  if (HIST == 2 && qcls < 4u) {
    // the '\r' of every "\r\n" that ends a quality line was counted as a line byte: take it back (u32 modular when
    // that '\r' lies in the previous range, like len)
    const uint32_t crq = (qcls == 0) ? cr4.x : (qcls == 1) ? cr4.y : (qcls == 2) ? cr4.z : cr4.w;
    if (lane == 0 && crq) atomicSub(&hist_lds[13 * kQRep], crq);
    uint32_t bad = st.qbad;                            // (window layout) a count that belongs to a bin >= 128 voids the workgroup's histogram
    const uint32_t pivots = wave_sum(st.piv_cnt);      // dwords of four pivot bytes that were counted in registers
    const uint32_t pv = st.piv4 & 0xFFu;
    if (kQWindow && pivots && pv >= (uint32_t)kQBins) bad = 1u;
    if (lane == 0 && pivots) atomicAdd(&hist_lds[(pv & (uint32_t)(kQBins - 1)) * kQRep], 4u * pivots);
    {   // the plane-matching form's register counts (flush_pending has widened them)
      const uint4 lo4 = sum4(st.hot_lo), hi4 = sum4(st.hot_hi);
      const uint32_t tot[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
#pragma unroll
      for (int k = 0; k < kHot; ++k) {
        const uint32_t v = ((k < 4 ? st.hotp[0] : st.hotp[1]) >> ((k & 3) * 8)) & 0xFFu;
        if (kQWindow && (uint32_t)k < st.n_hot && tot[k] && v >= (uint32_t)kQBins) bad = 1u;
        if (lane == 0 && (uint32_t)k < st.n_hot && tot[k]) atomicAdd(&hist_lds[(v & (uint32_t)(kQBins - 1)) * kQRep], tot[k]);
      }
    }
    if (kQWindow && bad && lane == 0) atomicOr(&hist_lds[kQFlagWord], 1u);
  }
  if (HIST == 1) {
    // per-range histogram partial [class][byte] (u32): sum the lane-keyed copies, then take back the bytes that
    // are not part of any line: newline j of the range ended a line of class j mod 4, and the '\r' of every
    // "\r\n" (per-class counts in cr4; u32 modular like len when that '\r' lies in the previous range)
    uint32_t* hp = a.hist_partials + range * 1024;
    for (int k = lane; k < 1024; k += 64) {
      const uint32_t c = (uint32_t)k >> 8, b = (uint32_t)k & 255u;
      const uint32_t* src = hist_lds + (b * 4 + c) * kHistRep;
      uint32_t v = 0;
#pragma unroll
      for (int r = 0; r < kHistRep; ++r) v += src[r];
      if (b == 10u) v -= (st.nl_total + 3u - c) >> 2;
      if (b == 13u) v -= (c == 0 ? cr4.x : c == 1 ? cr4.y : c == 2 ? cr4.z : cr4.w);
      hp[k] = v;
    }
  }
  }   // if (active)
  if (HIST == 2) {
    // workgroup histogram of the quality class: thread t sums the copies of byte value t
    __syncthreads();
    const uint32_t voided = kQWindow ? hist_lds[kQFlagWord] : 0u;
    for (uint32_t t = threadIdx.x; t < 256; t += 64 * WAVES) {
      uint32_t v = 0;
      if (t < (uint32_t)kQBins && !(kQWindow && t == 10u)) {
        const uint4* src = reinterpret_cast<const uint4*>(hist_lds + t * kQRep);
#pragma unroll
        for (int r = 0; r < kQRep / 4; ++r) { const uint4 q = src[r]; v += q.x + q.y + q.z + q.w; }
      }
      if (kQWindow && t == 255u && voided) v = kQPoison;      // fq_hist_verify: this workgroup's ranges go to the exact kernel
      a.hist_wg[(uint64_t)blockIdx.x * 256 + t] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// K2: ordered fold.  state = state (+) P_0 (+) P_1 (+) ... ; also records each range's starting
// phase (for the histogram fold) and finishes bytes / last_byte.
// ------------------------------------------------------------------------------------------------

constexpr int kFoldThreads = 256;
constexpr int kFold1 = 128;   // ranges folded per level-1 block

// Ordered fold of up to T partials, one per thread, without an ordered tree: only (newline count mod 4) of the
// predecessors matters, so   fold[c] = sum_t  P_t[(c - phase_t) & 3],  phase_t = (phase_in + sum_{u<t} nl_u) mod 4.
// One block-wide exclusive scan of nl (wave DPP scan + LDS carry); every thread then writes its 25 words, already
// rotated, as one row of an LDS matrix, and the columns are summed by (word, quarter-of-rows) threads.
// lds_acc[0..W_BYTES) must be zero on entry; result in lds_acc.  Returns this thread's starting phase.
// (A first version let all T threads ds_add_u64 into 25 shared words: 128-way same-address contention, 85 us.)
template <int T>
__device__ __forceinline__ uint32_t block_rotate_sum(const uint64_t* mine /*W_BYTES words, zeros if inactive*/,
                                                     uint32_t phase_in, uint64_t* lds_acc, uint32_t* lds_wave,
                                                     uint64_t (*rows)[W_BYTES + 1], int tid) {
  const int lane = tid & 63, w = tid >> 6;
  const uint32_t nl = (uint32_t)mine[W_NL];
  const uint32_t incl = wave_inclusive_scan(nl);
  if (lane == 63) lds_wave[w] = incl;
  __syncthreads();
  uint32_t before = 0;
#pragma unroll
  for (int k = 0; k < T / 64; ++k) before += (k < w) ? lds_wave[k] : 0u;
  const uint32_t phase = (phase_in + before + incl - nl) & 3u;
  rows[tid][W_NL] = mine[W_NL];
#pragma unroll
  for (int arr = W_GC; arr < W_BYTES; arr += 4) {
    // class r lands on (r + phase) & 3: out[c] = mine[arr + ((c - phase) & 3)].  Written as two conditional rotations — by one,
    // by two — of four registers: indexed by a run-time value, `mine` became an array in scratch memory (208 bytes per lane;
    // the runtime sets the device's scratch up inside the first launch of such a kernel, 35 - 50 ms of a process's first call)
    uint64_t m0 = mine[arr], m1 = mine[arr + 1], m2 = mine[arr + 2], m3 = mine[arr + 3];
    const bool by1 = (phase & 1u) != 0, by2 = (phase & 2u) != 0;
    const uint64_t a0 = by1 ? m3 : m0, a1 = by1 ? m0 : m1, a2 = by1 ? m1 : m2, a3 = by1 ? m2 : m3;
    rows[tid][arr + 0] = by2 ? a2 : a0;
    rows[tid][arr + 1] = by2 ? a3 : a1;
    rows[tid][arr + 2] = by2 ? a0 : a2;
    rows[tid][arr + 3] = by2 ? a1 : a3;
  }
  __syncthreads();
  const int word = tid & 31, part = tid >> 5;          // T / 32 row groups per word
  if (word < W_BYTES) {
    uint64_t sum = 0;
    const int r0 = part * 32;
#pragma unroll 8
    for (int r = 0; r < 32; ++r) sum += rows[r0 + r][word];
    atomicAdd(reinterpret_cast<unsigned long long*>(&lds_acc[word]), (unsigned long long)sum);   // T/32-way only
  }
  __syncthreads();
  return phase;
}

// K2 (single launch): every block folds kFold1 consecutive range partials (phase relative to the block start) and
// publishes its block partial; the LAST block to finish (agent-scope release -> ticket -> acquire, the hand-off
// recipe of cdna_hip_programming.md Guideline 16 in its counter form) folds the block partials
// into the running state (carry-in for streaming chunks), finishes bytes / last_byte and re-arms the ticket.
// rel_phase / block_phase (optional, histogram fold): starting phase of each range inside its block / of each block.
__global__ __launch_bounds__(kFold1) void fq_fold_fused(const uint64_t* partials, uint64_t n_ranges, uint64_t* block_out,
                                                         uint32_t* ticket, uint64_t* state, int reset,
                                                         uint8_t* rel_phase, uint8_t* block_phase,
                                                         const uint8_t* base, uint64_t n) {
  __shared__ uint64_t acc[kPartialWords];
  __shared__ uint32_t wave_tot[kFold1 / 64];
  __shared__ uint64_t rows[kFold1][W_BYTES + 1];
  __shared__ uint32_t is_last;
  const int tid = threadIdx.x;
  const uint32_t n_blocks = gridDim.x;
  if (tid < kPartialWords) acc[tid] = 0;
  __syncthreads();
  {
    const uint64_t r = (uint64_t)blockIdx.x * kFold1 + tid;
    uint64_t mine[W_BYTES];
    const uint64_t* src = partials + r * kPartialWords;
#pragma unroll
    for (int k = 0; k < W_BYTES; ++k) mine[k] = (r < n_ranges) ? src[k] : 0;
    const uint32_t ph = block_rotate_sum<kFold1>(mine, 0u, acc, wave_tot, rows, tid);
    if (rel_phase && r < n_ranges) rel_phase[r] = (uint8_t)ph;
    if (tid < kPartialWords) block_out[(uint64_t)blockIdx.x * kPartialWords + tid] = (tid < W_BYTES) ? acc[tid] : 0;
  }
  // ---- publish, take a ticket --------------------------------------------------------------------------
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint32_t t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    is_last = (t == n_blocks - 1) ? 1u : 0u;
    if (is_last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
  if (!is_last) return;
  // ---- last block: fold the block partials (kFold1 threads, ceil(n_blocks / kFold1) passes of <= kFold1 each) ---
  uint32_t phase_in = reset ? 0u : ((uint32_t)state[W_NL] & 3u);
  uint64_t run = 0;        // thread w < W_BYTES keeps word w of the running fold
  const uint64_t old = (!reset && tid < W_BYTES) ? state[tid] : 0;
  for (uint32_t b0 = 0; b0 < n_blocks; b0 += kFold1) {
    if (tid < kPartialWords) acc[tid] = 0;
    __syncthreads();
    const uint32_t b = b0 + tid;
    uint64_t mine[W_BYTES];
    const uint64_t* src = block_out + (uint64_t)b * kPartialWords;
#pragma unroll
    for (int k = 0; k < W_BYTES; ++k) mine[k] = (b < n_blocks) ? src[k] : 0;
    const uint32_t ph = block_rotate_sum<kFold1>(mine, phase_in, acc, wave_tot, rows, tid);
    if (block_phase && b < n_blocks) block_phase[b] = (uint8_t)ph;
    if (tid < W_BYTES) run += acc[tid];
    phase_in = (phase_in + (uint32_t)acc[W_NL]) & 3u;
    __syncthreads();
  }
  if (tid < W_BYTES) state[tid] = old + run;   // class words were already rotated by the carried-in phase
  if (tid == 0) {
    state[W_BYTES] = (reset ? 0 : state[W_BYTES]) + n;
    if (n) state[W_LAST] = base[n - 1]; else if (reset) state[W_LAST] = 0;
    *ticket = 0;   // re-arm for the next launch (stream order makes this visible to it)
  }
}

// K3 fold: state_hist[c][b] += sum_r hist_r[(c - phase_r) & 3][b],  phase_r = block_phase[r / kFold1] + rel_phase[r].
// One block per kFold1 ranges; thread t owns byte values t (4 classes each), reads are coalesced over t.
// todo (optional): only the ranges the exact kernel has (re)done carry a partial.
__global__ __launch_bounds__(256) void fq_fold_hist(const uint32_t* hist_partials, const uint8_t* rel_phase,
                                                    const uint8_t* block_phase, uint64_t n_ranges, const uint8_t* todo,
                                                    unsigned long long* state_hist) {
  const uint32_t b = threadIdx.x;
  const uint64_t r0 = (uint64_t)blockIdx.x * kFold1;
  const uint64_t r1 = (r0 + kFold1 < n_ranges) ? r0 + kFold1 : n_ranges;
  const uint32_t bp = block_phase[blockIdx.x];
  uint64_t acc[4] = {0, 0, 0, 0};
  bool any = false;
  for (uint64_t r = r0; r < r1; ++r) {
    if (todo && !todo[r]) continue;
    any = true;
    const uint32_t ph = (bp + rel_phase[r]) & 3u;
    const uint32_t* src = hist_partials + r * 1024 + b;
#pragma unroll
    for (uint32_t c = 0; c < 4; ++c)   // u32 partials are modular (take-backs may wrap): sign-extend
      acc[(c + ph) & 3u] += (uint64_t)(int64_t)(int32_t)src[c * 256];
  }
  if (!any) return;
#pragma unroll
  for (uint32_t c = 0; c < 4; ++c) atomicAdd(&state_hist[c * 256 + b], (unsigned long long)acc[c]);
}

// ---- K3 speculative form: verification of the guesses against the exact phases of the ordered fold ------------------
// ext[0] = 0: the session has no hypothesis yet; H + 1: header lines are class H relative to the session start.
// ext[1] / ext[2] = ranges whose histogram came from the fast form / had to be (re)done exactly (diagnostics).
// A guess g of range r says "header lines are class g relative to r"; with the exact starting phase p_r of the range
// (newlines before it, mod 4) that is class (g + p_r) & 3 relative to the session.  All guesses must agree on one H:
// force_h >= 0 when the session starts at the start of the input (H = 0 by definition), else H is taken from the
// first range that has a guess (and is checked again by the host when shards are combined / finalized).  A workgroup
// (kQWaves consecutive ranges, one shared histogram) with any disagreeing range is discarded as a whole; its ranges and all
// ranges without a guess are marked todo for the exact kernel.
constexpr int kExtH = 0, kExtFast = 1, kExtRedo = 2, kExtWords = 8;
constexpr uint64_t kFoldWgPer = 16;    // workgroup histograms summed per block of fq_fold_hist_wg
__global__ __launch_bounds__(256) void fq_hist_verify(const uint8_t* guess, const uint8_t* rel_phase, const uint8_t* block_phase,
                                                      uint64_t n_ranges, int force_h, uint64_t* ext, uint8_t* todo,
                                                      uint8_t* wg_ok, const uint32_t* hist_wg) {
  __shared__ uint32_t s_first, s_fast, s_redo;
  const uint32_t tid = threadIdx.x;
  if (tid == 0) { s_first = 0xFFFFFFFFu; s_fast = 0; s_redo = 0; }
  __syncthreads();
  auto session_class = [&](uint64_t r) { return ((uint32_t)guess[r] + block_phase[r / kFold1] + rel_phase[r]) & 3u; };
  // every block derives the same H (a pure function of the inputs), so blocks never wait for each other
  uint32_t H = (force_h >= 0) ? (uint32_t)force_h : ((uint32_t)ext[kExtH] ? (uint32_t)ext[kExtH] - 1u : kNoGuess);
  if (H == kNoGuess) {     // block-uniform
    for (uint64_t r0 = 0; r0 < n_ranges; r0 += 256) {
      const uint64_t r = r0 + tid;
      if (r < n_ranges && guess[r] != kNoGuess) atomicMin(&s_first, (uint32_t)r);
      __syncthreads();
      if (s_first != 0xFFFFFFFFu) break;
    }
    if (s_first != 0xFFFFFFFFu) H = session_class(s_first);
  }
  const uint64_t n_wg = (n_ranges + kQWaves - 1) / kQWaves;
  uint32_t fast = 0, redo = 0;
  const uint64_t b = (uint64_t)blockIdx.x * 256 + tid;
  if (b < n_wg) {
    const uint64_t r0 = b * kQWaves, r1 = (r0 + kQWaves < n_ranges) ? r0 + kQWaves : n_ranges;
    // (window layout: a workgroup that met a quality byte its 128 bins cannot hold says so in bin 255)
    bool poisoned = kQWindow && hist_wg[b * 256 + 255] == kQPoison;
    for (uint64_t r = r0; r < r1; ++r)
      if (guess[r] != kNoGuess && session_class(r) != H) poisoned = true;
    for (uint64_t r = r0; r < r1; ++r) {
      const bool again = poisoned || guess[r] == kNoGuess;
      todo[r] = again ? 1 : 0;
      if (again) ++redo; else ++fast;
    }
    wg_ok[b] = poisoned ? 0 : 1;
  }
  if (fast) atomicAdd(&s_fast, fast);
  if (redo) atomicAdd(&s_redo, redo);
  __syncthreads();
  if (tid == 0) {
    if (H != kNoGuess && s_fast) ext[kExtH] = H + 1u;      // every writer stores the same value
    if (s_fast) atomicAdd(reinterpret_cast<unsigned long long*>(&ext[kExtFast]), (unsigned long long)s_fast);
    if (s_redo) atomicAdd(reinterpret_cast<unsigned long long*>(&ext[kExtRedo]), (unsigned long long)s_redo);
  }
}

// state_hist[(H + 3) & 3][b] += sum over the verified workgroups of hist_wg[.][b]   (u32 modular partials: sign-extend)
__global__ __launch_bounds__(256) void fq_fold_hist_wg(const uint32_t* hist_wg, const uint8_t* wg_ok, uint64_t n_wg,
                                                       const uint64_t* ext, unsigned long long* state_hist) {
  constexpr uint64_t kPer = kFoldWgPer;
  const uint32_t t = threadIdx.x;
  if (!ext[kExtH]) return;                      // no range was taken from the fast form
  const uint32_t q = ((uint32_t)ext[kExtH] - 1u + 3u) & 3u;
  const uint64_t b0 = (uint64_t)blockIdx.x * kPer, b1 = (b0 + kPer < n_wg) ? b0 + kPer : n_wg;
  uint64_t acc = 0;
  for (uint64_t b = b0; b < b1; ++b)
    if (wg_ok[b]) acc += (uint64_t)(int64_t)(int32_t)hist_wg[b * 256 + t];
  if (acc) atomicAdd(&state_hist[q * 256 + t], (unsigned long long)acc);
}

// ------------------------------------------------------------------------------------------------
// K5: line index (record-boundary detection).  line_off[j] = offset, relative to the first byte of the input, of the
// first byte of line j.  Lines are what the reference's `lines(stream)` yields (src/fq_count.nim:38, src/fq_dedup.nim:42):
// every '\n' ends one; record i of a FASTQ is lines 4i .. 4i+3.  Two passes over HBM: K1 + K2 give every range its
// newline count, fq_nl_prefix turns those into the ordinal of the first line start each range will emit, and
// fq_index_lines re-scans the newlines (same tiles, same LDS-DMA ring) and scatters `position of '\n' + 1` to
// line_off[ordinal]: wave prefix-sum of the per-lane newline counts (DPP) + a wave-uniform running total.  Lanes that
// hold the k-th newline of consecutive lines write consecutive 8-byte slots, so the stores coalesce.
// ------------------------------------------------------------------------------------------------

// exclusive prefix sum of the per-range newline counts (one block; n_ranges is a few 10^4): first_ord[r] = line_base + 1 +
// number of '\n' in ranges < r  (the first '\n' of range r starts line first_ord[r]); first_ord[n_ranges] = total + line_base + 1
__global__ __launch_bounds__(1024) void fq_nl_prefix(const uint64_t* partials, uint64_t n_ranges, uint64_t line_base,
                                                     uint64_t* first_ord, uint32_t stride = kPartialWords) {
  __shared__ uint64_t wave_tot[16];
  __shared__ uint64_t carry;
  const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid == 0) carry = line_base + 1;
  __syncthreads();
  for (uint64_t r0 = 0; r0 < n_ranges; r0 += 1024) {
    const uint64_t r = r0 + tid;
    const uint32_t nl = (r < n_ranges) ? (uint32_t)partials[r * stride + W_NL] : 0u;   // <= 4096 * kMaxTilesPerRange (stride 1: a plain array of counts)
    const uint32_t incl = wave_inclusive_scan(nl);
    if (lane == 63) wave_tot[w] = incl;
    __syncthreads();
    uint64_t before = carry;
    for (uint32_t k = 0; k < w; ++k) before += wave_tot[k];
    if (r < n_ranges) first_ord[r] = before + incl - nl;
    __syncthreads();
    if (tid == 1023) carry = before + incl;
    __syncthreads();
  }
  if (tid == 0) first_ord[n_ranges] = carry;
}

struct IndexArgs {
  const uint8_t* base;        // first byte of the input (any alignment)
  uint64_t n;                 // bytes
  uint32_t tiles_per_range;   // the same ranges as the scan that produced the partials
  uint64_t n_ranges;
  const uint64_t* first_ord;  // fq_nl_prefix
  uint64_t* line_off;         // [>= total lines + 1]
  uint64_t off_base;          // offset of base[0] in the whole input (streaming chunks)
};

__global__ __launch_bounds__(64 * kWavesPerBlock) void fq_index_lines(IndexArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int lane = threadIdx.x & 63;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  uint8_t* ring = smem + wave * (2 * kTile);
  const uint32_t ring_lds = (uint32_t)(uintptr_t)ring;
  const uint64_t range = (uint64_t)blockIdx.x * kWavesPerBlock + wave;
  if (range >= a.n_ranges) return;
  const uint64_t B = (uint64_t)(uintptr_t)a.base, E = B + a.n;
  const uint64_t A0 = B & ~(uint64_t)(kTile - 1);
  const uint32_t n_tiles = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((E - A0 + kTile - 1) / kTile));
  const uint32_t t_begin = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(range * a.tiles_per_range));
  uint32_t t_end = t_begin + a.tiles_per_range;
  if (t_end > n_tiles) t_end = n_tiles;
  t_end = (uint32_t)__builtin_amdgcn_readfirstlane((int)t_end);
  const uint32_t full_lo = (uint32_t)__builtin_amdgcn_readfirstlane((A0 < B) ? 1 : 0);
  const uint32_t full_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)((A0 + (uint64_t)n_tiles * kTile > E) ? n_tiles - 1 : n_tiles));
  // ordinal of the line that the first '\n' of this range starts; pinned before the first DMA (see fq_scan_tiles)
  uint64_t ord = a.first_ord[range];
  ord = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(ord >> 32)) << 32) |
        (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ord);
  PlaneConsts pc;
  pc.init();

  auto issue = [&](uint32_t t, uint32_t slot) {
    const uint64_t ts = A0 + (uint64_t)t * kTile;
    const uint32_t dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)(ring_lds + slot * kTile));
    if (t >= full_lo && t < full_hi) {
      glds_tile<true>(reinterpret_cast<const uint8_t*>(ts + (uint64_t)lane * 16), dst);
    } else {
      const uint64_t safe = (B & ~15ull);
      uint64_t s[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint64_t ps = ts + (uint64_t)k * 1024 + (uint64_t)lane * 16;
        const bool ok = (ps + 16 > B) && (ps < E);
        s[k] = (ok ? ps : safe) - (uint64_t)k * 1024;
      }
      glds_tile_edge(reinterpret_cast<const uint8_t*>(s[0]), reinterpret_cast<const uint8_t*>(s[1]),
                     reinterpret_cast<const uint8_t*>(s[2]), reinterpret_cast<const uint8_t*>(s[3]), dst);
    }
  };

  if (t_begin < t_end) issue(t_begin, 0);
  uint32_t slot = 0;
  for (uint32_t t = t_begin; t < t_end; ++t) {
    if (t + 1 < t_end) { issue(t + 1, slot ^ 1u); wait_vmcnt<4>(); } else { wait_vmcnt<0>(); }
    const uint4* p = reinterpret_cast<const uint4*>(ring + slot * kTile + lane * 64);
    const uint4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
    uint32_t d[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
    uint32_t a0, a1, a2, a3, a4, b0, b1, b2, b3, b4;
    masks32_planes<false>(d, pc, a0, a1, a2, a3, a4);
    masks32_planes<false>(d + 8, pc, b0, b1, b2, b3, b4);
    uint64_t NL = ~((uint64_t)a0 | ((uint64_t)b0 << 32));
    const uint64_t ts = A0 + (uint64_t)t * kTile;
    if (!(t >= full_lo && t < full_hi)) {     // first / last tile of the input: only bytes inside [B, E) exist
      const int64_t ls = (int64_t)(ts + (uint64_t)lane * 64);
      int64_t lo = (int64_t)B - ls, hi = (int64_t)E - ls;
      lo = lo < 0 ? 0 : (lo > 64 ? 64 : lo);
      hi = hi < 0 ? 0 : (hi > 64 ? 64 : hi);
      const uint64_t mhi = (hi >= 64) ? ~0ull : ((1ull << hi) - 1);
      const uint64_t mlo = (lo >= 64) ? ~0ull : ((1ull << lo) - 1);
      NL &= mhi & ~mlo;
    }
    const uint32_t cnt = popc64(NL);
    const uint32_t incl = wave_inclusive_scan(cnt);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    if (total) {      // wave-uniform
      uint64_t o = ord + (incl - cnt);
      // offset of the byte AFTER bit k of this lane, relative to the whole input
      const uint64_t lane_off = a.off_base + (ts + (uint64_t)lane * 64 - B) + 1;
      uint64_t x = NL;
      while (x) {
        const uint32_t k = (uint32_t)__builtin_ctzll(x);
        x &= x - 1;
        a.line_off[o++] = lane_off + k;
      }
    }
    ord += total;
    slot ^= 1u;
  }
}

// ------------------------------------------------------------------------------------------------
// K5, one pass over the INPUT (the default): fq_index_masks streams the input once through the LDS-DMA ring (the ranges of
// K1: one wave, ~100 consecutive tiles) and keeps what the index needs of it — the newline mask of every lane's 64 bytes
// (one bit per input byte, 512 B per tile, coalesced) and the range's newline count; fq_nl_prefix turns the counts into
// first ordinals; fq_index_expand walks the MASKS (an eighth of the input) and scatters `position of '\n' + 1`.
// Algorithmic bytes: input x (1 + 1/8 + 1/8) + 8 B per line = 1.34 x input for 150 bp reads, against 2.1 x for the form
// that reads the input twice.  (A single kernel with a decoupled look-back over the ranges' counts was built and measured
// first: with 4096 ranges in flight a range's look-back walks 64 dependent steps of 64 predecessors and the kernel ran at
// 0.33 TB/s; ranges long enough to hide that do not fit their masks into registers.)
// ------------------------------------------------------------------------------------------------
struct IndexMaskArgs {
  const uint8_t* base;        // first byte of the input (any alignment)
  uint64_t n;                 // bytes
  uint32_t tiles_per_range;
  uint64_t n_ranges;
  uint64_t* masks;            // [n_tiles][64]: newline mask of lane L's 64 bytes of tile t at masks[t * 64 + L]
  uint64_t* counts;           // [n_ranges]
  uint32_t* flags_out;        // optional: bit 0 is set when the input may hold a '\r' directly before a '\n' (conservative: a '\r' in
                              // the last byte of a lane's 64 counts); fq-dedup skips its "\r\n" look-behind reads when it stays clear
};

__global__ __launch_bounds__(64 * kWavesPerBlock) void fq_index_masks(IndexMaskArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int lane = threadIdx.x & 63;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  uint8_t* ring = smem + wave * (2 * kTile);
  const uint32_t ring_lds = (uint32_t)(uintptr_t)ring;
  const uint64_t range = (uint64_t)blockIdx.x * kWavesPerBlock + wave;
  if (range >= a.n_ranges) return;
  const uint64_t B = (uint64_t)(uintptr_t)a.base, E = B + a.n;
  const uint64_t A0 = B & ~(uint64_t)(kTile - 1);
  const uint32_t n_tiles = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((E - A0 + kTile - 1) / kTile));
  const uint32_t t_begin = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(range * a.tiles_per_range));
  uint32_t t_end = t_begin + a.tiles_per_range;
  if (t_end > n_tiles) t_end = n_tiles;
  t_end = (uint32_t)__builtin_amdgcn_readfirstlane((int)t_end);
  const uint32_t full_lo = (uint32_t)__builtin_amdgcn_readfirstlane((A0 < B) ? 1 : 0);
  const uint32_t full_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)((A0 + (uint64_t)n_tiles * kTile > E) ? n_tiles - 1 : n_tiles));
  const bool want_cr = (bool)__builtin_amdgcn_readfirstlane(a.flags_out != nullptr ? 1 : 0);
  PlaneConsts pc;
  pc.init();

  auto issue = [&](uint32_t t, uint32_t slot) {
    const uint64_t ts = A0 + (uint64_t)t * kTile;
    const uint32_t dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)(ring_lds + slot * kTile));
    if (t >= full_lo && t < full_hi) {
      glds_tile<true>(reinterpret_cast<const uint8_t*>(ts + (uint64_t)lane * 16), dst);
    } else {
      const uint64_t safe = (B & ~15ull);
      uint64_t s[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint64_t ps = ts + (uint64_t)k * 1024 + (uint64_t)lane * 16;
        const bool ok = (ps + 16 > B) && (ps < E);
        s[k] = (ok ? ps : safe) - (uint64_t)k * 1024;
      }
      glds_tile_edge(reinterpret_cast<const uint8_t*>(s[0]), reinterpret_cast<const uint8_t*>(s[1]),
                     reinterpret_cast<const uint8_t*>(s[2]), reinterpret_cast<const uint8_t*>(s[3]), dst);
    }
  };

  uint64_t cr_seen = 0;
  uint32_t cnt = 0;                                 // per lane; 64 per tile at most, kMaxTilesPerRange tiles
  if (t_begin < t_end) issue(t_begin, 0);
  uint32_t slot = 0;
  for (uint32_t t = t_begin; t < t_end; ++t) {
    if (t + 1 < t_end) { issue(t + 1, slot ^ 1u); wait_vmcnt<4>(); } else { wait_vmcnt<0>(); }
    const uint4* p = reinterpret_cast<const uint4*>(ring + slot * kTile + lane * 64);
    const uint4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
    uint32_t d[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
    uint32_t a0, a1, a2, a3, a4, b0, b1, b2, b3, b4, xa[8], xb[8];
    masks32_planes_x<false>(d, pc, xa, a0, a1, a2, a3, a4);
    masks32_planes_x<false>(d + 8, pc, xb, b0, b1, b2, b3, b4);
    uint64_t NL = ~((uint64_t)a0 | ((uint64_t)b0 << 32));
    if (want_cr) {
      const uint64_t CR = ~((uint64_t)plane_ne<0x0D>(xa) | ((uint64_t)plane_ne<0x0D>(xb) << 32));
      cr_seen |= ((CR << 1) & NL) | (CR >> 63);
    }
    if (!(t >= full_lo && t < full_hi)) {     // first / last tile of the input: only bytes inside [B, E) exist
      const uint64_t ts = A0 + (uint64_t)t * kTile;
      const int64_t ls = (int64_t)(ts + (uint64_t)lane * 64);
      int64_t lo = (int64_t)B - ls, hi = (int64_t)E - ls;
      lo = lo < 0 ? 0 : (lo > 64 ? 64 : lo);
      hi = hi < 0 ? 0 : (hi > 64 ? 64 : hi);
      const uint64_t mhi = (hi >= 64) ? ~0ull : ((1ull << hi) - 1);
      const uint64_t mlo = (lo >= 64) ? ~0ull : ((1ull << lo) - 1);
      NL &= mhi & ~mlo;
    }
    __builtin_nontemporal_store(NL, &a.masks[(uint64_t)t * 64 + lane]);
    cnt += popc64(NL);
    slot ^= 1u;
  }
  const uint32_t total = wave_sum(cnt);
  if (lane == 0) a.counts[range] = total;
  // (a look before the atomic: on a "\r\n" input EVERY wave would otherwise queue one on the same word)
  if (want_cr && __builtin_amdgcn_ballot_w64(cr_seen != 0) != 0 && lane == 0 &&
      !(__hip_atomic_load(a.flags_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1u))
    atomicOr(a.flags_out, 1u);
}

struct IndexExpandArgs {
  const uint64_t* masks;      // fq_index_masks
  uint64_t lead;              // bytes between the first tile's start and the input's first byte (B - A0)
  uint32_t n_tiles;
  uint32_t tiles_per_range;
  uint64_t n_ranges;
  const uint64_t* first_ord;  // fq_nl_prefix over the ranges' counts
  uint64_t* line_off;         // [cap]
  uint64_t cap;
  uint64_t off_base;          // offset of the input's first byte in the whole input (streaming chunks)
};

// one wave per range again, tile by tile over the masks: wave prefix sum of the per-lane counts + a running ordinal
__global__ __launch_bounds__(256) void fq_index_expand(IndexExpandArgs a) {
  const int lane = threadIdx.x & 63;
  const uint64_t range = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (range >= a.n_ranges) return;
  const uint32_t t_begin = (uint32_t)(range * a.tiles_per_range);
  uint32_t t_end = t_begin + a.tiles_per_range;
  if (t_end > a.n_tiles) t_end = a.n_tiles;
  uint64_t ord = a.first_ord[range];
  // the masks of the next tile are requested before this tile's offsets are written
  uint64_t x = t_begin < t_end ? __builtin_nontemporal_load(&a.masks[(uint64_t)t_begin * 64 + lane]) : 0;
  for (uint32_t t = t_begin; t < t_end; ++t) {
    const uint64_t nx = t + 1 < t_end ? __builtin_nontemporal_load(&a.masks[(uint64_t)(t + 1) * 64 + lane]) : 0;
    const uint32_t cnt = popc64(x);
    const uint32_t incl = wave_inclusive_scan(cnt);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    if (total) {      // wave-uniform
      uint64_t o = ord + (incl - cnt);
      const uint64_t lane_off = a.off_base + ((uint64_t)t * kTile + (uint64_t)lane * 64 - a.lead) + 1;   // offset of the byte AFTER bit 0 of this lane
      while (x) {
        const uint32_t k = (uint32_t)__builtin_ctzll(x);
        x &= x - 1;
        if (o < a.cap) a.line_off[o] = lane_off + k;
        ++o;
      }
    }
    ord += total;
    x = nx;
  }
}

// ------------------------------------------------------------------------------------------------
// K5, compact form (r4, the default): the pass over the input writes the newline POSITIONS of every tile — 16-bit offsets inside the
// tile, packed, behind their count — instead of a bit per byte: ~2 x 46 bytes per 4 KiB tile of 150 bp FASTQ where the masks are 512,
// and the second kernel turns them into offsets with one coalesced 8-byte store per line where the mask form's lanes each walk their
// own bits.  A tile with more than kPosCap - 1 newlines (lines shorter than 33 bytes on average) raises a flag: the caller then runs
// the mask form (above), which has no such limit.
// Algorithmic bytes: input x (1 + 2 x ~0.023) + 8 B per line = 1.14 x input for 150 bp reads (mask form: 1.34 x).
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kPosCap = 128;     // uint16 entries per tile: [0] the count, [1 ..] the positions
constexpr uint32_t kPosHashCap = 32;  // hashed lines per tile (fq-dedup): a tile of 150 bp FASTQ holds 11 or 12 headers
constexpr uint32_t kPosStage = 2 * kPosCap + 4 * kPosHashCap + 8 * kPosHashCap;      // a wave's staging: the slot, the (start, length) list, the hashes
constexpr uint32_t kIndexPosLds = kWavesPerBlock * (2 * kTile + kPosStage);      // fq_index_pos: the waves' tile rings + their staging areas

// fq-dedup rides along (IndexPosArgs::hash_at): the lines of this tile that start with '@', begin behind one of its newlines and end at
// the next are hashed HERE, where their bytes are in LDS — lane j looks at the line behind newline j (entries j and j + 1 of the staged
// positions); the lines found are listed, and four lanes take a line each round, every fourth 8-byte word per lane, summed over the four
// with two cross-lane steps: the hash is a sum over words (scfq_hdrhash.hpp).
// The hash kernel of fq-dedup read 1.4 lines of 128 bytes per 57-byte header — 4 - 5 GB for 10 GB of input — to do the same.
__device__ __forceinline__ void index_hash_lines(const uint8_t* tile, uint16_t* stage, uint32_t* hl, uint64_t* hout, uint32_t total, int lane,
                                                 uint64_t* out) {
  const uint32_t j = (uint32_t)lane;
  bool cand = j >= 1u && j + 1u <= total;
  uint32_t S = 0, len = 0;
  if (cand) {
    const uint32_t p0 = stage[j], p1 = stage[j + 1u];
    S = p0 + 1u;
    uint32_t e = p1;
    if (S < e && tile[e - 1u] == '\r') --e;       // (Nim's readLine: "\r\n" ends a line as "\n" does)
    len = e - S;
    cand = S < p1 && tile[S] == '@' && len <= scfq_hdrhash::kMaxLen;
  }
  const uint64_t hmask = __builtin_amdgcn_ballot_w64(cand);
  if (hmask == 0) return;
  const uint32_t rank = (uint32_t)__builtin_popcountll(hmask & ((1ull << lane) - 1ull));
  cand = cand && rank < kPosHashCap;
  if (cand) { hl[rank] = S | len << 16; stage[j] = (uint16_t)(stage[j] | 0x8000u); }
  uint32_t n_h = (uint32_t)__builtin_popcountll(hmask);
  if (n_h > kPosHashCap) n_h = kPosHashCap;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const uint32_t k0 = (uint32_t)lane & 3u;
  for (uint32_t r0 = 0; r0 < n_h; r0 += 16u) {      // wave-uniform: sixteen lines a round, four lanes each (words k0, k0 + 4, ...)
    const uint32_t hh = r0 + ((uint32_t)lane >> 2);
    uint32_t A = 0, B = 0, ln = 0;
    if (hh < n_h) {
      const uint32_t v = hl[hh];
      const uint32_t Sl = v & 0xFFFFu;
      ln = v >> 16;
      const uint32_t n_words = (ln + 7u) >> 3;
      for (uint32_t kk = k0; kk < n_words; kk += 4u) {
        // the word's eight bytes from LDS: three aligned dwords and two byte alignments (the line starts anywhere)
        const uint32_t at = Sl + 8u * kk;
        const uint32_t* q = reinterpret_cast<const uint32_t*>(tile + (at & ~3u));
        const uint32_t d0 = q[0], d1 = q[1], d2 = q[2];      // (may reach past the tile's end: still this workgroup's LDS, masked below)
        const uint32_t sh = at & 3u;
        uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, sh), hi = __builtin_amdgcn_alignbyte(d2, d1, sh);
        const uint32_t valid = ln - 8u * kk;                 // >= 1
        if (valid < 4u) { lo &= (1u << (8u * valid)) - 1u; hi = 0; }
        else if (valid == 4u) hi = 0;
        else if (valid < 8u) hi &= (1u << (8u * (valid - 4u))) - 1u;
        scfq_hdrhash::hh_word(lo, hi, kk, A, B);
      }
    }
    // the four lanes of a line add up (DPP quad permutes: the neighbour in the pair, then the other pair); the mixing of the sums is
    // left to the kernel that hands the keys out (fq_index_expand_pos: one lane per header there)
    A += (uint32_t)__builtin_amdgcn_mov_dpp((int)A, 0xB1, 0xF, 0xF, true); B += (uint32_t)__builtin_amdgcn_mov_dpp((int)B, 0xB1, 0xF, 0xF, true);
    A += (uint32_t)__builtin_amdgcn_mov_dpp((int)A, 0x4E, 0xF, 0xF, true); B += (uint32_t)__builtin_amdgcn_mov_dpp((int)B, 0x4E, 0xF, 0xF, true);
    if (hh < n_h && k0 == 0u) hout[hh] = (uint64_t)A | (uint64_t)(B & 0xFFFFFFu) << 32 | (uint64_t)ln << scfq_hdrhash::kHashBits;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if ((uint32_t)lane < n_h) out[lane] = hout[lane];
  asm volatile("" ::: "memory");
}

struct IndexPosArgs {
  const uint8_t* base;        // first byte of the input (any alignment)
  uint64_t n;                 // bytes
  uint32_t tiles_per_range;
  uint64_t n_ranges;
  uint16_t* pos;              // [n_tiles][kPosCap]
  uint64_t* counts;           // [n_ranges]
  uint32_t* flags;            // bit 0: the input may hold a '\r' directly before a '\n' (as fq_index_masks; only when want_cr); bit 1: a tile overflowed
  uint32_t want_cr;
  // fq-dedup (optional): the hashes of the lines that start with '@' and lie inside ONE tile (start behind a newline of the tile, end
  // at the next one, at most 255 bytes), in tile order, at most kPosHashCap per tile: [n_tiles][kPosHashCap] of A | (B & 2^24 - 1) << 32 |
  // length << 56 (scfq_hdrhash.hpp: the sums over the line's words); the entry of the newline in front of such a line carries bit 15
  uint64_t* hash_at;
  uint64_t hash_seed;
};

__global__ __launch_bounds__(64 * kWavesPerBlock) void fq_index_pos(IndexPosArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int lane = threadIdx.x & 63;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  uint8_t* ring = smem + wave * (2 * kTile);
  uint16_t* stage = reinterpret_cast<uint16_t*>(smem + kWavesPerBlock * (2 * kTile) + wave * kPosStage);      // (launch: kIndexPosLds bytes)
  uint32_t* hl = reinterpret_cast<uint32_t*>(stage + kPosCap);                 // fq-dedup: (start | length << 16) of the lines to hash
  uint64_t* hout = reinterpret_cast<uint64_t*>(hl + kPosHashCap);              // ... and their hashes
  const bool want_hash = (bool)__builtin_amdgcn_readfirstlane(a.hash_at != nullptr ? 1 : 0);
  const uint32_t ring_lds = (uint32_t)(uintptr_t)ring;
  const uint64_t range = (uint64_t)blockIdx.x * kWavesPerBlock + wave;
  if (range >= a.n_ranges) return;
  const uint64_t B = (uint64_t)(uintptr_t)a.base, E = B + a.n;
  const uint64_t A0 = B & ~(uint64_t)(kTile - 1);
  const uint32_t n_tiles = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((E - A0 + kTile - 1) / kTile));
  const uint32_t t_begin = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(range * a.tiles_per_range));
  uint32_t t_end = t_begin + a.tiles_per_range;
  if (t_end > n_tiles) t_end = n_tiles;
  t_end = (uint32_t)__builtin_amdgcn_readfirstlane((int)t_end);
  const uint32_t full_lo = (uint32_t)__builtin_amdgcn_readfirstlane((A0 < B) ? 1 : 0);
  const uint32_t full_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)((A0 + (uint64_t)n_tiles * kTile > E) ? n_tiles - 1 : n_tiles));
  const bool want_cr = (bool)__builtin_amdgcn_readfirstlane((int)a.want_cr);
  PlaneConsts pc;
  pc.init();

  auto issue = [&](uint32_t t, uint32_t slot) {
    const uint64_t ts = A0 + (uint64_t)t * kTile;
    const uint32_t dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)(ring_lds + slot * kTile));
    if (t >= full_lo && t < full_hi) {
      glds_tile<true>(reinterpret_cast<const uint8_t*>(ts + (uint64_t)lane * 16), dst);
    } else {
      const uint64_t safe = (B & ~15ull);
      uint64_t s[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint64_t ps = ts + (uint64_t)k * 1024 + (uint64_t)lane * 16;
        const bool ok = (ps + 16 > B) && (ps < E);
        s[k] = (ok ? ps : safe) - (uint64_t)k * 1024;
      }
      glds_tile_edge(reinterpret_cast<const uint8_t*>(s[0]), reinterpret_cast<const uint8_t*>(s[1]),
                     reinterpret_cast<const uint8_t*>(s[2]), reinterpret_cast<const uint8_t*>(s[3]), dst);
    }
  };

  uint64_t cr_seen = 0;
  uint32_t range_total = 0;                         // wave-uniform
  bool overflow = false;                            // wave-uniform
  if (t_begin < t_end) issue(t_begin, 0);
  uint32_t slot = 0;
  for (uint32_t t = t_begin; t < t_end; ++t) {
    if (t + 1 < t_end) { issue(t + 1, slot ^ 1u); wait_vmcnt<4>(); } else { wait_vmcnt<0>(); }
    const uint4* p = reinterpret_cast<const uint4*>(ring + slot * kTile + lane * 64);
    const uint4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
    uint32_t d[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
    uint32_t a0, a1, a2, a3, a4, b0, b1, b2, b3, b4, xa[8], xb[8];
    masks32_planes_x<false>(d, pc, xa, a0, a1, a2, a3, a4);
    masks32_planes_x<false>(d + 8, pc, xb, b0, b1, b2, b3, b4);
    uint64_t NL = ~((uint64_t)a0 | ((uint64_t)b0 << 32));
    if (want_cr) {
      const uint64_t CR = ~((uint64_t)plane_ne<0x0D>(xa) | ((uint64_t)plane_ne<0x0D>(xb) << 32));
      cr_seen |= ((CR << 1) & NL) | (CR >> 63);
    }
    if (!(t >= full_lo && t < full_hi)) {     // first / last tile of the input: only bytes inside [B, E) exist
      const uint64_t ts = A0 + (uint64_t)t * kTile;
      const int64_t ls = (int64_t)(ts + (uint64_t)lane * 64);
      int64_t lo = (int64_t)B - ls, hi = (int64_t)E - ls;
      lo = lo < 0 ? 0 : (lo > 64 ? 64 : lo);
      hi = hi < 0 ? 0 : (hi > 64 ? 64 : hi);
      const uint64_t mhi = (hi >= 64) ? ~0ull : ((1ull << hi) - 1);
      const uint64_t mlo = (lo >= 64) ? ~0ull : ((1ull << lo) - 1);
      NL &= mhi & ~mlo;
    }
    const uint32_t cnt = popc64(NL);
    const uint32_t incl = wave_inclusive_scan(cnt);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    uint32_t* out = reinterpret_cast<uint32_t*>(a.pos + (uint64_t)t * kPosCap);      // (256-byte aligned: a slot is 128 x 2 bytes)
    if (total < kPosCap) {                    // wave-uniform
      // the lanes put their positions into the wave's 256 bytes of LDS, in order, and the wave stores the slot's used part as whole
      // dwords, coalesced (stored straight from the lanes — 2-byte stores to scattered entries, three store instructions per tile —
      // the kernel was SLOWER than the mask form, 2.03 against 1.90 ms, with a third of its writes)
      if (lane == 0) stage[0] = (uint16_t)total;
      uint32_t o = incl - cnt + 1u;
      uint64_t x = NL;
      while (__builtin_amdgcn_ballot_w64(x != 0) != 0) {      // as many rounds as the fullest lane holds newlines (two or three for FASTQ)
        if (x) {
          stage[o++] = (uint16_t)((uint32_t)lane * 64u + (uint32_t)__builtin_ctzll(x));
          x &= x - 1;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (one wave: its LDS operations complete in order; nothing is read before they have)
      if (want_hash && total >= 2u) index_hash_lines(ring + slot * kTile, stage, hl, hout, total, lane, a.hash_at + (uint64_t)t * kPosHashCap);
      if ((uint32_t)lane < (total + 2u) / 2u) __builtin_nontemporal_store(reinterpret_cast<const uint32_t*>(stage)[lane], &out[lane]);
      asm volatile("" ::: "memory");
    } else {
      overflow = true;
      if (lane == 0) out[0] = 0xFFFFu;
    }
    range_total += total;
    slot ^= 1u;
  }
  if (lane == 0) a.counts[range] = range_total;
  // (a look before each atomic: on a "\r\n" input EVERY wave would otherwise queue one on the same word)
  uint32_t bits = (want_cr && __builtin_amdgcn_ballot_w64(cr_seen != 0) != 0) ? 1u : 0u;
  if (overflow) bits |= 2u;
  if (bits && lane == 0 && (__hip_atomic_load(a.flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bits) != bits) atomicOr(a.flags, bits);
}

struct IndexExpandPosArgs {
  const uint16_t* pos;        // fq_index_pos
  const uint32_t* flags;      // ... bit 1: some tile overflowed, nothing is written here (the caller runs the mask form)
  uint64_t lead;              // bytes between the first tile's start and the input's first byte (B - A0)
  uint32_t n_tiles;
  uint32_t tiles_per_range;
  uint64_t n_ranges;
  const uint64_t* first_ord;  // fq_nl_prefix over the ranges' counts
  uint64_t* line_off;         // [cap]
  uint64_t cap;
  uint64_t off_base;
  // fq-dedup (optional; hash_at as written by fq_index_pos): line 4r is record r's header — its key (the hash's low hash_bits, or all
  // ones = "not hashed by the index pass": dd_hash_headers does those), its number, and its (start | length << 40)
  const uint64_t* hash_at;
  void* keys;                 // uint32_t[cap_records] (key_bytes == 4) or uint64_t[cap_records]
  uint32_t* idx;
  uint64_t* hdr;
  uint64_t cap_records;
  uint32_t key_bytes;
  uint32_t hash_bits;
  uint64_t hash_seed;
  uint32_t* unk;              // [n_tiles][kPosUnk]: the records of the tile that got the all-ones key (0: none) ...
  uint32_t* flags_rw;         // ... bit 2 of the index's flag word: a tile had more of them, or 64+ newlines: the list is not complete
};
constexpr uint32_t kPosUnk = 4;

__device__ __forceinline__ void index_put_record(const IndexExpandPosArgs& a, uint64_t r, uint64_t start, bool hashed, uint64_t stored) {
  if (r >= a.cap_records) return;
  const uint64_t len = hashed ? stored >> scfq_hdrhash::kHashBits : 0xFFFFFFull;      // (saturated: looked up again through the line index)
  const uint64_t h = scfq_hdrhash::hh_final((uint32_t)stored, (uint32_t)(stored >> 32) & 0xFFFFFFu, len, a.hash_seed);
  const uint64_t key = hashed ? (a.hash_bits < 64 ? h & ((1ull << a.hash_bits) - 1ull) : h) : ~0ull;
  if (a.key_bytes == 4) static_cast<uint32_t*>(a.keys)[r] = (uint32_t)key; else static_cast<uint64_t*>(a.keys)[r] = key;
  a.idx[r] = (uint32_t)r;
  a.hdr[r] = start | len << 40;
}

// one wave per range, tile by tile: lane j holds entries j and 64 + j of the tile (entry 0 is the count), entry j is line ord + j - 1
__global__ __launch_bounds__(256) void fq_index_expand_pos(IndexExpandPosArgs a) {
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t range = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (range >= a.n_ranges) return;
  if (*a.flags & 2u) return;
  const uint32_t t_begin = (uint32_t)(range * a.tiles_per_range);
  uint32_t t_end = t_begin + a.tiles_per_range;
  if (t_end > a.n_tiles) t_end = a.n_tiles;
  uint64_t ord = a.first_ord[range];
  static_assert(kPosCap == 128, "two entries per lane");
  const uint16_t* s = a.pos + (uint64_t)t_begin * kPosCap;
  // (the first half of the next tile's entries is requested before this tile's offsets are written; the second half only by a tile
  // that has that many: 46 newlines per tile in 150 bp FASTQ)
  uint32_t v0 = t_begin < t_end ? s[lane] : 0u;
  // (fq-dedup: the tile's hashes come with its entries — lane l holds hash l & 31 — so that no load waits for a rank)
  const uint64_t* hs = a.keys ? a.hash_at + (uint64_t)t_begin * kPosHashCap : nullptr;
  uint64_t h0 = (hs && t_begin < t_end) ? hs[lane & 31u] : 0;
  bool unk_over = false;      // wave-uniform
  for (uint32_t t = t_begin; t < t_end; ++t) {
    const uint16_t* cur = s;
    s += kPosCap;
    const uint32_t n0 = t + 1 < t_end ? s[lane] : 0u;
    uint64_t hn = 0;
    if (hs) { hs += kPosHashCap; if (t + 1 < t_end) hn = hs[lane & 31u]; }
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)v0, 0);
    const uint64_t tile_off = a.off_base + ((uint64_t)t * kTile - a.lead) + 1;      // offset of the byte AFTER the tile's byte 0
    const bool mine = lane >= 1u && lane <= total;
    const uint32_t pos = v0 & 0x7FFFu;                                              // (bit 15: the line behind this newline was hashed)
    if (mine) { const uint64_t o = ord + lane - 1u; if (o < a.cap) a.line_off[o] = tile_off + pos; }
    if (a.keys) {      // kernel-uniform
      const bool hashed = mine && (v0 & 0x8000u);
      const uint64_t hm = __builtin_amdgcn_ballot_w64(hashed);
      const uint32_t rank = (uint32_t)__builtin_popcountll(hm & ((1ull << lane) - 1ull)) & 31u;
      const uint64_t stored = (uint64_t)(uint32_t)__shfl((int)(uint32_t)h0, (int)rank, 64) | (uint64_t)(uint32_t)__shfl((int)(uint32_t)(h0 >> 32), (int)rank, 64) << 32;
      const uint64_t o = ord + lane - 1u;
      const bool head = mine && (o & 3u) == 0;
      if (head) index_put_record(a, o >> 2, tile_off + pos, hashed, stored);
      if (a.unk) {
        const bool un = head && !hashed;
        const uint64_t um = __builtin_amdgcn_ballot_w64(un);
        const uint32_t n_un = (uint32_t)__builtin_popcountll(um);
        uint32_t* u = a.unk + (uint64_t)t * kPosUnk;
        if (un) { const uint32_t ur = (uint32_t)__builtin_popcountll(um & ((1ull << lane) - 1ull)); if (ur < kPosUnk) u[ur] = (uint32_t)(o >> 2); }
        if (lane < kPosUnk && lane >= n_un) u[lane] = 0u;
        if (n_un > kPosUnk || total >= 64u) unk_over = true;
      }
    }
    if (total >= 64u) {      // wave-uniform
      const uint32_t v1 = cur[64 + lane] & 0x7FFFu;
      if (64u + lane <= total) {
        const uint64_t o = ord + 63u + lane;
        if (o < a.cap) a.line_off[o] = tile_off + v1;
        if (a.keys && (o & 3u) == 0) index_put_record(a, o >> 2, tile_off + v1, false, 0);
      }
    }
    ord += total;
    v0 = n0;
    h0 = hn;
  }
  if (unk_over && lane == 0u && !(__hip_atomic_load(a.flags_rw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 4u)) atomicOr(a.flags_rw, 4u);
}

// ------------------------------------------------------------------------------------------------
// Diagnostic (NOT the product path): the load structure of fq_scan_tiles alone -- same ranges, same 2-slot
// non-temporal LDS-DMA ring, one ds_read per lane and tile, no classification or accounting.  Its time is the
// practical ceiling the scan kernel is compared with on the same device (bench.py "stream_ceiling").
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fq_stream_null(const uint8_t* base, uint32_t n_tiles, uint32_t tiles_per_range,
                                                      uint32_t* sink) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int lane = threadIdx.x & 63;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  uint8_t* ring = smem + wave * (2 * kTile);
  const uint32_t ring_lds = (uint32_t)(uintptr_t)ring;
  const uint32_t range = blockIdx.x * kWavesPerBlock + wave;
  const uint32_t t0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(range * tiles_per_range));
  uint32_t t1 = t0 + tiles_per_range;
  if (t1 > n_tiles) t1 = n_tiles;
  t1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)t1);
  if (t0 >= n_tiles) return;
  uint32_t acc = 0, slot = 0;
  glds_tile<true>(base + (uint64_t)t0 * kTile + lane * 16, ring_lds);
  for (uint32_t t = t0; t < t1; ++t) {
    if (t + 1 < t1) {
      glds_tile<true>(base + (uint64_t)(t + 1) * kTile + lane * 16,
                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(ring_lds + (slot ^ 1u) * kTile)));
      wait_vmcnt<4>();
    } else {
      wait_vmcnt<0>();
    }
    acc += *reinterpret_cast<const uint32_t*>(ring + slot * kTile + lane * 64);
    slot ^= 1u;
  }
  if (acc == 0x9E3779B9u) sink[0] = acc;   // keeps the reads alive
}

// ------------------------------------------------------------------------------------------------
// Diagnostic cross-check kernel (NOT the product path): one thread scans 256 bytes byte-serially
// and emits its own partial; used by tests as an independent device implementation.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fq_scan_simple(const uint8_t* base, uint64_t n, int32_t prev_byte,
                                                      uint64_t n_chunks, uint64_t* partials) {
  const uint64_t c = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= n_chunks) return;
  const uint64_t lo = c * 256, hi = (lo + 256 < n) ? lo + 256 : n;
  uint64_t o[kPartialWords];
  for (int k = 0; k < kPartialWords; ++k) o[k] = 0;
  uint32_t r = 0;
  int prev = (lo == 0) ? prev_byte : (int)base[lo - 1];
  for (uint64_t k = lo; k < hi; ++k) {
    const uint8_t b = base[k];
    if (prev == -1 || prev == '\n') {
      o[W_STARTS + r]++;
      if (b == '@') o[W_FAT + r]++;
      if (b == '+') o[W_FPLUS + r]++;
    }
    if (b == '\n') {
      if (prev == '\r') o[W_LEN + r]--;
      o[W_NL]++;
      r = (r + 1) & 3u;
    } else {
      o[W_LEN + r]++;
      if (b == 'G' || b == 'C') o[W_GC + r]++;
      if (b == 'N') o[W_N + r]++;
    }
    prev = b;
  }
  uint64_t* out = partials + c * kPartialWords;
  for (int k = 0; k < kPartialWords; ++k) out[k] = o[k];
}

}  // namespace scfq
