// gz_inflate_kernels.hpp — device-side inflate of ORDINARY gzip members (gfx950): BASELINE configs[3], the reference's
// ".gz" path (src/fq_count.nim:30-34 -> newGZFileStream -> zlib gzread, gzip_stream.nim:16-17).
//
// A deflate stream is serial twice over: a block's bit position is only known once the block before it has been decoded,
// and a back-reference may reach into the 32 KiB before the block.  As pugz / rapidgzip (and csrc/scfq_pgz.hpp on the
// host) do it, in four kernels over the COMPRESSED bytes resident in HBM:
//   G1 gz_sync_search     one workgroup per segment scans the bit positions after the segment's nominal start for
//                         something that parses as a dynamic-Huffman block header (BTYPE, HLIT / HDIST ranges: 32 positions
//                         per lane at once from a chunk staged in LDS; a complete code-length code; the few survivors
//                         then decode the code lengths and ask for complete literal/length and distance codes and an
//                         end-of-block code).  A wrong sync never
//                         survives the host's chain check: the segment before must arrive at exactly that bit.
//   G2 gz_segment_decode  one wave per segment (the lane-parallel symbol loop of bgzf_inflate: 64 bit positions decoded at
//                         once, tables in LDS), output = 16-bit symbols behind 32768 marker symbols 0x8000 | k ("byte k of the window I do
//                         not have yet"); back-references copy symbols whether known or not.  Ends at the first block
//                         boundary at or after the next segment's start, or with the member's final block.
//   G3 gz_window_chain    the 32 KiB window in front of every segment, sequentially along the chain (one workgroup, two
//      gz_window_maps     windows in LDS): W[k+1] = last 32 KiB of (W[k] ++ symbols of k), markers resolved through W[k];
//                         long chains in three short walks over groups of 64 segments (window maps compose).
//   G4 gz_resolve         every symbol becomes a byte (markers through the segment's window), written at the segment's
//                         offset of the inflated stream; a marker that points before the member's start is corrupt data.
//   G5 gz_crc32_tiles     raw CRC-32 (zero init) of 1 MiB tiles of the inflated bytes; the host folds the tiles and checks
//                         every member's CRC-32 / ISIZE trailer.
// The host (scfq_gzdev.hpp: ingest_gz_device; batches of ~4096 segments, three in flight) validates the chain, re-decodes gaps (a false sync, the first block of a
// further member), and falls back to the host readers on anything it cannot prove consistent: results are gzread's
// byte stream or SCFQ_EGZ, never something in between.
#pragma once
#include "bgzf_inflate_kernel.hpp"

namespace scfq_dinflate {

constexpr uint32_t kGzWindow = 32768;

// ---- bit access for the search kernel: 64 bits starting at an arbitrary bit position (the buffer is padded) ----------
// (WP: a pointer to the words in global memory, or to the staged copy of a chunk in LDS)
template <typename WP>
__device__ __forceinline__ uint64_t bits64_at(WP words, uint64_t bit) {
  const uint64_t w = bit >> 6;
  const uint32_t sh = (uint32_t)bit & 63u;
  const uint64_t lo = words[w], hi = words[w + 1];
  return sh ? (lo >> sh) | (hi << (64u - sh)) : lo;
}

constexpr uint32_t kSyncThreads = 256;
constexpr uint32_t kSyncChunkBits = 65536;      // positions examined between two looks at the survivor list (r5: twice r2's — see sync_deep_tab)
constexpr uint32_t kSyncListCap = 512;          // (0.09 % of the positions get that far: ~60 per chunk)
constexpr uint32_t kSyncDeepSlots = 64;         // candidates of one round of the deep test: one per lane of the workgroup's first wave, 128 bytes of LDS each

// A dynamic block's header at position p: BFINAL, BTYPE = 10, HLIT <= 29, HDIST <= 29, HCLEN, then HCLEN + 4 three-bit code
// lengths that must form a complete code.  The cheap tests for 32 consecutive positions at once (bit i of the result: position
// bit + i may be a header): BTYPE, HLIT and HDIST are fixed bit patterns, so they are shifts and ANDs of the 64 bits at `bit` (the fields of position
// bit + 31 end at bit + 47).  About a fifth of the positions survive; only those pay for the Kraft sum.
template <typename WP>
__device__ __forceinline__ uint32_t sync_fields32(WP words, uint64_t bit) {
  const uint64_t x = bits64_at(words, bit);
  uint64_t m = (~x >> 1) & (x >> 2);                                     // BTYPE 10 (bit + 1 clear, bit + 2 set)
  m &= ~((x >> 4) & (x >> 5) & (x >> 6) & (x >> 7));                     // HLIT <= 29: not 1111x
  m &= ~((x >> 9) & (x >> 10) & (x >> 11) & (x >> 12));                  // HDIST <= 29
  return (uint32_t)m;
}
// the code-length code of a position that passed sync_fields32: complete (zlib rejects anything else for this code).
// The Kraft sum of the HCLEN + 4 three-bit lengths WITHOUT a loop (r5): the 61 bits behind position + 13 come in one fetch; b0 / b1 / b2 hold
// bit 0 / 1 / 2 of every length at the field's first bit, the seven non-zero values are seven AND patterns of them, and a length v weighs
// 128 >> v: seven population counts.  The loop it replaces ran HCLEN + 4 = 4 .. 19 times per candidate, each lane of a wave with a
// different count — a fifth of all bit positions get here, and this sum was two thirds of the search's time.
template <typename WP>
__device__ __forceinline__ bool sync_kraft(WP words, uint64_t bit) {
  const uint64_t w = bits64_at(words, bit + 13);
  const uint32_t hclen = ((uint32_t)w & 15u) + 4u;
  const uint64_t fm = 0x1249249249249249ull & ((1ull << (3u * hclen)) - 1ull);       // first bit of each of the HCLEN + 4 fields (3 x 19 = 57 bits at most)
  const uint64_t w2 = w >> 4;
  const uint64_t b0 = w2 & fm, b1 = (w2 >> 1) & fm, b2 = (w2 >> 2) & fm;
  const uint64_t n0 = ~b0, n1 = ~b1, n2 = ~b2;
  const uint32_t kraft = 64u * (uint32_t)__popcll(b0 & n1 & n2) + 32u * (uint32_t)__popcll(n0 & b1 & n2) + 16u * (uint32_t)__popcll(b0 & b1 & n2) +
                         8u * (uint32_t)__popcll(n0 & n1 & b2) + 4u * (uint32_t)__popcll(b0 & n1 & b2) + 2u * (uint32_t)__popcll(n0 & b1 & b2) +
                         (uint32_t)__popcll(b0 & b1 & b2);
  return kraft == 128u;
}

// the rest of the header: decode the HLIT + HDIST code lengths with the code-length code, ask for a complete literal/length
// code with an end-of-block code and a distance code zlib accepts.  Everything lives in registers (counts and the symbols
// sorted by code length are packed into 64-bit words), and a candidate is dropped the moment one of its codes is
// over-subscribed: random bits get there within a few dozen lengths, so the survivors of the quick tests cost little.
// lit_mask: bit j set = the file is known to hold no byte in [32 j, 32 j + 32) (from the host's sample of its first 192 KiB): a
// candidate whose literal/length code gives one of those literals a code is not a block of this file.  A header made of chance bits
// hands out code lengths all over the alphabet — for a text file (no byte >= 128: half the literals) this takes the false syncs from
// one in 4000 searches to practically none, and with them the extra decode round each of them cost (5 - 9 ms).  A true block that
// does use such a byte is simply not found here: the walk decodes it as a gap, results do not depend on the mask.
template <typename WP>
__device__ inline bool sync_deep(WP words, uint64_t bit, uint64_t end_bit, uint32_t lit_mask) {
  const uint64_t w = bits64_at(words, bit);
  const uint32_t hlit = (uint32_t)((w >> 3) & 31u) + 257u, hdist = (uint32_t)((w >> 8) & 31u) + 1u, hclen = (uint32_t)((w >> 13) & 15u) + 4u;
  const uint64_t w2 = bits64_at(words, bit + 17);
  uint64_t pl = 0;                                         // code length of code-length symbol s in bits [3s, 3s+3)
  for (uint32_t k = 0; k < hclen; ++k) pl |= ((w2 >> (3u * k)) & 7ull) << (3u * kClOrder[k]);
  uint64_t cnt8 = 0;                                       // number of symbols with code length l in bits [8l, 8l+8)
  for (uint32_t s = 0; s < 19; ++s) cnt8 += 1ull << (8u * (uint32_t)((pl >> (3u * s)) & 7u));
  // symbols sorted by (code length, symbol): 19 x 5 bits in two words (12 per word)
  uint64_t srt0 = 0, srt1 = 0;
  {
    uint32_t at = 0;
    for (uint32_t len = 1; len <= 7; ++len)
      for (uint32_t s = 0; s < 19; ++s)
        if (((pl >> (3u * s)) & 7u) == len) { if (at < 12) srt0 |= (uint64_t)s << (5u * at); else srt1 |= (uint64_t)s << (5u * (at - 12)); ++at; }
  }
  uint64_t pos = bit + 17 + 3u * hclen;
  const uint32_t total = hlit + hdist;
  uint32_t k = 0, prev = 0, kraft_l = 0, kraft_d = 0, n_d = 0, max_d = 0, len256 = 0;
  while (k < total) {
    if (pos + 64 > end_bit) return false;
    uint64_t b = bits64_at(words, pos);
    // canonical decode, one bit at a time (codes are at most 7 bits)
    uint32_t code = 0, first = 0, index = 0, sym = 0xFFu, used = 0;
    for (uint32_t len = 1; len <= 7; ++len) {
      code |= (uint32_t)(b & 1u);
      b >>= 1;
      ++used;
      const uint32_t cnt = (uint32_t)(cnt8 >> (8u * len)) & 0xFFu;
      if (code - first < cnt) {
        const uint32_t at = index + (code - first);
        sym = at < 12 ? (uint32_t)(srt0 >> (5u * at)) & 31u : (uint32_t)(srt1 >> (5u * (at - 12))) & 31u;
        break;
      }
      index += cnt;
      first = (first + cnt) << 1;
      code <<= 1;
    }
    if (sym == 0xFFu) return false;
    uint32_t rep = 1, val = sym;
    if (sym == 16) { if (k == 0) return false; val = prev; rep = 3 + ((uint32_t)b & 3u); used += 2; }
    else if (sym == 17) { val = 0; rep = 3 + ((uint32_t)b & 7u); used += 3; }
    else if (sym == 18) { val = 0; rep = 11 + ((uint32_t)b & 127u); used += 7; }
    pos += used;
    if (k + rep > total) return false;
    if (val) {
      // the run [k, k + rep) may straddle the literal/length | distance border
      const uint32_t in_l = k >= hlit ? 0u : (k + rep <= hlit ? rep : hlit - k), in_d = rep - in_l;
      kraft_l += in_l * (32768u >> val);
      kraft_d += in_d * (32768u >> val);
      if (kraft_l > 32768u || kraft_d > 32768u) return false;          // over-subscribed: no need to read on
      if (k <= 256u && 256u < k + rep) len256 = val;
      if (lit_mask && k < 256u) {
        // literals [k, min(k + rep, 256)) get a code: none of them may lie in a masked 32-value class
        const uint32_t hi = (k + rep < 256u ? k + rep : 256u) - 1u, c0 = k >> 5, c1 = hi >> 5;
        const uint32_t span = (c1 >= 31u ? 0xFFFFFFFFu : ((2u << c1) - 1u)) & ~((1u << c0) - 1u);
        if (lit_mask & span) return false;
      }
      if (in_d) { n_d += in_d; max_d = val > max_d ? val : max_d; }
    }
    k += rep;
    prev = val;
  }
  if (len256 == 0 || kraft_l != 32768u) return false;      // (zlib also takes an incomplete literal/length code of one 1-bit code: never a real block)
  return kraft_d == 32768u || n_d == 0 || (n_d == 1 && max_d == 1);
}

// The deep test, r5 form.  Cycle stamps over the r2 kernel (make sprof; profiles/r05/gz_search_cycles.txt) put 87 % of a search's
// time HERE although only 30 of a chunk's 32768 positions get this far: a lane walks its candidate's code lengths one canonical bit
// at a time behind a 133-step sort of the 19 code-length symbols, seven lanes of a wave at work, the wave as slow as its slowest.
// Now: (1) the code-length code becomes a 128-entry byte table in LDS, indexed by the next 7 bits MSB-first — canonical codes of one
// symbol are then a contiguous, aligned run, written with at most four wide stores; (2) a code length costs one table read; (3) the
// stream comes through a 64-bit buffer refilled every few codes; (4) the candidates of a chunk twice as long go through ONE round on
// the lanes of the first wave (the other three wait at the barrier and cost nothing).  Same accept / reject decisions as sync_deep.
// inverse of kClOrder: the field (0 .. 18) of the HCLEN list that holds the length of code-length symbol s
constexpr uint8_t kClField[19] = {3, 17, 15, 13, 11, 9, 7, 5, 4, 6, 8, 10, 12, 14, 16, 18, 0, 1, 2};
template <typename WP>
__device__ inline bool sync_deep_tab(WP words, uint64_t bit, uint64_t end_bit, uint32_t lit_mask, uint8_t* tab /* 128 bytes of LDS, this lane's */) {
  const uint64_t w = bits64_at(words, bit);
  const uint32_t hlit = (uint32_t)((w >> 3) & 31u) + 257u, hdist = (uint32_t)((w >> 8) & 31u) + 1u, hclen = (uint32_t)((w >> 13) & 15u) + 4u;
  const uint64_t w2 = bits64_at(words, bit + 17) & ((1ull << (3u * hclen)) - 1ull);        // fields behind HCLEN + 4 read as length 0
  // symbols per code length (the population counts of sync_kraft), first code of every length
  uint32_t nc[8];
  {
    const uint64_t fm = 0x1249249249249249ull;
    const uint64_t b0 = w2 & fm, b1 = (w2 >> 1) & fm, b2 = (w2 >> 2) & fm, n0 = ~b0 & fm, n1 = ~b1 & fm, n2 = ~b2 & fm;
    const uint32_t c1 = (uint32_t)__popcll(b0 & n1 & n2), c2 = (uint32_t)__popcll(n0 & b1 & n2), c3 = (uint32_t)__popcll(b0 & b1 & n2), c4 = (uint32_t)__popcll(n0 & n1 & b2),
                   c5 = (uint32_t)__popcll(b0 & n1 & b2), c6 = (uint32_t)__popcll(n0 & b1 & b2);
    nc[1] = 0; nc[2] = c1 << 1; nc[3] = (nc[2] + c2) << 1; nc[4] = (nc[3] + c3) << 1; nc[5] = (nc[4] + c4) << 1; nc[6] = (nc[5] + c5) << 1; nc[7] = (nc[6] + c6) << 1;
  }
  // (the code is complete — sync_kraft — so the runs below tile the 128 entries exactly)
  uint64_t ncp = 0;                                        // next code of length l in bits [8 l, 8 l + 8)
#pragma unroll
  for (int l = 1; l <= 7; ++l) ncp |= (uint64_t)nc[l] << (8 * l);
#pragma unroll
  for (int sy = 0; sy < 19; ++sy) {
    const uint32_t l = (uint32_t)(w2 >> (3u * kClField[sy])) & 7u;
    if (l) {
      const uint32_t code = (uint32_t)(ncp >> (8u * l)) & 0xFFu;
      ncp += 1ull << (8u * l);
      const uint32_t run = 128u >> l, at = code << (7u - l);                   // entries [at, at + run), at a multiple of run
      const uint32_t e = ((uint32_t)sy << 3) | l, e4 = e * 0x01010101u;
      if (run >= 16u) {
        for (uint32_t o = 0; o < run; o += 16u) *reinterpret_cast<uint4*>(tab + at + o) = make_uint4(e4, e4, e4, e4);
      } else if (run == 8u) {
        *reinterpret_cast<uint2*>(tab + at) = make_uint2(e4, e4);
      } else if (run == 4u) {
        *reinterpret_cast<uint32_t*>(tab + at) = e4;
      } else if (run == 2u) {
        *reinterpret_cast<uint16_t*>(tab + at) = (uint16_t)e4;
      } else {
        tab[at] = (uint8_t)e;
      }
    }
  }
  uint64_t base = bit + 17 + 3u * hclen;                  // the stream: 64 bits at `base`, `used` of them taken
  if (base + 128 > end_bit) return false;
  uint64_t bb = bits64_at(words, base);
  uint32_t used = 0;
  const uint32_t total = hlit + hdist;
  uint32_t k = 0, prev = 0, kraft_l = 0, kraft_d = 0, n_d = 0, max_d = 0, len256 = 0;
  while (k < total) {
    if (used > 48u) {                                      // (a code and its extra bits take 14 bits at most)
      base += used; used = 0;
      if (base + 128 > end_bit) return false;
      bb = bits64_at(words, base);
    }
    const uint32_t x = (uint32_t)(bb >> used);
    const uint32_t e = tab[__builtin_bitreverse32(x) >> 25];
    const uint32_t cl = e & 7u, sym = e >> 3;
    const uint32_t xb = x >> cl;
    uint32_t rep = 1, val = sym, take = cl;
    if (sym == 16) { if (k == 0) return false; val = prev; rep = 3 + (xb & 3u); take += 2; }
    else if (sym == 17) { val = 0; rep = 3 + (xb & 7u); take += 3; }
    else if (sym == 18) { val = 0; rep = 11 + (xb & 127u); take += 7; }
    used += take;
    if (k + rep > total) return false;
    if (val) {
      // the run [k, k + rep) may straddle the literal/length | distance border
      const uint32_t in_l = k >= hlit ? 0u : (k + rep <= hlit ? rep : hlit - k), in_d = rep - in_l;
      kraft_l += in_l * (32768u >> val);
      kraft_d += in_d * (32768u >> val);
      if (kraft_l > 32768u || kraft_d > 32768u) return false;          // over-subscribed: no need to read on
      if (k <= 256u && 256u < k + rep) len256 = val;
      if (lit_mask && k < 256u) {
        // literals [k, min(k + rep, 256)) get a code: none of them may lie in a masked 32-value class
        const uint32_t hi = (k + rep < 256u ? k + rep : 256u) - 1u, c0 = k >> 5, c1 = hi >> 5;
        const uint32_t span = (c1 >= 31u ? 0xFFFFFFFFu : ((2u << c1) - 1u)) & ~((1u << c0) - 1u);
        if (lit_mask & span) return false;
      }
      if (in_d) { n_d += in_d; max_d = val > max_d ? val : max_d; }
    }
    k += rep;
    prev = val;
  }
  if (len256 == 0 || kraft_l != 32768u) return false;
  return kraft_d == 32768u || n_d == 0 || (n_d == 1 && max_d == 1);
}

// found[s] = first bit position >= from[s] (and < from[s] + max_bits, < end_bit) that passes; ~0 when none does.
// Every chunk of 32768 positions is first STAGED in LDS (4.7 KiB: the chunk and the longest block header behind its last
// position), one coalesced pass, and both tests read it from there: straight from global memory each of the 128 steps of the
// quick test and each code length of the deep one was a memory round trip (6.2 ms for 4096 searches over 0.5 GB).
#ifdef SCFQ_SPROF      // measurement builds only (make sprof): where the search's cycles go, summed over all workgroups (thread 0's clock)
__device__ unsigned long long g_sprof[8];      // stage, fields + Kraft, deep test, chunks, list entries, workgroups
#define SCFQ_SP_T(var_) const uint64_t var_ = __builtin_readcyclecounter()
#define SCFQ_SP_ADD(slot_, v_) do { if (threadIdx.x == 0) atomicAdd(&g_sprof[slot_], (unsigned long long)(v_)); } while (0)
#else
#define SCFQ_SP_T(var_) do {} while (0)
#define SCFQ_SP_ADD(slot_, v_) do {} while (0)
#endif
constexpr uint32_t kSyncStageWords = kSyncChunkBits / 64 + 80;       // the chunk's words + 1 (the chunk starts inside a word) + 4600 bits of header + 128 bits of look-ahead + spare
__global__ __launch_bounds__(kSyncThreads) void gz_sync_search(const uint64_t* __restrict__ words, uint64_t end_bit, const uint64_t* __restrict__ from,
                                                              uint32_t n_seg, uint64_t max_bits, uint64_t* __restrict__ found, uint32_t lit_mask) {
  __shared__ uint32_t list[kSyncListCap];
  __shared__ uint32_t n_list, best;
  __shared__ uint64_t stage[kSyncStageWords];
  __shared__ __attribute__((aligned(16))) uint8_t dtab[kSyncDeepSlots * 128];
  typedef __attribute__((address_space(3))) const uint64_t* lds_words;
  const lds_words sw = (lds_words)(uintptr_t)(uint32_t)(uintptr_t)stage;
  const uint32_t s = blockIdx.x;
  if (s >= n_seg) return;
  const uint64_t base = from[s];
  const uint64_t limit = (base + max_bits < end_bit) ? base + max_bits : end_bit;
  const uint64_t last_word = (end_bit >> 6) + 3;                         // (the buffer is padded well beyond that)
  uint64_t result = ~0ull;
  for (uint64_t c0 = base; c0 < limit; c0 += kSyncChunkBits) {
    const uint64_t w0 = c0 >> 6, bit0 = w0 << 6;                         // staged word k = words[w0 + k]; a position p is bit p - bit0 of the stage
    SCFQ_SP_T(sp0);
    if (threadIdx.x == 0) { n_list = 0; best = 0xFFFFFFFFu; }
    for (uint32_t k = threadIdx.x; k < kSyncStageWords; k += kSyncThreads) stage[k] = w0 + k <= last_word ? words[w0 + k] : 0ull;
    __syncthreads();
    SCFQ_SP_T(sp1);
    const uint64_t rel_end = end_bit - bit0 < (uint64_t)(kSyncStageWords - 2) * 64u ? end_bit - bit0 : (uint64_t)(kSyncStageWords - 2) * 64u;
    for (uint32_t j = threadIdx.x; j < kSyncChunkBits / 32u; j += kSyncThreads) {
      const uint64_t p0 = c0 + 32u * j;
      uint32_t cand = sync_fields32(sw, p0 - bit0);
      // (a position must leave 128 bits in front of the limit)
      if (p0 + 128u > limit) cand = 0;
      else if (p0 + 31u + 128u > limit) cand &= (2u << (uint32_t)(limit - 128u - p0)) - 1u;
      while (cand) {
        const uint32_t i = (uint32_t)__builtin_ctz(cand);
        cand &= cand - 1u;
        if (sync_kraft(sw, p0 + i - bit0)) {
          const uint32_t at = atomicAdd(&n_list, 1u);
          if (at < kSyncListCap) list[at] = 32u * j + i;
        }
      }
    }
    __syncthreads();
    SCFQ_SP_T(sp2);
    const uint32_t n = n_list < kSyncListCap ? n_list : kSyncListCap;
    // (a list that overflowed lost some LATER candidates of this chunk at worst out of order: every kept one is still
    // tested, the minimum over them is taken, and a missed earlier true block only makes this segment a gap for the host)
    // (the first wave only: one candidate per lane and round, its table in the lane's own 128 bytes; the other waves go straight to
    // the barrier — spread over all four, each wave ran the whole walk for a handful of lanes)
    if (threadIdx.x < kSyncDeepSlots)
      for (uint32_t t = threadIdx.x; t < n; t += kSyncDeepSlots)
        if (sync_deep_tab(sw, c0 + list[t] - bit0, rel_end, lit_mask, dtab + threadIdx.x * 128u)) atomicMin(&best, list[t]);
    __syncthreads();
    SCFQ_SP_T(sp3);
    SCFQ_SP_ADD(0, sp1 - sp0); SCFQ_SP_ADD(1, sp2 - sp1); SCFQ_SP_ADD(2, sp3 - sp2); SCFQ_SP_ADD(3, 1); SCFQ_SP_ADD(4, n);
    if (best != 0xFFFFFFFFu) { result = c0 + best; break; }       // block-uniform
    __syncthreads();
  }
  if (threadIdx.x == 0) found[s] = result;
}

// ---- G2 ----------------------------------------------------------------------------------------------------------------
struct GzSeg {
  uint64_t start_bit;       // first bit of a block header (relative to the compressed buffer)
  uint64_t stop_bit;        // decode until the first block boundary at or after this bit (or the member's final block)
  uint64_t sym_off;         // index, in the symbol pool, of the segment's first MARKER symbol (output follows the 32768 markers)
  uint32_t cap;             // output symbols the segment may produce
  uint32_t reserved;
};
struct GzSegOut {
  uint64_t end_bit;         // bit after the last block decoded
  uint32_t n_sym;           // symbols produced
  uint32_t status;          // kGzOk: at a block boundary; kGzMemberEnd: the final block was decoded; else an error
};
enum : uint32_t { kGzOk = 0, kGzMemberEnd = 1, kGzErrData = 2, kGzErrOverflow = 3 };

// A wave's LDS: bgzf_inflate's (the code-length table lies where the distance table is built afterwards: 8064 bytes per wave, twenty waves per CU).
constexpr int kGzWaveLdsBytes = kWaveLdsBytes;
__global__ __launch_bounds__(64 * kWavesPerWg) void gz_segment_decode(const uint8_t* __restrict__ comp, uint64_t comp_bytes, const GzSeg* __restrict__ segs,
                                                                     uint32_t n_seg, uint16_t* syms, GzSegOut* __restrict__ outs, uint32_t serial_loop) {
  extern __shared__ uint32_t lds[];
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t b = blockIdx.x * kWavesPerWg + wave;
  uint32_t* lit = lds + wave * (kGzWaveLdsBytes / 4);
  uint32_t* dist = lit + kLitEntries;
  uint32_t* cltab = dist;
  uint8_t* lens = reinterpret_cast<uint8_t*>(dist + kDistEntries);
  uint16_t* sorted = reinterpret_cast<uint16_t*>(lens + 320);
  uint16_t* count = sorted + 320;
  uint16_t* offs = count + 16;
  __shared__ uint32_t build_ok[kWavesPerWg];
  __shared__ uint32_t s_len[32], s_dist[32];
  if (threadIdx.x < 29) s_len[threadIdx.x] = ((uint32_t)kLenBase[threadIdx.x] << 16) | ((uint32_t)kLenExtra[threadIdx.x] << 4) | kVal;
  if (threadIdx.x < 30) s_dist[threadIdx.x] = ((uint32_t)kDistBase[threadIdx.x] << 16) | ((uint32_t)kDistExtra[threadIdx.x] << 4) | kVal;
  __syncthreads();
  if (b >= n_seg) return;

  const GzSeg sg = segs[b];
  const uint64_t seg_byte = sg.start_bit >> 3;                       // the read position `ip` counts bytes from here
  const uint64_t left = comp_bytes - seg_byte;
  // (a segment reads at most 1 GiB of input — the symbol loop's bounds tests use sign-bit arithmetic; one that would need more
  // ends with a data error and the file goes to the host path)
  const uint32_t ip_end = left < 0x3FFFFFFFull ? (uint32_t)left : 0x3FFFFFFFu;
  const uint64_t stop64 = sg.stop_bit > seg_byte * 8 ? sg.stop_bit - seg_byte * 8 : 0;
  const uint32_t stop_rel = stop64 < 0xF0000000ull ? (uint32_t)stop64 : 0xF0000000u;     // bit position relative to seg_byte
  uint32_t ip = 0;
  uint16_t* const o = syms + sg.sym_off;
  const uint32_t cap_total = kGzWindow + sg.cap;                      // symbols, markers included
  // the window this segment does not have: 32768 markers in front of its output
  for (uint32_t k = lane; k < kGzWindow / 2; k += 64)
    reinterpret_cast<uint32_t*>(o)[k] = (0x8000u | (2u * k)) | ((0x8000u | (2u * k + 1u)) << 16);
  uint64_t bb = 0;
  uint32_t bc = 0;
  uint32_t pos = kGzWindow, err = kGzOk;
  bool last = false;

  typedef uint32_t dword4 __attribute__((ext_vector_type(4)));
  typedef const dword4 __attribute__((address_space(4), aligned(4))) const_dword4;
  typedef const uint8_t __attribute__((address_space(4))) const_byte;
  const_byte* const in4 = (const_byte*)(uintptr_t)(comp + (seg_byte & ~3ull));
  const uint8_t* const in = comp + seg_byte;
  const uint32_t in_off = (uint32_t)(seg_byte & 3ull);
  dword4 pf;
  uint32_t pf_sh;
#define SCFQ_GPREFETCH()                                                                       \
  do {                                                                                         \
    const uint32_t a_ = in_off + (ip < ip_end ? ip : ip_end);                                  \
    pf_sh = (a_ & 3u) << 3;                                                                    \
    pf = *(const_dword4*)(in4 + (a_ & ~3u));                                                   \
  } while (0)
#define SCFQ_GREFILL()                                                                         \
  do {                                                                                         \
    const uint32_t up_ = 31u - pf_sh;                                                          \
    const uint32_t lo_ = (pf.x >> pf_sh) | ((pf.y << 1) << up_), hi_ = (pf.y >> pf_sh) | ((pf.z << 1) << up_); \
    bb |= (((uint64_t)hi_ << 32) | lo_) << bc;                                                 \
    ip += (63u - bc) >> 3;                                                                     \
    bc |= 56u;                                                                                 \
    SCFQ_GPREFETCH();                                                                          \
  } while (0)
#define SCFQ_GTAKE(n_) do { bb >>= (n_); bc -= (n_); } while (0)
  SCFQ_GPREFETCH();
  SCFQ_GREFILL();
  SCFQ_GTAKE((uint32_t)(sg.start_bit & 7ull));

  uint32_t guard = 0;
  while (!last && err == kGzOk) {
    if (ip * 8u - bc >= stop_rel) break;              // at a block boundary at or after the next segment's start
    SCFQ_GREFILL();
    if (ip > ip_end + 16) { err = kGzErrData; break; }
    last = bb & 1;
    const uint32_t type = (uint32_t)(bb >> 1) & 3u;
    SCFQ_GTAKE(3);
    if (type == 0) {
      SCFQ_GTAKE(bc & 7);
      const uint32_t len = (uint32_t)bb & 0xFFFF, nlen = (uint32_t)(bb >> 16) & 0xFFFF;
      SCFQ_GTAKE(32);
      if ((len ^ nlen) != 0xFFFF) { err = kGzErrData; break; }
      ip -= bc >> 3;
      bb = 0; bc = 0;
      if (ip > ip_end || ip_end - ip < len) { err = kGzErrData; break; }
      if (pos + len > cap_total) { err = kGzErrOverflow; break; }
      for (uint32_t k = lane; k < len; k += 64) o[pos + k] = (uint16_t)in[ip + k];
      ip += len; pos += len;
      SCFQ_GPREFETCH();
      continue;
    }
    if (type == 3) { err = kGzErrData; break; }
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
      SCFQ_GTAKE(14);
      if (hlit > 286 || hdist > 30) { err = kGzErrData; break; }
      if (lane == 0) for (int k = 0; k < 19; ++k) lens[k] = 0;
      for (uint32_t k = 0; k < hclen; ++k) {
        if ((k & 7) == 0) SCFQ_GREFILL();
        if (lane == 0) lens[kClOrder[k]] = (uint8_t)(bb & 7);
        SCFQ_GTAKE(3);
      }
      if (lane == 0) build_ok[wave] = build_table(lens, 19, kKindCodeLen, cltab, 7, 128, sorted, count, offs, s_len, s_dist) ? 1u : 0u;
      __builtin_amdgcn_wave_barrier();
      if (!__builtin_amdgcn_readfirstlane((int)build_ok[wave])) { err = kGzErrData; break; }
      uint32_t k = 0, prev = 0;
      const uint32_t total = hlit + hdist;
      while (k < total) {
        if (++guard > (1u << 28)) { err = kGzErrData; break; }
        SCFQ_GREFILL();
        const uint32_t e = uni(cltab[bb & 127]);
        if (!(e & kVal)) { err = kGzErrData; break; }
        SCFQ_GTAKE(e & 15);
        const uint32_t sym = e >> 16;
        uint32_t rep = 1, val = sym;
        if (sym == 16) { if (k == 0) { err = kGzErrData; break; } val = prev; rep = 3 + ((uint32_t)bb & 3); SCFQ_GTAKE(2); }
        else if (sym == 17) { val = 0; rep = 3 + ((uint32_t)bb & 7); SCFQ_GTAKE(3); }
        else if (sym == 18) { val = 0; rep = 11 + ((uint32_t)bb & 127); SCFQ_GTAKE(7); }
        if (k + rep > total) { err = kGzErrData; break; }
        if (lane == 0) for (uint32_t r = 0; r < rep; ++r) lens[k + r] = (uint8_t)val;
        k += rep;
        prev = val;
      }
      if (err) break;
      if (lane == 0) {
        bool ok = lens[256] != 0;
        ok = ok && build_table(lens + hlit, (int)hdist, kKindDist, dist, kDistRoot, kDistEntries, sorted, count, offs, s_len, s_dist);
        ok = ok && build_table(lens, (int)hlit, kKindLitLen, lit, kLitRoot, kLitEntries, sorted, count, offs, s_len, s_dist);
        build_ok[wave] = ok ? 1u : 0u;
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (!__builtin_amdgcn_readfirstlane((int)build_ok[wave])) { err = kGzErrData; break; }
    // ---- symbols: symbol_loop of bgzf_inflate_kernel.hpp, 16 bits per output symbol ------------------------------------
    {
      SymState sst;
      sst.bb = bb; sst.bc = bc; sst.ip = ip; sst.pos = pos; sst.err = kOk;
      sst.pf[0] = pf.x; sst.pf[1] = pf.y; sst.pf[2] = pf.z; sst.pf[3] = pf.w; sst.pf_sh = pf_sh;
      if (serial_loop == 1u) symbol_loop<true>(&sst, comp + (seg_byte & ~3ull), in_off, ip_end, o, cap_total, (uint32_t)(uintptr_t)lit, (uint32_t)(uintptr_t)dist);
      else if (serial_loop == 2u) symbol_loop_dense<true>(&sst, comp + (seg_byte & ~3ull), in_off, ip_end, o, cap_total, (uint32_t)(uintptr_t)lit, (uint32_t)(uintptr_t)dist, (uint32_t)(uintptr_t)lens);
      else symbol_loop_lanes<true>(&sst, comp + (seg_byte & ~3ull), in_off, ip_end, o, cap_total, (uint32_t)(uintptr_t)lit, (uint32_t)(uintptr_t)dist, (uint32_t)(uintptr_t)lens);
      bb = ((uint64_t)uni((uint32_t)(sst.bb >> 32)) << 32) | uni((uint32_t)sst.bb);
      bc = uni(sst.bc); ip = uni(sst.ip); pos = uni(sst.pos);
      const uint32_t e2 = uni(sst.err);
      pf.x = uni(sst.pf[0]); pf.y = uni(sst.pf[1]); pf.z = uni(sst.pf[2]); pf.w = uni(sst.pf[3]); pf_sh = uni(sst.pf_sh);
      if (e2) err = kGzErrData;
    }
    if (pos > cap_total && err == kGzOk) err = kGzErrOverflow;
    if (ip > ip_end + 16 && err == kGzOk) err = kGzErrData;
  }
  if (err == kGzOk && (uint64_t)ip * 8 - bc > (uint64_t)ip_end * 8) err = kGzErrData;     // bits taken beyond the data
#undef SCFQ_GREFILL
#undef SCFQ_GPREFETCH
#undef SCFQ_GTAKE
  if (lane == 0) {
    GzSegOut r;
    r.end_bit = seg_byte * 8 + ((uint64_t)ip * 8 - bc);
    r.n_sym = pos - kGzWindow;
    r.status = err ? err : (last ? kGzMemberEnd : kGzOk);
    outs[b] = r;
  }
}

// ---- G3 ----------------------------------------------------------------------------------------------------------------
struct GzChain {           // one entry per segment of the validated chain, in stream order
  uint64_t sym_off;        // as in GzSeg
  uint64_t out_off;        // offset of the segment's first byte in the inflated stream
  uint32_t n_sym;
  uint32_t valid_before;   // bytes of real history in front of the segment (0 at a member's start, 32768 once a member is that long)
  uint32_t chain_id;       // segments of one member form one chain (a member starts with an empty window)
  uint32_t reserved;
};

// windows[k] = the 32 KiB in front of segment k (only its last valid_before bytes mean anything).  One workgroup per
// member (blockIdx.x = chain id, the entries of a chain are contiguous: [first[c], first[c + 1])).
// win_init (optional): the window in front of the FIRST chain's first segment (a member that began in an earlier batch);
// win_final (optional): receives the window behind the LAST chain's last segment (a member that goes on in the next batch);
// win_each (optional): chain c starts from win_each[c] (the third step of the grouped form below).
//
// A chain of thousands of segments is not walked in one go: what a run of segments does to the window is a MAP (byte i of
// the window behind the run = a literal, or byte j of the window in front of it), maps compose, and composition is
// associative.  So the host cuts every chain into groups of <= 64 entries and launches
//   1. gz_window_maps   one workgroup per group: the group's map, from the identity, entry by entry (16-bit symbols in LDS)
//   2. gz_window_chain  over the MAPS of a chain (a map has the form of a segment of 32768 symbols): the window in front of every group
//   3. gz_window_chain  one workgroup per group, started from its window (win_each): the window in front of every entry
// — three short sequential walks (<= 64, n / 64, <= 64 steps) instead of one of n steps.
__global__ __launch_bounds__(1024) void gz_window_chain(const GzChain* __restrict__ chain, const uint32_t* __restrict__ first, const uint16_t* __restrict__ syms,
                                                       uint8_t* __restrict__ windows, const uint8_t* __restrict__ win_init, uint8_t* __restrict__ win_final,
                                                       const uint8_t* __restrict__ win_each) {
  __shared__ __attribute__((aligned(16))) uint8_t W[2][kGzWindow];
  // the chain entries of the next 1024 segments, so that the loop never waits for a dependent global load: [k & 1023]
  __shared__ uint64_t m_off[1024];
  __shared__ uint32_t m_n[1024];
  const uint32_t tid = threadIdx.x;
  const uint32_t k0 = first[blockIdx.x], k1 = first[blockIdx.x + 1];
  const bool carry_out = win_final != nullptr && blockIdx.x + 1 == gridDim.x;
  const uint8_t* w0 = win_each ? win_each + (uint64_t)blockIdx.x * kGzWindow : ((win_init != nullptr && blockIdx.x == 0) ? win_init : nullptr);
  for (uint32_t i = tid; i < kGzWindow / 4; i += 1024)
    reinterpret_cast<uint32_t*>(W[0])[i] = w0 ? reinterpret_cast<const uint32_t*>(w0)[i] : 0u;
  auto load_meta = [&](uint32_t base) {      // entries [base, base + 1024) into slots [0, 1024)
    const uint32_t k = base + tid;
    if (k < k1) { m_off[tid] = chain[k].sym_off; m_n[tid] = chain[k].n_sym; }
  };
  load_meta(k0);
  __syncthreads();
  // The usual case — a segment of at least 32768 symbols: the new window is its last 32768 symbols.  Thread t owns the symbol
  // pairs t, t + 1024, ... (sixteen 4-byte loads in flight at once, each wave instruction 256 contiguous bytes: with 32
  // consecutive symbols per thread every load instruction touched 64 cache lines and the step took 4.8 us instead of ~2).
  // The loads of segment k + 1 are issued before segment k is resolved, so the chain pays LDS time per segment, not a memory
  // round trip.
  uint32_t w[16], wn[16];
  auto fetch = [&](uint32_t k, uint32_t* dst) {
    const uint32_t n_k = m_n[(k - k0) & 1023u];
    if (n_k < kGzWindow) return;
    const uint16_t* src = syms + m_off[(k - k0) & 1023u] + kGzWindow + (n_k - kGzWindow);
#pragma unroll
    for (int q = 0; q < 16; ++q) __builtin_memcpy(&dst[q], src + 2u * (tid + 1024u * q), 4);     // (the run starts at any symbol: 2-byte aligned)
  };
  if (k0 < k1) fetch(k0, w);
  uint32_t cur = 0;
  for (uint32_t k = k0; k < k1; ++k) {
    const uint32_t n = m_n[(k - k0) & 1023u];
    const uint64_t sym_off = m_off[(k - k0) & 1023u];
    uint8_t* wout = windows + (uint64_t)k * kGzWindow;
    for (uint32_t i = tid; i < kGzWindow / 16; i += 1024)
      reinterpret_cast<uint4*>(wout)[i] = reinterpret_cast<const uint4*>(W[cur])[i];
    if (k + 1 == k1 && !carry_out) break;                    // nobody needs the window after the chain's last segment
    if (k + 1 < k1 && ((k + 1 - k0) & 1023u) == 0) {         // the next 1024 entries (everyone has read entry k by now)
      __syncthreads();
      load_meta(k + 1);
      __syncthreads();
    }
    if (k + 2 < k1 || (k + 1 < k1 && carry_out)) fetch(k + 1, wn);      // (entry k + 1 is resolved when it is not the last, or the window goes on)
    if (n >= kGzWindow) {
      uint16_t* dst = reinterpret_cast<uint16_t*>(W[cur ^ 1]);
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const uint32_t s0 = w[q] & 0xFFFFu, s1 = w[q] >> 16;
        const uint32_t v0 = (s0 & 0x8000u) ? (uint32_t)W[cur][s0 & 0x7FFFu] : (s0 & 0xFFu);
        const uint32_t v1 = (s1 & 0x8000u) ? (uint32_t)W[cur][s1 & 0x7FFFu] : (s1 & 0xFFu);
        dst[tid + 1024u * q] = (uint16_t)(v0 | (v1 << 8));
      }
    } else {
      const uint16_t* s = syms + sym_off + kGzWindow;        // the segment's output symbols
      for (uint32_t i = tid; i < kGzWindow; i += 1024) {
        uint8_t v;
        if (i >= kGzWindow - n) {
          const uint16_t sy = s[n - kGzWindow + i];          // (n + i - 32768 >= 0 here)
          v = (sy & 0x8000u) ? W[cur][sy & 0x7FFFu] : (uint8_t)sy;
        } else {
          v = W[cur][i + n];                                 // a short segment: the window slides by n
        }
        W[cur ^ 1][i] = v;
      }
    }
    __syncthreads();
    cur ^= 1;
#pragma unroll
    for (int q = 0; q < 16; ++q) w[q] = wn[q];
  }
  if (carry_out && k0 < k1) {
    for (uint32_t i = tid; i < kGzWindow / 16; i += 1024)
      reinterpret_cast<uint4*>(win_final)[i] = reinterpret_cast<const uint4*>(W[cur])[i];
  }
}

// step 1 of the grouped form: maps[(g + 1) * 32768 ..] = the map of group g = entries [gfirst[g], gfirst[g + 1]) (at most 64),
// as 32768 symbols: a literal, or 0x8000 | j = byte j of the window in front of the group
__global__ __launch_bounds__(1024) void gz_window_maps(const GzChain* __restrict__ chain, const uint32_t* __restrict__ gfirst, const uint16_t* __restrict__ syms,
                                                      uint16_t* __restrict__ maps) {
  __shared__ __attribute__((aligned(16))) uint16_t W[kGzWindow];
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t k0 = gfirst[blockIdx.x], k1 = gfirst[blockIdx.x + 1];
  // the group's entries live in the lanes of every wave (entry j in lane j): no dependent global load inside the loop
  uint32_t my_n = 0, my_lo = 0, my_hi = 0;
  if (k0 + lane < k1) { const uint64_t o = chain[k0 + lane].sym_off; my_n = chain[k0 + lane].n_sym; my_lo = (uint32_t)o; my_hi = (uint32_t)(o >> 32); }
  for (uint32_t q = 0; q < 16; ++q) {
    const uint32_t i = 2u * (tid + 1024u * q);
    reinterpret_cast<uint32_t*>(W)[tid + 1024u * q] = (0x8000u | i) | ((0x8000u | (i + 1u)) << 16);
  }
  __syncthreads();
  uint32_t w[16], wn[16], r[16];
  auto entry = [&](uint32_t j, uint32_t* n, uint64_t* off) {
    *n = (uint32_t)__shfl((int)my_n, (int)j);
    *off = (uint64_t)(uint32_t)__shfl((int)my_lo, (int)j) | ((uint64_t)(uint32_t)__shfl((int)my_hi, (int)j) << 32);
  };
  auto fetch = [&](uint32_t j, uint32_t* dst) {
    uint32_t n_k; uint64_t off;
    entry(j, &n_k, &off);
    if (n_k < kGzWindow) return;
    const uint16_t* src = syms + off + kGzWindow + (n_k - kGzWindow);
#pragma unroll
    for (int q = 0; q < 16; ++q) __builtin_memcpy(&dst[q], src + 2u * (tid + 1024u * q), 4);
  };
  const uint32_t cnt = k1 - k0;
  if (cnt) fetch(0, w);
  for (uint32_t j = 0; j < cnt; ++j) {
    uint32_t n; uint64_t off;
    entry(j, &n, &off);
    if (j + 1 < cnt) fetch(j + 1, wn);
    if (n >= kGzWindow) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const uint32_t s0 = w[q] & 0xFFFFu, s1 = w[q] >> 16;
        const uint32_t v0 = (s0 & 0x8000u) ? (uint32_t)W[s0 & 0x7FFFu] : (s0 & 0xFFu);
        const uint32_t v1 = (s1 & 0x8000u) ? (uint32_t)W[s1 & 0x7FFFu] : (s1 & 0xFFu);
        r[q] = v0 | (v1 << 16);
      }
    } else {
      const uint16_t* sg = syms + off + kGzWindow;           // a short segment: the window slides by n
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        uint32_t v[2];
        for (uint32_t h = 0; h < 2; ++h) {
          const uint32_t i = 2u * (tid + 1024u * q) + h;
          if (i >= kGzWindow - n) {
            const uint32_t sy = sg[n - kGzWindow + i];
            v[h] = (sy & 0x8000u) ? (uint32_t)W[sy & 0x7FFFu] : (sy & 0xFFu);
          } else {
            v[h] = W[i + n];
          }
        }
        r[q] = v[0] | (v[1] << 16);
      }
    }
    __syncthreads();                                         // everyone has read the old window
#pragma unroll
    for (int q = 0; q < 16; ++q) reinterpret_cast<uint32_t*>(W)[tid + 1024u * q] = r[q];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q) w[q] = wn[q];
  }
  uint16_t* out = maps + (uint64_t)(blockIdx.x + 1) * kGzWindow;
  for (uint32_t i = tid; i < kGzWindow / 8; i += 1024) reinterpret_cast<uint4*>(out)[i] = reinterpret_cast<const uint4*>(W)[i];
}

// The maps of consecutive groups folded into ONE: run <- M_{n-1} o ... o M_0 o run (maps[(g + 1) * 32768 ..] = M_g, as gz_window_maps
// writes them; run: 32768 symbols, the identity 0x8000 | i to begin with).  What a whole STRETCH of a member does to the window —
// the thing the ranks of a sharded count exchange (scfq_gzdev.hpp: GzStretch).  One workgroup; the running map goes back and forth
// between two 64 KiB arrays in global memory (the L2 holds them; no LDS: this kernel must not wait for the decode kernels' LDS).
__global__ __launch_bounds__(1024) void gz_map_fold(const uint16_t* __restrict__ maps, uint32_t n_groups, uint16_t* run_a, uint16_t* run_b) {
  const uint32_t tid = threadIdx.x;
  uint16_t* cur = run_a;
  uint16_t* nxt = run_b;
  for (uint32_t g = 0; g < n_groups; ++g) {
    const uint16_t* m = maps + (uint64_t)(g + 1) * kGzWindow;
    for (uint32_t i = tid; i < kGzWindow; i += 1024) {
      const uint32_t sy = m[i];
      // (agent-scope load: served by the L2 — the array was written by other waves of this workgroup one step ago, and this CU's L1
      // may still hold the line from two steps ago)
      nxt[i] = (sy & 0x8000u) ? __hip_atomic_load(&cur[sy & 0x7FFFu], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (uint16_t)sy;
    }
    __threadfence();
    __syncthreads();
    uint16_t* t = cur; cur = nxt; nxt = t;
  }
  // the result is wanted in run_a
  if (cur != run_a) {
    for (uint32_t i = tid; i < kGzWindow; i += 1024) run_a[i] = __hip_atomic_load(&cur[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// The output symbols of proven chain entries, copied out of a batch's symbol pool into a store that outlives the batch (GzStretch with an
// exchange: the symbols wait for the window the other ranks' maps give, instead of being decoded a second time).  tab[e] = {source offset
// (the entry's first OUTPUT symbol, behind its markers), destination offset, symbols}; offsets in symbols from the two bases, multiples of 8.
__global__ __launch_bounds__(256) void gz_pack_symbols(const uint16_t* __restrict__ src_base, uint16_t* __restrict__ dst_base, const uint64_t* __restrict__ tab) {
  const uint64_t so = tab[3ull * blockIdx.x], dof = tab[3ull * blockIdx.x + 1], n = tab[3ull * blockIdx.x + 2];
  const uint4* s = reinterpret_cast<const uint4*>(src_base + so);
  uint4* d = reinterpret_cast<uint4*>(dst_base + dof);
  const uint64_t n8 = (n + 7) / 8;           // (the room of both is a multiple of 8 symbols)
  for (uint64_t i = threadIdx.x; i < n8; i += 256) d[i] = s[i];
}

// ---- G4 ----------------------------------------------------------------------------------------------------------------
constexpr uint32_t kResolveTile = 1u << 17;                  // symbols per workgroup: the 32 KiB window load is a quarter of the tile's traffic
// work item w: chain entry entry[w], symbols [tile[w] * kResolveTile, ...) of it
__global__ __launch_bounds__(256) void gz_resolve(const GzChain* __restrict__ chain, const uint32_t* __restrict__ entry, const uint32_t* __restrict__ tile,
                                                  const uint16_t* __restrict__ syms, const uint8_t* __restrict__ windows, uint8_t* __restrict__ out,
                                                  uint32_t* __restrict__ status) {
  __shared__ __attribute__((aligned(16))) uint8_t W[kGzWindow];
  const uint32_t k = entry[blockIdx.x];
  const GzChain c = chain[k];
  const uint8_t* win = windows + (uint64_t)k * kGzWindow;
  for (uint32_t i = threadIdx.x; i < kGzWindow / 16; i += 256) reinterpret_cast<uint4*>(W)[i] = reinterpret_cast<const uint4*>(win)[i];
  __syncthreads();
  const uint32_t lo = tile[blockIdx.x] * kResolveTile;
  const uint32_t hi = (lo + kResolveTile < c.n_sym) ? lo + kResolveTile : c.n_sym;
  const uint16_t* s = syms + c.sym_off + kGzWindow;
  uint8_t* o = out + c.out_off;
  const uint32_t missing = kGzWindow - c.valid_before;       // window slots [0, missing) do not exist
  uint32_t bad = 0;
  // 8 symbols (16 bytes in, 8 bytes out) per lane and step; the ends of the tile symbol by symbol (out_off has any alignment)
  uint32_t i = lo + threadIdx.x * 8u;
  for (; i + 8u <= hi; i += 256u * 8u) {
    uint4 q;
    __builtin_memcpy(&q, s + i, 16);                         // sym_off is a multiple of 8 symbols
    const uint32_t w4[4] = {q.x, q.y, q.z, q.w};
    uint32_t b8[2] = {0, 0};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t sy = (w4[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu;
      uint32_t v = sy & 0xFFu;
      if (sy & 0x8000u) { const uint32_t idx = sy & 0x7FFFu; bad |= (idx < missing) ? 1u : 0u; v = W[idx]; }
      b8[j >> 2] |= v << ((j & 3) * 8);
    }
    __builtin_memcpy(o + i, b8, 8);
  }
  // the tile's ragged end (fewer than 8 symbols per lane left): one symbol per lane and step
  const uint32_t full_end = lo + ((hi - lo) / 8u) * 8u;
  for (uint32_t j = full_end + threadIdx.x; j < hi; j += 256u) {
    const uint32_t sy = s[j];
    uint32_t v = sy & 0xFFu;
    if (sy & 0x8000u) { const uint32_t idx = sy & 0x7FFFu; bad |= (idx < missing) ? 1u : 0u; v = W[idx]; }
    o[j] = (uint8_t)v;
  }
  if (bad) atomicOr(status, 1u << kGzErrData);
}

// ---- G5 ----------------------------------------------------------------------------------------------------------------
// Raw CRC-32 (zero initial value, no final inversion: R(M) = M(x) x^32 mod P, reflected) of the kCrcTile-byte tiles of a
// virtual message "pad zero bytes, then data[0 .. n)" whose length is a multiple of kCrcTile.  Leading zeros do not change R,
// R(A || B) = R(A) x^(8|B|) + R(B), and crc32(M) = R(M) ^ (0xFFFFFFFF x^(8|M|)) ^ 0xFFFFFFFF: the host folds the tiles
// (Horner with x^(8 kCrcTile)) and conditions the result.  Inside a tile thread t owns the 64-byte pieces t, t + 256, ...:
// its pieces are folded with x^(8 * 16384) per step, the 256 threads are then shifted to the end of the tile and xor-ed.
constexpr uint32_t kCrcTile = 1u << 20, kCrcStep = 256u * 64u;
__global__ __launch_bounds__(256) void gz_crc32_tiles(const uint8_t* __restrict__ data, uint64_t n, uint64_t pad, uint32_t* __restrict__ tile_crc) {
  __shared__ uint32_t red[256];
  __shared__ uint32_t tab[4][256];               // four bytes per step: tab[0] the byte-wise table, tab[k][i] the CRC of byte i followed by k zero bytes
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
  const uint64_t tile0 = (uint64_t)blockIdx.x * kCrcTile;    // virtual offset
  const uint32_t x_step = x_pow_8n(kCrcStep);
  uint32_t acc = 0;
  for (uint32_t st = 0; st < kCrcTile / kCrcStep; ++st) {
    const uint64_t v = tile0 + (uint64_t)st * kCrcStep + (uint64_t)t * 64u;     // virtual offset of this thread's piece
    uint32_t c = 0;
    if (v + 64 <= pad) {
      // all zeros: contributes nothing
    } else if (v >= pad && v + 64 <= pad + n) {
      uint4 q[4];
      __builtin_memcpy(q, data + (v - pad), 64);             // any alignment
      const uint32_t w[16] = {q[0].x, q[0].y, q[0].z, q[0].w, q[1].x, q[1].y, q[1].z, q[1].w, q[2].x, q[2].y, q[2].z, q[2].w, q[3].x, q[3].y, q[3].z, q[3].w};
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        c ^= w[j];
        c = tab[3][c & 255u] ^ tab[2][(c >> 8) & 255u] ^ tab[1][(c >> 16) & 255u] ^ tab[0][c >> 24];
      }
    } else {
      for (uint32_t j = 0; j < 64; ++j) {
        const uint64_t p = v + j;
        const uint32_t byte = (p >= pad && p < pad + n) ? data[p - pad] : 0u;
        c = crc_byte(c, byte);
      }
    }
    acc = gf2_mulmod(x_step, acc) ^ c;
  }
  // thread t's pieces end 64 * (255 - t) bytes before the end of every step
  red[t] = gf2_mulmod(x_pow_8n(64u * (255u - t)), acc);
  __syncthreads();
  for (uint32_t s = 128; s > 0; s >>= 1) {
    if (t < s) red[t] ^= red[t + s];
    __syncthreads();
  }
  if (t == 0) tile_crc[blockIdx.x] = red[0];
}

}  // namespace scfq_dinflate
