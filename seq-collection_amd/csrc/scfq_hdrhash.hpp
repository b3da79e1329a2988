// scfq_hdrhash.hpp — the header hash of fq-dedup, in a form that can be computed word by word in ANY order (r4).
//
// fq-dedup groups records by a hash of their header line and settles every group by exact string compares (scfq_dedup.hip), so the
// hash only has to spread well; it is internal and never leaves the device.  A header is cut into 8-byte little-endian words (the last
// one zero-padded); word k contributes hh_word(lo, hi, k) to two 32-bit sums, and hh_final mixes the sums with the length.  Because the
// contributions are SUMMED, eight lanes can each take one word of a header and add up with three cross-lane steps — which is how
// the line-index pass hashes the headers it has in LDS anyway (fq_scan_kernels.hpp: fq_index_pos) — and a thread that walks a header
// alone (scfq_dedup.hip: the records the index pass could not do) gets the same value.  32-bit full-rate operations only, one
// v_mul_lo_u32 per half word: a 64-bit multiply-xorshift per word (r1 - r3) is sixteen quarter-rate instructions.
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>

namespace scfq_hdrhash {

// the index pass hands a header's two sums over in 56 bits (A, 24 bits of B), its length (<= kMaxLen) in the 8 above them
constexpr uint32_t kHashBits = 56;
constexpr uint32_t kMaxLen = 255;

__host__ __device__ __forceinline__ void hh_word(uint32_t lo, uint32_t hi, uint32_t k, uint32_t& A, uint32_t& B) {
  // (the word's place enters through 24-bit multiplies: full rate, and exact for every k below 2^24 — a header of 128 MB)
  const uint32_t k1 = (k + 1u) & 0xFFFFFFu;
  uint32_t a = lo + k1 * 0x9E3779u;
  uint32_t b = hi ^ (k1 * 0x85EBCBu);
  a ^= a >> 15; a *= 0x2C1B3C6Du; a ^= a >> 12;
  b ^= b >> 13; b *= 0x297A2D39u; b ^= b >> 15;
  A += a ^ (b << 7 | b >> 25);
  B += b + (a << 11 | a >> 21);
}

// the two sums, the length and the seed mixed into 64 bits: the sums are sums of well-mixed words already, so two 32-bit multiplies
// and a few shifts are enough to spread them over the bits the sort looks at (87 K colliding pairs among 28 M records at 32 bits,
// as n^2 / 2^33 predicts).  Of B its low 24 bits: the index pass hands the sums over as A | B << 32 | length << 56 and leaves this
// step to the kernel behind it.
__host__ __device__ __forceinline__ uint64_t hh_final(uint32_t A, uint32_t B, uint64_t len, uint64_t seed) {
  const uint32_t l = (uint32_t)len ^ (uint32_t)(len >> 32);
  uint32_t a = A ^ (uint32_t)seed ^ (l + (l << 13));
  uint32_t b = (B & 0xFFFFFFu) ^ (uint32_t)(seed >> 32);
  a ^= b << 9; a ^= a >> 16; a *= 0x85EBCA6Bu; a ^= a >> 13;
  b += a; b ^= b >> 15; b *= 0xC2B2AE35u; b ^= b >> 16;
  a ^= b;
  return (uint64_t)a << 32 | b;
}

}  // namespace scfq_hdrhash
