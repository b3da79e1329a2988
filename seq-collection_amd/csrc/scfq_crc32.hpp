// scfq_crc32.hpp — CRC-32 (the gzip one: reflected 0xEDB88320) by carry-less multiplication, for the host inflate paths.
// zlib 1.2.11's crc32() does ≈1.9 GB/s on the GPU box's cores, which is the speed of the inflate it is supposed to check; the
// folding method (Gopal et al., "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ") does >10 GB/s.
// Constants are x^n mod P, bit-reflected and shifted left by one, n = 544, 480 (fold by 512 bits), 160, 96 (fold by 128
// bits), 64; u = floor(x^64 / P) and P for the Barrett step (derivation script in the header of tests/test_crc32_host.py).
// Same interface as zlib's crc32_z: takes and returns the conditioned (inverted) value; falls back to zlib when the CPU
// lacks PCLMULQDQ or the buffer is short.
#pragma once
#include <zlib.h>

#include <cstddef>
#include <cstdint>

#if defined(__x86_64__)
#include <immintrin.h>

namespace scfq_crc {

__attribute__((target("pclmul,sse4.1"))) inline __m128i fold128(__m128i acc, __m128i next, __m128i r4r3) {
  const __m128i h = _mm_clmulepi64_si128(acc, r4r3, 0x11);
  acc = _mm_clmulepi64_si128(acc, r4r3, 0x00);
  return _mm_xor_si128(_mm_xor_si128(acc, h), next);
}

__attribute__((target("pclmul,sse4.1"))) inline uint32_t fold_raw(const uint8_t* p, size_t len /* multiple of 16, >= 64 */, uint32_t raw) {
  const __m128i r2r1 = _mm_set_epi64x(0x1c6e41596ll, 0x154442bd4ll);
  const __m128i r4r3 = _mm_set_epi64x(0x0ccaa009ell, 0x1751997d0ll);
  const __m128i r5 = _mm_set_epi64x(0, 0x163cd6124ll);
  const __m128i mask32 = _mm_set_epi32(0, 0, 0, -1);
  const __m128i ru_poly = _mm_set_epi64x(0x1f7011641ll, 0x1db710641ll);
  __m128i x1 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(p));
  __m128i x2 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(p + 16));
  __m128i x3 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(p + 32));
  __m128i x4 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(p + 48));
  x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)raw));
  p += 64; len -= 64;
  while (len >= 64) {
    const __m128i h1 = _mm_clmulepi64_si128(x1, r2r1, 0x11), h2 = _mm_clmulepi64_si128(x2, r2r1, 0x11);
    const __m128i h3 = _mm_clmulepi64_si128(x3, r2r1, 0x11), h4 = _mm_clmulepi64_si128(x4, r2r1, 0x11);
    x1 = _mm_clmulepi64_si128(x1, r2r1, 0x00); x2 = _mm_clmulepi64_si128(x2, r2r1, 0x00);
    x3 = _mm_clmulepi64_si128(x3, r2r1, 0x00); x4 = _mm_clmulepi64_si128(x4, r2r1, 0x00);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, h1), _mm_loadu_si128(reinterpret_cast<const __m128i*>(p)));
    x2 = _mm_xor_si128(_mm_xor_si128(x2, h2), _mm_loadu_si128(reinterpret_cast<const __m128i*>(p + 16)));
    x3 = _mm_xor_si128(_mm_xor_si128(x3, h3), _mm_loadu_si128(reinterpret_cast<const __m128i*>(p + 32)));
    x4 = _mm_xor_si128(_mm_xor_si128(x4, h4), _mm_loadu_si128(reinterpret_cast<const __m128i*>(p + 48)));
    p += 64; len -= 64;
  }
  // four lanes into one, 128 bits at a time
  x1 = fold128(x1, x2, r4r3);
  x1 = fold128(x1, x3, r4r3);
  x1 = fold128(x1, x4, r4r3);
  while (len >= 16) {
    x1 = fold128(x1, _mm_loadu_si128(reinterpret_cast<const __m128i*>(p)), r4r3);
    p += 16; len -= 16;
  }
  // 128 -> 64 bits (this also appends 32 zero bits), then 64 -> 32 with a Barrett reduction
  __m128i t = _mm_clmulepi64_si128(x1, r4r3, 0x10);            // R4 * x1.low
  x1 = _mm_xor_si128(_mm_srli_si128(x1, 8), t);
  __m128i x2b = _mm_srli_si128(x1, 4);
  x1 = _mm_and_si128(x1, mask32);
  x1 = _mm_clmulepi64_si128(x1, r5, 0x00);
  x1 = _mm_xor_si128(x1, x2b);
  __m128i keep = x1;
  x1 = _mm_and_si128(x1, mask32);
  x1 = _mm_clmulepi64_si128(x1, ru_poly, 0x10);                // * u
  x1 = _mm_and_si128(x1, mask32);
  x1 = _mm_clmulepi64_si128(x1, ru_poly, 0x00);                // * P
  x1 = _mm_xor_si128(x1, keep);
  return (uint32_t)_mm_extract_epi32(x1, 1);
}

inline bool have_pclmul() {
  static const bool v = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
  return v;
}

// crc32_z with the bulk done by carry-less multiplication
inline uint32_t crc32(uint32_t crc, const uint8_t* p, size_t n) {
  if (n < 256 || !have_pclmul()) return (uint32_t)::crc32_z(crc, p, n);
  const size_t bulk = n & ~(size_t)15;
  crc = ~fold_raw(p, bulk, ~crc);
  return (uint32_t)::crc32_z(crc, p + bulk, n - bulk);
}

}  // namespace scfq_crc
#else
namespace scfq_crc {
inline uint32_t crc32(uint32_t crc, const uint8_t* p, size_t n) { return (uint32_t)::crc32_z(crc, p, n); }
}
#endif
