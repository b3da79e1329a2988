// scfq_bgzf.hpp — parallel host inflate for BGZF (blocked gzip, what `bgzip` writes) inputs.
//
// the reference reads ".gz" through zlib gzread one byte at a time
// (src/fq_count.nim:32 -> zip/gzipfiles; same calls as gzip_stream.nim:16-17), a single serial inflate
// stream. A gzip file made of many small members whose compressed size is recorded in the header
// (BGZF: extra subfield 'B','C' = block size - 1) can be inflated block-parallel, because every
// block's input range and output size (ISIZE trailer) are known before inflating. The bytes produced
// are exactly what gzread yields for the same file (concatenation of all members); anything that
// is not a well-formed BGZF block hands the rest of the file back to serial gzread.
#pragma once
#include <unistd.h>
#include <zlib.h>

#include "scfq_crc32.hpp"
#include "scfq_inflate.hpp"

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

namespace scfq_bgzf {

inline uint16_t rd16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
inline uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

// Parses one member header at p (n bytes available). Returns total block size and header length, or 0 when the
// bytes are not a BGZF block header (plain gzip member, garbage, or too few bytes).
inline uint32_t block_size(const uint8_t* p, uint64_t n, uint32_t* hdr_len) {
  if (n < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || p[3] != 4) return 0;   // FLG must be exactly FEXTRA
  const uint32_t xlen = rd16(p + 10);
  if (12ull + xlen > n) return 0;
  uint32_t q = 12, bsize = 0;
  while (q + 4 <= 12 + xlen) {
    const uint32_t slen = rd16(p + q + 2);
    if (p[q] == 'B' && p[q + 1] == 'C' && slen == 2 && q + 6 <= 12 + xlen) bsize = (uint32_t)rd16(p + q + 4) + 1;
    q += 4 + slen;
  }
  if (bsize < 12 + xlen + 8) return 0;
  *hdr_len = 12 + xlen;
  return bsize;
}

inline bool probe(int fd) {
  uint8_t h[64];
  const ssize_t n = pread(fd, h, sizeof h, 0);
  uint32_t hl;
  return n >= 18 && block_size(h, (uint64_t)n, &hl) != 0;
}

struct Block { uint64_t in_off; uint32_t in_len, hdr_len, isize, crc; uint64_t out_off; };

// Inflate blocks[lo,hi) from cbuf into dst. Returns 0 or -1 (corrupt block / CRC / length mismatch).
// Default: the library's own decoder (scfq_inflate.hpp) into a thread-local scratch (its fast loop writes a little past
// the end of what it produces, and the neighbouring block's bytes belong to another thread), then one memcpy.
// SCFQ_INFLATE=zlib, or a block that claims more than 64 KiB, takes the zlib path below.
inline int inflate_blocks_zlib(const uint8_t* cbuf, const std::vector<Block>& blocks, size_t lo, size_t hi, uint8_t* dst);

inline int inflate_blocks(const uint8_t* cbuf, const std::vector<Block>& blocks, size_t lo, size_t hi, uint8_t* dst) {
  static const bool own = [] { const char* e = std::getenv("SCFQ_INFLATE"); return !(e && e[0] == 'z'); }();
  if (!own) return inflate_blocks_zlib(cbuf, blocks, lo, hi, dst);
  constexpr uint32_t kMaxBlock = 1u << 16;
  std::vector<uint8_t> scratch(kMaxBlock + 2 * scfq_inflate::kOutSlack);
  auto dec = std::unique_ptr<scfq_inflate::Decoder>(new scfq_inflate::Decoder());
  for (size_t i = lo; i < hi; ++i) {
    const Block& b = blocks[i];
    if (b.isize > kMaxBlock) { if (inflate_blocks_zlib(cbuf, blocks, i, i + 1, dst)) return -1; continue; }
    const uint8_t* in = cbuf + b.in_off + b.hdr_len;
    dec->begin(in, in + (b.in_len - b.hdr_len - 8));
    uint8_t* out = scratch.data();
    const int r = dec->run(out, scratch.data() + scratch.size());
    if (r != scfq_inflate::kStreamEnd || (uint32_t)(out - scratch.data()) != b.isize) return -1;
    if (scfq_crc::crc32(0u, scratch.data(), b.isize) != b.crc) return -1;
    std::memcpy(dst + b.out_off, scratch.data(), b.isize);
  }
  return 0;
}

inline int inflate_blocks_zlib(const uint8_t* cbuf, const std::vector<Block>& blocks, size_t lo, size_t hi, uint8_t* dst) {
  z_stream zs;
  std::memset(&zs, 0, sizeof zs);
  if (inflateInit2(&zs, -15) != Z_OK) return -1;
  int rc = 0;
  for (size_t i = lo; i < hi && !rc; ++i) {
    const Block& b = blocks[i];
    inflateReset(&zs);
    zs.next_in = const_cast<Bytef*>(cbuf + b.in_off + b.hdr_len);
    zs.avail_in = b.in_len - b.hdr_len - 8;
    zs.next_out = dst + b.out_off;
    zs.avail_out = b.isize;
    const int r = inflate(&zs, Z_FINISH);
    if (r != Z_STREAM_END || zs.avail_out != 0) { rc = -1; break; }
    if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), dst + b.out_off, b.isize) != b.crc) rc = -1;
  }
  inflateEnd(&zs);
  return rc;
}

}  // namespace scfq_bgzf
