// scfq_index_aux.hpp — internal: what fq-dedup (scfq_dedup.hip) asks of the line index (scfq_api.hip) beyond the offsets.
#pragma once
#include <cstdint>

// With `aux` the index pass hashes the headers it has in LDS (csrc/scfq_hdrhash.hpp; the compact form of the index only) and the
// kernel that knows the line numbers hands out, for every header line behind a newline (record r = line 4r, r >= 1):
// keys[r] = the hash's low hash_bits (all ones: not hashed here), idx[r] = r, hdr[r] = start | length << 40 (length 0xFFFFFF: unknown).
struct scfq_index_aux {
  void* keys;             // device: uint32_t[cap_records] (key_bytes == 4) or uint64_t[cap_records]
  uint32_t* idx;          // device
  uint64_t* hdr;          // device
  uint64_t cap_records;
  uint32_t key_bytes;
  uint32_t hash_bits;     // <= 56
  uint64_t seed;
  uint32_t* unk;          // device, optional: [unk_tiles][4] record numbers that got the all-ones key, tile by tile (0: none)
  uint64_t unk_tiles;     // capacity of unk in tiles
  int filled;             // out: keys / idx / hdr were written (else: the mask form of the index ran; nothing was)
  int unk_complete;       // out: unk lists every record with the all-ones key (but record 0, which has no newline in front of it)
  uint64_t n_tiles;       // out: tiles of the index pass
};

// the index plus flags — bit 0: the input may hold "\r\n" line ends
extern "C" int scfq_index_lines_ex(const void* dptr, uint64_t n, uint64_t* d_line_off, uint64_t cap, uint64_t* lines_out, uint32_t* flags_out);
extern "C" int scfq_index_lines_ex2(const void* dptr, uint64_t n, uint64_t* d_line_off, uint64_t cap, uint64_t* lines_out, uint32_t* flags_out, scfq_index_aux* aux);
