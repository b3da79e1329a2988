// scfq_arena.hpp — device memory of the device gzip path in PIECES, never one allocation of tens of GB.
//
// Measured on MI355X (scripts/ubench/alloc_cost.hip, scripts/gpu_alloc_patterns.sh, profiles/r03/alloc_patterns.txt): hipMalloc of a
// piece of up to ~16 GB costs 0.3 ms whatever ran before (72 GB as 18 pieces of 4 GB: 4.4 ms); ONE hipMalloc of 28 GB costs 0.5 s in a
// fresh process and 2.6 s in a process that follows another one that used the same memory.  Round 2 sized the symbol pools of a
// 10 GB .gz as two allocations of 28 GB and one of 11 GB: 1.5 - 2.3 s of every first call (`sc fq-count x.fq.gz` is one call per
// process, sc.nim:114-116), ten times the inflate itself.
//
// The virtual-memory API (hipMemAddressReserve / hipMemCreate / hipMemMap: contiguous range, pieces mapped side by side) was tried first
// and dropped (scripts/ubench/vmm_probe.hip, profiles/r03/vmm_probe.txt): on this runtime pieces of different sizes in one range fail
// with "invalid value", and kernels over a range whose address had been mapped, unmapped and mapped again summed WRONG bytes.
//
// So nothing here needs a large contiguous range:
//   SymPool   the 16-bit symbols of a batch's segments.  A segment needs ITS symbols in one piece, not the batch: chunks of 512 MB,
//             allocated as segments are placed, a segment's offset given relative to the first chunk (the kernels add it to that
//             base with 64-bit wrap-around: chunks may lie anywhere)
//   DevBuf    a plain buffer that is REPLACED by a bigger one when a batch needs more; the old one may still be read by a kernel
//             in flight, so it goes to a list that is freed when the call ends (hipFree waits for the device)
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <vector>

namespace scfq_arena {

struct DevBuf {
  uint8_t* p = nullptr;
  uint64_t cap = 0;
  // at least `want` bytes; a replaced buffer is appended to `retired` (freed by the caller when the device is idle).
  // *delta: change of the bytes held.  hipSuccess or the allocation's error (the buffer is then unchanged).
  hipError_t ensure(uint64_t want, std::vector<void*>* retired, int64_t* delta) {
    *delta = 0;
    if (want <= cap) return hipSuccess;
    const uint64_t bytes = (want + want / 8 + 4095) & ~4095ull;
    void* q = nullptr;
    const hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess) { (void)hipGetLastError(); return e; }
    if (p) retired->push_back(p);
    *delta = (int64_t)bytes - (int64_t)cap;      // (a retired buffer counts as given back: it is freed before the call returns)
    p = static_cast<uint8_t*>(q);
    cap = bytes;
    return hipSuccess;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

class SymPool {
 public:
  static constexpr uint64_t kChunkSyms = 1ull << 28;      // 512 MB of 16-bit symbols per chunk (about five hundred segments; r3 - r4: 1 GB — a set of 2.4 GB then held 3)
  void rewind() { cur_ = 0; used_ = 0; }                   // a new batch places its segments from the start again
  uint16_t* base() const { return chunks_.empty() ? nullptr : chunks_[0].p; }
  uint64_t bytes() const { uint64_t t = 0; for (const Chunk& c : chunks_) t += 2 * c.cap; return t; }
  // room for n symbols (a multiple of 8) in one piece: *off = its offset from base(), in symbols, modulo 2^64.
  // *delta: bytes newly allocated.  hipSuccess or the allocation's error.
  hipError_t take(uint64_t n, uint64_t* off, int64_t* delta) {
    *delta = 0;
    while (cur_ < chunks_.size() && chunks_[cur_].cap - used_ < n) { ++cur_; used_ = 0; }
    if (cur_ == chunks_.size()) {
      const uint64_t cap = std::max<uint64_t>(kChunkSyms, (n + 7) & ~7ull);
      void* q = nullptr;
      const hipError_t e = hipMalloc(&q, 2 * cap);
      if (e != hipSuccess) { (void)hipGetLastError(); return e; }
      chunks_.push_back(Chunk{static_cast<uint16_t*>(q), cap});
      *delta = (int64_t)(2 * cap);
      used_ = 0;
    }
    // (the chunks are unrelated allocations: their distance is taken between ADDRESSES — two's complement when the chunk lies below the
    // first one; hipMalloc hands out 256-byte-aligned pointers, so the division by the symbol size is exact, and an arithmetic shift
    // keeps a negative distance negative)
    const int64_t dist = (int64_t)((uintptr_t)chunks_[cur_].p - (uintptr_t)chunks_[0].p);
    *off = (uint64_t)(dist >> 1) + used_;
    used_ += n;
    return hipSuccess;
  }
  void release() { for (Chunk& c : chunks_) (void)hipFree(c.p); chunks_.clear(); rewind(); }

 private:
  struct Chunk { uint16_t* p; uint64_t cap; };
  std::vector<Chunk> chunks_;
  size_t cur_ = 0;
  uint64_t used_ = 0;
};

}  // namespace scfq_arena
