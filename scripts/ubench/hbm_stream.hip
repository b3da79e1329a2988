// Micro-benchmark: read-stream ceilings on MI355X for the load structures the scan kernel could use.
//   (a) LDS-DMA ring exactly like fq_scan_tiles (one wave = contiguous range of 4 KiB tiles, RING slots, nt or not),
//       consuming each tile with a single ds_read per lane (no VALU work)
//   (b) plain global_load_dwordx4 grid-stride sum (register path)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
constexpr int kTile = 4096;
template <bool NT>
__device__ __forceinline__ void glds_tile(const uint8_t* lane_src, uint32_t lds) {
  uint32_t keep;
  if (NT) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off nt\n\tglobal_load_lds_dwordx4 %1, off offset:1024 nt\n\t"
      "global_load_lds_dwordx4 %1, off offset:2048 nt\n\tglobal_load_lds_dwordx4 %1, off offset:3072 nt\n\ts_mov_b32 m0, %0"
      : "=&s"(keep) : "v"(lane_src), "s"(lds) : "memory");
  else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\t"
      "global_load_lds_dwordx4 %1, off offset:2048\n\tglobal_load_lds_dwordx4 %1, off offset:3072\n\ts_mov_b32 m0, %0"
      : "=&s"(keep) : "v"(lane_src), "s"(lds) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int RING, bool NT, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_ring(const uint8_t* base, uint64_t n_tiles, uint32_t tpr, uint32_t* out) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int lane = threadIdx.x & 63;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  uint8_t* ring = smem + wave * RING * kTile;
  const uint32_t ring_lds = (uint32_t)(uintptr_t)ring;
  const uint64_t range = (uint64_t)blockIdx.x * WAVES + wave;
  const uint64_t t0 = range * tpr;
  uint64_t t1 = t0 + tpr; if (t1 > n_tiles) t1 = n_tiles;
  if (t0 >= n_tiles) return;
  uint32_t acc = 0;
#pragma unroll
  for (int k = 0; k < RING - 1; ++k)
    if (t0 + k < t1) glds_tile<NT>(base + (t0 + k) * kTile + lane * 16, (uint32_t)__builtin_amdgcn_readfirstlane((int)(ring_lds + k * kTile)));
  uint32_t slot = 0;
  for (uint64_t t = t0; t < t1; ++t) {
    const uint32_t s2 = slot >= 1 ? slot - 1 : RING - 1;
    if (t + RING - 1 < t1) glds_tile<NT>(base + (t + RING - 1) * kTile + lane * 16, (uint32_t)__builtin_amdgcn_readfirstlane((int)(ring_lds + s2 * kTile)));
    const uint64_t after = t1 - 1 - t;
    if (after >= (uint64_t)(RING - 1)) wait_vm<4 * (RING - 1)>();
    else if (RING > 3 && after == 2) wait_vm<8>();
    else if (RING > 2 && after == 1) wait_vm<4>();
    else wait_vm<0>();
    acc += *reinterpret_cast<const uint32_t*>(ring + slot * kTile + lane * 64);
    slot = slot == RING - 1 ? 0 : slot + 1;
  }
  out[range * 64 + lane] = acc;
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_regs(const u32x4* p, uint64_t n16, uint32_t* out) {
  uint32_t acc = 0;
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    u32x4 a = __builtin_nontemporal_load(p + i), b = __builtin_nontemporal_load(p + i + stride);
    u32x4 c = __builtin_nontemporal_load(p + i + 2 * stride), d = __builtin_nontemporal_load(p + i + 3 * stride);
    acc += a.x ^ b.y ^ c.z ^ d.w;
  }
  for (; i < n16; i += stride) acc += p[i].x;
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int RING, bool NT, int WAVES>
void run_ring(const uint8_t* d, uint64_t bytes, uint32_t tpr, uint32_t* out) {
  const uint64_t n_tiles = bytes / kTile, ranges = (n_tiles + tpr - 1) / tpr;
  const unsigned blocks = (unsigned)((ranges + WAVES - 1) / WAVES);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int it = 0; it < 6; ++it) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_ring<RING, NT, WAVES>), dim3(blocks), dim3(64 * WAVES), WAVES * RING * kTile, 0, d, n_tiles, tpr, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (it && ms < best) best = ms;
  }
  printf("lds-dma ring=%d nt=%d waves/wg=%d tpr=%u : %.3f ms  %.1f GB/s\n", RING, (int)NT, WAVES, tpr, best, bytes / best / 1e6);
}

int main() {
  const uint64_t bytes = 10ull * 1000 * 1000 * 1000 / kTile * kTile;
  uint8_t* d; hipMalloc(&d, bytes); hipMemset(d, 0x41, bytes);
  uint32_t* out; hipMalloc(&out, 64ull << 20);
  run_ring<2, true, 4>(d, bytes, 100, out); run_ring<2, false, 4>(d, bytes, 100, out);
  run_ring<3, true, 4>(d, bytes, 100, out); run_ring<4, true, 4>(d, bytes, 100, out);
  run_ring<2, true, 8>(d, bytes, 100, out); run_ring<2, true, 2>(d, bytes, 100, out);
  run_ring<2, true, 4>(d, bytes, 25, out); run_ring<2, true, 4>(d, bytes, 400, out);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int blocks : {2048, 4096, 8192}) {
    float best = 1e9;
    for (int it = 0; it < 6; ++it) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_regs, dim3(blocks), dim3(256), 0, 0, (const u32x4*)d, bytes / 16, out);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (it && ms < best) best = ms;
    }
    printf("global_load_dwordx4 nt, %d blocks grid-stride : %.3f ms  %.1f GB/s\n", blocks, best, bytes / best / 1e6);
  }
  return 0;
}
