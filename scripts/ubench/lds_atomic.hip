// Micro-benchmark: LDS atomic (ds_add_u32, no return) cost per wave-instruction per CU on MI355X as a function of
// active lanes and address pattern. 8 waves per CU (2 per SIMD), each issuing ITER x 16 atomics.
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 2048
template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t* out) {
  __shared__ uint32_t h[8192];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 8192; i += 256) h[i] = 0;
  __syncthreads();
  uint32_t idx;
  bool active = true;
  if (MODE == 0) idx = w * 64 + lane;                       // 64 lanes, distinct consecutive dwords
  if (MODE == 1) idx = w * 64 + (lane & 15);                // 64 lanes on 16 addresses (4-way same address)
  if (MODE == 2) idx = w * 64 + (lane & 3);                 // 64 lanes on 4 addresses (16-way)
  if (MODE == 3) idx = w * 64;                              // 64 lanes on 1 address
  if (MODE == 4) { idx = w * 64 + lane; active = lane < 16; }   // 16 active lanes, distinct
  if (MODE == 5) { idx = w * 64 + lane; active = (lane & 3) == 0; }   // 16 active lanes spread
  if (MODE == 6) idx = w * 64 + lane * 17 % 64 + (lane & 1) * 1024;  // distinct, scattered
  if (MODE == 7) { idx = w * 64 + lane; active = lane < 4; }
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (active) __hip_atomic_fetch_add(&h[idx], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();
  out[blockIdx.x * 256 + threadIdx.x] = h[threadIdx.x];
}
template <int MODE> void run(const char* name, uint32_t* d) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * 2;   // 2 blocks of 4 waves per CU
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_cu = 8.0 * ITER * 16;
  printf("%-44s %.3f ms -> %.1f ns per wave-atomic per CU (%.1f cycles @2.0 GHz)\n", name, ms, ms * 1e6 / instr_per_cu, ms * 1e6 / instr_per_cu * 2.0);
}
int main() {
  uint32_t* d; hipMalloc(&d, 512 * 256 * 4);
  run<0>("64 lanes, 64 distinct consecutive", d); run<6>("64 lanes, distinct scattered", d);
  run<1>("64 lanes on 16 addresses", d); run<2>("64 lanes on 4 addresses", d); run<3>("64 lanes on 1 address", d);
  run<4>("16 active lanes (0..15), distinct", d); run<5>("16 active lanes (every 4th), distinct", d); run<7>("4 active lanes", d);
  return 0;
}
