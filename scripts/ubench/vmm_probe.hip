// Which piece layouts does the virtual-memory API take?  (round 3: an arena's second piece failed with "invalid argument")
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void k_fill(uint64_t* p, uint64_t n8, uint64_t v) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) p[i] = v + i;
}
__global__ void k_sum(const uint64_t* p, uint64_t n8, unsigned long long* out) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  unsigned long long s = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) s += p[i];
  atomicAdd(out, s);
}
int main() {
  hipSetDevice(0);
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  size_t gmin = 0, grec = 0;
  hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum);
  hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended);
  std::printf("granularity min %zu recommended %zu\n", gmin, grec);
  hipMemAccessDesc ad{};
  ad.location = prop.location;
  ad.flags = hipMemAccessFlagsProtReadWrite;
  const size_t M = 1ull << 20;
  std::vector<std::vector<size_t>> layouts = {{2 * M, 2 * M}, {2 * M, 4 * M}, {4 * M, 2 * M}, {1024 * M, 2 * M}, {2 * M, 1024 * M}, {6 * M, 10 * M, 1024 * M, 34 * M}, {64 * M, 64 * M, 64 * M}};
  for (int align_case = 0; align_case < 2; ++align_case)
  for (auto& lay : layouts) {
    void* va = nullptr;
    hipError_t e = hipMemAddressReserve(&va, 4096 * M, align_case ? 2 * M : 0, nullptr, 0);
    std::printf("reserve(align %s): %s  %p |", align_case ? "2M" : "0", hipGetErrorName(e), va);
    if (e != hipSuccess) { std::printf("\n"); continue; }
    size_t at = 0;
    std::vector<hipMemGenericAllocationHandle_t> hs;
    bool ok = true;
    for (size_t sz : lay) {
      hipMemGenericAllocationHandle_t h;
      e = hipMemCreate(&h, sz, &prop, 0);
      if (e != hipSuccess) { std::printf(" create(%zuM) %s", sz / M, hipGetErrorName(e)); ok = false; break; }
      hs.push_back(h);
      e = hipMemMap((char*)va + at, sz, 0, h, 0);
      if (e != hipSuccess) { std::printf(" map(%zuM at %zuM) %s", sz / M, at / M, hipGetErrorName(e)); ok = false; break; }
      e = hipMemSetAccess((char*)va + at, sz, &ad, 1);
      if (e != hipSuccess) { std::printf(" access(%zuM at %zuM) %s", sz / M, at / M, hipGetErrorName(e)); ok = false; break; }
      std::printf(" %zuM ok", sz / M);
      at += sz;
    }
    (void)hipGetLastError();
    if (ok) {
      unsigned long long* d = nullptr; hipMalloc(&d, 8); hipMemset(d, 0, 8);
      hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, 0, (uint64_t*)va, at / 8, 5ull);
      hipLaunchKernelGGL(k_sum, dim3(1024), dim3(256), 0, 0, (const uint64_t*)va, at / 8, d);
      unsigned long long got = 0; e = hipMemcpy(&got, d, 8, hipMemcpyDeviceToHost);
      const unsigned long long n = at / 8, want = 5ull * n + n * (n - 1) / 2;
      std::printf(" | kernels over %zuM: %s %s", at / M, hipGetErrorName(e), got == want ? "sum ok" : "SUM WRONG");
      // host -> device copy across a piece border
      std::vector<uint64_t> hb(1 << 20, 7); e = hipMemcpy((char*)va + lay[0] - 4 * M / 8, hb.data(), hb.size() * 8 < at - lay[0] + M / 2 ? hb.size() * 8 : M / 2, hipMemcpyHostToDevice);
      std::printf(" memcpy across border: %s", hipGetErrorName(e));
      hipFree(d);
    }
    std::printf("\n");
    size_t o = 0;
    for (size_t i = 0; i < hs.size(); ++i) { hipMemUnmap((char*)va + o, lay[i]); hipMemRelease(hs[i]); o += lay[i]; }
    hipMemAddressFree(va, 4096 * M);
    (void)hipGetLastError();
  }
  return 0;
}
