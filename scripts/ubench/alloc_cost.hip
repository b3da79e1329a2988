// Micro-benchmark: what a first call pays for device memory on MI355X (the cold `sc fq-count x.fq.gz` of round 3).
//   (a) hipMalloc / hipFree by size, first and second time in the process
//   (b) the same bytes as N smaller hipMallocs
//   (c) virtual-memory form: hipMemAddressReserve once, hipMemCreate + hipMemMap + hipMemSetAccess per granule
//   (d) hipMallocAsync from an own pool
//   (e) first touch (memset kernel) of fresh memory against a second pass
//   (f) a hipMalloc on a second thread while a long kernel runs: does either wait for the other?
//   (g) hipHostMalloc (pinned staging)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <atomic>
#include <string>
#include <thread>
#include <unistd.h>
#include <vector>
using clk = std::chrono::steady_clock;
static double ms_since(clk::time_point t) { return std::chrono::duration<double, std::milli>(clk::now() - t).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("FAIL %s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_fill(uint64_t* p, uint64_t n8, uint64_t v) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) p[i] = v;
}
__global__ void k_spin(uint64_t* out, uint64_t cycles) {
  const uint64_t t0 = wall_clock64();
  uint64_t t = t0;
  while (t - t0 < cycles) t = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t - t0;
}

int main(int argc, char** argv) {
  const double max_gb = argc > 1 ? std::atof(argv[1]) : 32.0;
  auto t0 = clk::now();
  CK(hipSetDevice(0));
  CK(hipFree(nullptr));
  std::printf("context up: %.1f ms\n", ms_since(t0));
  if (argc > 2 && std::string(argv[1]) == "busy") {
    // alloc_cost busy <GB> <GB> ...: the same allocations while the device is busy the way an ingest keeps it busy — H2D copies of
    // 16 MiB pieces from pinned memory on one stream, a kernel that streams 4 GB on another
    hipStream_t s_copy, s_k;
    CK(hipStreamCreateWithFlags(&s_copy, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s_k, hipStreamNonBlocking));
    void *pin = nullptr, *dst = nullptr, *work = nullptr;
    CK(hipHostMalloc(&pin, 16u << 20, hipHostMallocDefault));
    CK(hipMalloc(&dst, 16u << 20));
    CK(hipMalloc(&work, (size_t)4 << 30));
    std::atomic<bool> stop{false};
    std::thread feeder([&] {
      (void)hipSetDevice(0);
      while (!stop) {
        for (int i = 0; i < 8; ++i) (void)hipMemcpyAsync(dst, pin, 16u << 20, hipMemcpyHostToDevice, s_copy);
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, s_k, (uint64_t*)work, ((size_t)4 << 30) / 8, 9ull);
        (void)hipStreamSynchronize(s_copy);
      }
    });
    std::this_thread::sleep_for(std::chrono::milliseconds(20));
    double total = 0;
    for (int i = 2; i < argc; ++i) {
      const size_t n = (size_t)(std::atof(argv[i]) * 1e9) & ~4095ull;
      void* p = nullptr;
      auto t = clk::now();
      CK(hipMalloc(&p, n));
      const double a = ms_since(t);
      total += a;
      std::printf("  busy hipMalloc %7.3f GB: %9.2f ms\n", n / 1e9, a);
    }
    stop = true;
    feeder.join();
    std::printf("  all allocations under copies + kernels: %.2f ms\n", total);
    std::fflush(stdout);
    _exit(0);
  }
  if (argc > 2 && std::string(argv[1]) == "pattern") {
    // alloc_cost pattern <GB> <GB> ...: the allocations of one process, in order, each timed; every buffer is then written once
    // and the process exits WITHOUT freeing (what a CLI does) — run it twice back to back to see what the second process pays
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    std::vector<std::pair<void*, size_t>> bufs;
    double total = 0;
    for (int i = 2; i < argc; ++i) {
      const size_t n = (size_t)(std::atof(argv[i]) * 1e9) & ~4095ull;
      void* p = nullptr;
      auto t = clk::now();
      CK(hipMalloc(&p, n));
      const double a = ms_since(t);
      total += a;
      std::printf("  hipMalloc %7.3f GB: %9.2f ms\n", n / 1e9, a);
      bufs.emplace_back(p, n);
    }
    auto t = clk::now();
    for (auto& b : bufs) hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, st, (uint64_t*)b.first, b.second / 8, 7ull);
    CK(hipStreamSynchronize(st));
    std::printf("  all allocations %.2f ms, fill of everything %.2f ms, process up for %.1f ms\n", total, ms_since(t), ms_since(t0));
    std::fflush(stdout);
    _exit(0);
  }
  size_t fr = 0, tot = 0;
  CK(hipMemGetInfo(&fr, &tot));
  std::printf("free %.1f GB of %.1f GB\n", fr / 1e9, tot / 1e9);
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));

  // (a) + (e)
  for (double gb : {0.064, 0.25, 1.0, 4.0, 16.0, max_gb}) {
    const size_t n = (size_t)(gb * 1e9) & ~4095ull;
    for (int rep = 0; rep < 2; ++rep) {
      void* p = nullptr;
      auto t = clk::now();
      CK(hipMalloc(&p, n));
      const double a = ms_since(t);
      t = clk::now();
      hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, st, (uint64_t*)p, n / 8, 1ull);
      CK(hipStreamSynchronize(st));
      const double f1 = ms_since(t);
      t = clk::now();
      hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, st, (uint64_t*)p, n / 8, 2ull);
      CK(hipStreamSynchronize(st));
      const double f2 = ms_since(t);
      t = clk::now();
      CK(hipFree(p));
      const double fr_ms = ms_since(t);
      std::printf("hipMalloc %7.3f GB rep %d: alloc %8.2f ms (%6.2f ms/GB)  first fill %7.2f  second fill %7.2f  free %7.2f ms\n", gb, rep, a, a / gb, f1, f2, fr_ms);
    }
  }
  // (b) 16 GB as 64 x 256 MB
  {
    std::vector<void*> ps(64, nullptr);
    auto t = clk::now();
    for (auto& p : ps) CK(hipMalloc(&p, 256u << 20));
    const double a = ms_since(t);
    t = clk::now();
    for (auto& p : ps) CK(hipFree(p));
    std::printf("64 x 256 MiB hipMalloc: %8.2f ms (%6.2f ms/GB), free %7.2f ms\n", a, a / 17.18, ms_since(t));
  }
  // (c) VMM
  {
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    hipError_t e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
    if (e != hipSuccess) { std::printf("VMM: not available (%s)\n", hipGetErrorString(e)); (void)hipGetLastError(); }
    else {
      std::printf("VMM granularity %zu\n", gran);
      const size_t total = (size_t)64 << 30;
      void* va = nullptr;
      auto t = clk::now();
      e = hipMemAddressReserve(&va, total, 0, nullptr, 0);
      std::printf("VMM reserve 64 GiB: %s, %.2f ms\n", hipGetErrorString(e), ms_since(t));
      if (e == hipSuccess) {
        for (size_t chunk : {(size_t)256 << 20, (size_t)1 << 30, (size_t)4 << 30}) {
          const int n_chunks = (int)(((size_t)8 << 30) / chunk);
          std::vector<hipMemGenericAllocationHandle_t> hs(n_chunks);
          double t_create = 0, t_map = 0, t_acc = 0;
          bool ok = true;
          for (int i = 0; i < n_chunks && ok; ++i) {
            auto a0 = clk::now();
            ok = hipMemCreate(&hs[i], chunk, &prop, 0) == hipSuccess;
            t_create += ms_since(a0);
            a0 = clk::now();
            ok = ok && hipMemMap((char*)va + (size_t)i * chunk, chunk, 0, hs[i], 0) == hipSuccess;
            t_map += ms_since(a0);
            hipMemAccessDesc ad{};
            ad.location = prop.location;
            ad.flags = hipMemAccessFlagsProtReadWrite;
            a0 = clk::now();
            ok = ok && hipMemSetAccess((char*)va + (size_t)i * chunk, chunk, &ad, 1) == hipSuccess;
            t_acc += ms_since(a0);
          }
          if (!ok) { std::printf("VMM chunk %zu MiB failed: %s\n", chunk >> 20, hipGetErrorString(hipGetLastError())); break; }
          auto a0 = clk::now();
          hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, st, (uint64_t*)va, ((size_t)8 << 30) / 8, 3ull);
          CK(hipStreamSynchronize(st));
          const double fill = ms_since(a0);
          a0 = clk::now();
          for (int i = 0; i < n_chunks; ++i) { (void)hipMemUnmap((char*)va + (size_t)i * chunk, chunk); (void)hipMemRelease(hs[i]); }
          std::printf("VMM 8 GiB in %4zu MiB chunks: create %7.2f map %7.2f access %7.2f ms (%.2f ms/GB), fill %.2f, unmap+release %.2f ms\n", chunk >> 20, t_create, t_map, t_acc,
                      (t_create + t_map + t_acc) / 8.59, fill, ms_since(a0));
        }
        (void)hipMemAddressFree(va, total);
      }
    }
  }
  // (d) pool
  {
    hipMemPoolProps pp{};
    pp.allocType = hipMemAllocationTypePinned;
    pp.location.type = hipMemLocationTypeDevice;
    pp.location.id = 0;
    hipMemPool_t pool;
    if (hipMemPoolCreate(&pool, &pp) == hipSuccess) {
      uint64_t thr = ~0ull;
      (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr);
      for (int rep = 0; rep < 2; ++rep) {
        void* p = nullptr;
        auto t = clk::now();
        CK(hipMallocFromPoolAsync(&p, (size_t)8 << 30, pool, st));
        CK(hipStreamSynchronize(st));
        const double a = ms_since(t);
        t = clk::now();
        CK(hipFreeAsync(p, st));
        CK(hipStreamSynchronize(st));
        std::printf("pool 8 GiB rep %d: alloc %.2f ms, free %.2f ms\n", rep, a, ms_since(t));
      }
      (void)hipMemPoolDestroy(pool);
    } else { std::printf("pool: not available\n"); (void)hipGetLastError(); }
  }
  // (f) hipMalloc on a second thread under a running kernel
  {
    uint64_t* d = nullptr;
    CK(hipMalloc(&d, 4096));
    auto t = clk::now();
    hipLaunchKernelGGL(k_spin, dim3(256 * 8), dim3(256), 0, st, d, 100000000ull * 2 / 10);      // wall_clock64 ticks at 100 MHz: 200 ms
    double alloc_ms = 0, fill_ms = 0;
    std::thread th([&] {
      (void)hipSetDevice(0);
      void* p = nullptr;
      auto a0 = clk::now();
      (void)hipMalloc(&p, (size_t)8 << 30);
      alloc_ms = ms_since(a0);
      hipStream_t s2;
      (void)hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
      a0 = clk::now();
      hipLaunchKernelGGL(k_fill, dim3(64), dim3(256), 0, s2, (uint64_t*)p, (64u << 20) / 8, 5ull);
      (void)hipStreamSynchronize(s2);
      fill_ms = ms_since(a0);
      (void)hipFree(p);
      (void)hipStreamDestroy(s2);
    });
    CK(hipStreamSynchronize(st));
    const double spin = ms_since(t);
    th.join();
    std::printf("under a %.1f ms spinning kernel: 8 GiB hipMalloc on a 2nd thread took %.2f ms, a 64 MiB fill behind it %.2f ms\n", spin, alloc_ms, fill_ms);
    CK(hipFree(d));
  }
  // (g) pinned
  for (size_t mb : {64, 256}) {
    void* p = nullptr;
    auto t = clk::now();
    CK(hipHostMalloc(&p, mb << 20, hipHostMallocDefault));
    const double a = ms_since(t);
    t = clk::now();
    CK(hipHostFree(p));
    std::printf("hipHostMalloc %zu MiB: %.2f ms, free %.2f ms\n", mb, a, ms_since(t));
  }
  return 0;
}
