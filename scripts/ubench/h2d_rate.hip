// Micro-benchmark: what the host-to-device feed of the device gzip path reaches on this box, and what slows it (round 5).
// The path moves compressed bytes  page cache -> pinned ring (12 pread threads) -> HBM (hipMemcpyAsync, an SDMA engine), the fill of
// one half of the ring running while the other half crosses PCIe.  Measured here, for pieces of 16 / 64 / 128 MiB:
//   (a) the DMA alone, pieces back to back from a pinned ring nobody writes
//   (b) the DMA while N threads fill the other half by pread from a page-cached file (the path's own pattern)
//   (c) the same with the ring allocated write-combined / non-coherent / NUMA-local to the device
//   (d) a kernel that reads the pinned ring over PCIe itself (zero-copy) instead of an SDMA copy
// and where things live: the CPUs this process may run on, their NUMA nodes, the device's NUMA node.
// usage: h2d_rate [file of >= 1 GiB to pread from]
#include <hip/hip_runtime.h>
#include <sched.h>
#include <fcntl.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <vector>
using clk = std::chrono::steady_clock;
static double ms_since(clk::time_point t) { return std::chrono::duration<double, std::milli>(clk::now() - t).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("FAIL %s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_pull(const uint4* __restrict__ src, uint4* __restrict__ dst, uint64_t n16) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
}

static void fill_piece(int fd, uint64_t off, uint8_t* dst, uint64_t len, int threads) {
  std::vector<std::thread> th;
  const uint64_t per = ((len / threads) + 4095) & ~4095ull;
  for (int t = 0; t < threads; ++t) {
    const uint64_t a = std::min<uint64_t>(len, per * t), b = std::min<uint64_t>(len, per * (t + 1));
    if (a >= b) break;
    th.emplace_back([=] { uint64_t o = a; while (o < b) { const ssize_t r = pread(fd, dst + o, b - o, (off_t)(off + o)); if (r <= 0) break; o += (uint64_t)r; } });
  }
  for (auto& t : th) t.join();
}

int main(int argc, char** argv) {
  CK(hipSetDevice(0));
  CK(hipFree(nullptr));
  // ---- where things live ----
  cpu_set_t set; CPU_ZERO(&set);
  sched_getaffinity(0, sizeof set, &set);
  std::string cpus;
  int ncpu = 0;
  for (int c = 0; c < CPU_SETSIZE; ++c) if (CPU_ISSET(c, &set)) { ++ncpu; if (cpus.size() < 200) cpus += std::to_string(c) + " "; }
  std::printf("allowed CPUs: %d (%s)\n", ncpu, cpus.c_str());
  for (int n = 0; n < 8; ++n) {
    std::ifstream f("/sys/devices/system/node/node" + std::to_string(n) + "/cpulist");
    std::string l; if (f && std::getline(f, l)) std::printf("NUMA node %d CPUs: %s\n", n, l.c_str());
  }
  {
    char bus[64] = {0};
    CK(hipDeviceGetPCIBusId(bus, sizeof bus, 0));
    std::string b = bus; for (auto& ch : b) ch = (char)std::tolower(ch);
    std::ifstream f("/sys/bus/pci/devices/" + b + "/numa_node");
    std::string l; std::printf("device 0 at %s, NUMA node %s\n", bus, (f && std::getline(f, l)) ? l.c_str() : "?");
  }
  const char* path = argc > 1 ? argv[1] : nullptr;
  const int fd = path ? open(path, O_RDONLY) : -1;
  const uint64_t fsize = fd >= 0 ? (uint64_t)lseek(fd, 0, SEEK_END) : 0;
  if (fd >= 0) { std::vector<uint8_t> tmp(64 << 20); for (uint64_t o = 0; o < std::min<uint64_t>(fsize, 2ull << 30); o += tmp.size()) (void)!pread(fd, tmp.data(), tmp.size(), (off_t)o); }      // page cache warm
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  uint8_t* d = nullptr; CK(hipMalloc(&d, 512ull << 20));
  hipEvent_t e0, e1, ev[2]; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&ev[0])); CK(hipEventCreate(&ev[1]));
  // (e) no pinned ring and no engine at all: the host's threads write DEVICE memory themselves — fine-grained device memory is mapped into
  // the process (large BAR), a pread / memcpy into it is posted writes over PCIe.  One pass over host memory instead of two.
  for (unsigned flag : {(unsigned)hipDeviceMallocFinegrained, (unsigned)hipDeviceMallocUncached}) {
    uint8_t* fg = nullptr;
    if (hipExtMallocWithFlags(reinterpret_cast<void**>(&fg), 512ull << 20, flag) != hipSuccess) { (void)hipGetLastError(); std::printf("device memory with flag %u: refused\n", flag); continue; }
    std::vector<uint8_t> src(256ull << 20, 7);
    for (int threads : {4, 8, 12, 16}) {
      auto t0 = clk::now();
      for (int rep = 0; rep < 4; ++rep) {
        std::vector<std::thread> th;
        const uint64_t per = src.size() / threads;
        for (int t = 0; t < threads; ++t) th.emplace_back([&, t] { std::memcpy(fg + (uint64_t)(rep & 1) * src.size() + per * t, src.data() + per * t, per); });
        for (auto& x : th) x.join();
      }
      std::printf("host threads memcpy into device memory (flag %u), %2d threads: %6.1f GB/s\n", flag, threads, 4.0 * src.size() / ms_since(t0) / 1e6);
    }
    if (fd >= 0 && fsize >= (1ull << 30)) {
      for (int threads : {8, 12, 16}) {
        auto t0 = clk::now();
        for (int rep = 0; rep < 4; ++rep) fill_piece(fd, (uint64_t)rep * (256ull << 20), fg + (uint64_t)(rep & 1) * (256ull << 20), 256ull << 20, threads);
        std::printf("host threads pread into device memory (flag %u), %2d threads:  %6.1f GB/s\n", flag, threads, 4.0 * (256ull << 20) / ms_since(t0) / 1e6);
      }
    }
    // the bytes did arrive: a device-side copy back to the host buffer's twin
    std::vector<uint8_t> back(1 << 20);
    CK(hipMemcpy(back.data(), fg, back.size(), hipMemcpyDeviceToHost));
    std::printf("  (first bytes read back through the device: %u %u %u)\n", back[0], back[1], back[4095]);
    CK(hipFree(fg));
  }
  struct Kind { const char* name; unsigned flags; };
  const Kind kinds[] = {{"default", hipHostMallocDefault}, {"write-combined", hipHostMallocWriteCombined}, {"non-coherent", hipHostMallocNonCoherent}, {"numa-user", hipHostMallocNumaUser}};
  for (const Kind& k : kinds) {
    uint8_t* pin = nullptr;
    auto ta = clk::now();
    if (hipHostMalloc(reinterpret_cast<void**>(&pin), 256ull << 20, k.flags) != hipSuccess) { (void)hipGetLastError(); std::printf("%-15s hipHostMalloc refused\n", k.name); continue; }
    const double alloc_ms = ms_since(ta);
    std::memset(pin, 1, 256ull << 20);
    for (uint64_t piece : {16ull << 20, 64ull << 20, 128ull << 20}) {
      const int reps = (int)((2ull << 30) / piece);
      // (a) DMA alone
      CK(hipEventRecord(e0, s));
      for (int i = 0; i < reps; ++i) CK(hipMemcpyAsync(d + (uint64_t)(i & 1) * piece, pin + (uint64_t)(i & 1) * piece, piece, hipMemcpyHostToDevice, s));
      CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
      float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
      std::printf("%-15s piece %4llu MiB  DMA alone            %6.1f GB/s", k.name, (unsigned long long)(piece >> 20), (double)piece * reps / ms / 1e6);
      // (b) the path's pattern: fill half b by pread while half b^1 crosses
      if (fd >= 0 && fsize >= (1ull << 30)) {
        for (int threads : {12, 6}) {
          auto t0 = clk::now();
          double fill_ms = 0;
          for (int i = 0; i < reps; ++i) {
            const int b = i & 1;
            if (i >= 2) CK(hipEventSynchronize(ev[b]));
            auto tf = clk::now();
            fill_piece(fd, ((uint64_t)i * piece) % (fsize - piece), pin + (uint64_t)b * piece, piece, threads);
            fill_ms += ms_since(tf);
            CK(hipMemcpyAsync(d + (uint64_t)b * piece, pin + (uint64_t)b * piece, piece, hipMemcpyHostToDevice, s));
            CK(hipEventRecord(ev[b], s));
          }
          CK(hipStreamSynchronize(s));
          const double w = ms_since(t0);
          std::printf("   fill(%2d thr)+DMA %5.1f GB/s (fill alone would be %5.1f)", threads, (double)piece * reps / w / 1e6, (double)piece * reps / fill_ms / 1e6);
        }
      }
      std::printf("\n");
    }
    // (d) a kernel pulls the pinned bytes over PCIe
    {
      const uint64_t n = 128ull << 20;
      for (int wg : {64, 256, 1024}) {
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(k_pull, dim3(wg), dim3(256), 0, s, (const uint4*)pin, (uint4*)d, n / 16);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        std::printf("%-15s kernel pulls 128 MiB x 8 with %4d workgroups: %6.1f GB/s\n", k.name, wg, (double)n * 8 / ms / 1e6);
      }
    }
    std::printf("%-15s (hipHostMalloc of 256 MiB took %.1f ms)\n", k.name, alloc_ms);
    CK(hipHostFree(pin));
  }
  return 0;
}
