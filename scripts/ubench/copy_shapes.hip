// Micro-benchmark behind DESIGN.md's fq-dedup gather row: what a device-to-device copy reaches on this GPU, by shape.
//   build: make -C seq-collection_amd ubench     run (GPU box): scripts/ubench/copy_shapes [GiB=8]
// Every variant moves the same bytes; the figure is reads + writes per second.
//   memcpy        hipMemcpyDtoDAsync
//   flat          one 16-byte load + store per thread, grid-stride, both sides 16-byte aligned
//   chunk<U>      one wave per CHUNK-byte piece (the gather's shape: 32 records = 11.5 KB), U loads in flight per lane before the first store
//   nt ...        the same with non-temporal loads / stores (aligned pieces only)
//   chunk, src+7  the same with an unaligned source (a record starts anywhere)
//   chunk, dst+16 destination pieces that start 16 bytes into a 128-byte line (what the output offsets of a gather look like)
//   lds           unaligned source through LDS: aligned 16-byte global loads, byte-shifted reads from LDS, aligned stores
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

typedef uint32_t v4u __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_flat(const uint8_t* src, uint8_t* dst, uint64_t n16) {
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256)
    reinterpret_cast<v4u*>(dst)[i] = reinterpret_cast<const v4u*>(src)[i];
}

template <int U, bool NTL = false, bool NTS = false>
__global__ __launch_bounds__(256) void k_chunk(const uint8_t* src, uint8_t* dst, uint64_t n_chunks, uint32_t chunk) {
  const uint64_t g = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g >= n_chunks) return;
  const uint32_t lane = threadIdx.x & 63;
  const uint8_t* sp = src + g * chunk;
  uint8_t* dp = dst + g * chunk;
  const uint32_t body = chunk / 16;
  uint32_t k = lane;
  for (; k + (U - 1) * 64 < body; k += U * 64) {
    v4u v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (NTL) v[u] = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(sp + (uint64_t)(k + 64u * u) * 16));
      else __builtin_memcpy(&v[u], sp + (uint64_t)(k + 64u * u) * 16, 16);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (NTS) __builtin_nontemporal_store(v[u], reinterpret_cast<v4u*>(dp + (uint64_t)(k + 64u * u) * 16));
      else *reinterpret_cast<v4u*>(dp + (uint64_t)(k + 64u * u) * 16) = v[u];
    }
  }
  for (; k < body; k += 64) {
    v4u v;
    __builtin_memcpy(&v, sp + (uint64_t)k * 16, 16);
    *reinterpret_cast<v4u*>(dp + (uint64_t)k * 16) = v;
  }
}

// unaligned source through LDS: the wave loads the aligned 16-byte words that cover its piece (one more than it stores), reads them
// back shifted by the source's misalignment and stores aligned
__global__ __launch_bounds__(256) void k_lds(const uint8_t* src, uint8_t* dst, uint64_t n_chunks, uint32_t chunk) {
  __shared__ __attribute__((aligned(16))) uint8_t sh[4][4 * 1024 + 32];
  const uint64_t g = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g >= n_chunks) return;
  const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint8_t* sp = src + g * chunk;
  uint8_t* dp = dst + g * chunk;
  const uint32_t mis = (uint32_t)((uintptr_t)sp & 15);
  const uint8_t* sa = sp - mis;
  for (uint32_t at = 0; at < chunk; at += 4096) {          // 4 KB of output per round: 257 aligned source words
    const uint32_t here = (chunk - at < 4096 ? chunk - at : 4096) / 16;
    v4u v[4], extra;
#pragma unroll
    for (int u = 0; u < 4; ++u) if (lane + 64u * u < here) v[u] = *reinterpret_cast<const v4u*>(sa + at + (lane + 64u * u) * 16);
    if (lane == 0) extra = *reinterpret_cast<const v4u*>(sa + at + here * 16);
#pragma unroll
    for (int u = 0; u < 4; ++u) if (lane + 64u * u < here) *reinterpret_cast<v4u*>(&sh[w][(lane + 64u * u) * 16]) = v[u];
    if (lane == 0) *reinterpret_cast<v4u*>(&sh[w][here * 16]) = extra;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (lane + 64u * u < here) {
        const uint8_t* q = &sh[w][(lane + 64u * u) * 16 + mis];
        uint64_t a, b;
        __builtin_memcpy(&a, q, 8);
        __builtin_memcpy(&b, q + 8, 8);
        v4u o = {(uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32)};
        *reinterpret_cast<v4u*>(dp + at + (lane + 64u * u) * 16) = o;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

int main(int argc, char** argv) {
  const uint64_t gib = argc > 1 ? (uint64_t)std::atoll(argv[1]) : 8;
  const uint32_t chunk = 11520;                      // 32 records of 360 bytes
  const uint64_t n_chunks = (gib << 30) / chunk;
  const uint64_t n = n_chunks * chunk;
  uint8_t *src, *dst;
  CHK(hipMalloc(&src, n + 4096));
  CHK(hipMalloc(&dst, n + 4096));
  CHK(hipMemset(src, 0x5A, n + 4096));
  CHK(hipMemset(dst, 0, n + 4096));
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  auto time_it = [&](const char* name, auto&& launch) {
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
      CHK(hipEventRecord(e0, nullptr));
      launch();
      CHK(hipEventRecord(e1, nullptr));
      CHK(hipEventSynchronize(e1));
      CHK(hipGetLastError());
      float ms = 0;
      CHK(hipEventElapsedTime(&ms, e0, e1));
      if (rep && ms < best) best = ms;
    }
    std::printf("%-28s %8.3f ms  %6.2f TB/s (reads + writes)\n", name, best, 2.0 * (double)n / best / 1e9);
  };
  const unsigned cb = (unsigned)((n_chunks + 3) / 4);
  time_it("memcpy", [&] { CHK(hipMemcpyDtoDAsync(dst, src, n, nullptr)); });
  time_it("flat, 8192 blocks", [&] { hipLaunchKernelGGL(k_flat, dim3(8192), dim3(256), 0, nullptr, src, dst, n / 16); });
  time_it("flat, 65536 blocks", [&] { hipLaunchKernelGGL(k_flat, dim3(65536), dim3(256), 0, nullptr, src, dst, n / 16); });
  time_it("chunk<1>", [&] { hipLaunchKernelGGL(k_chunk<1>, dim3(cb), dim3(256), 0, nullptr, src, dst, n_chunks, chunk); });
  time_it("chunk<4>", [&] { hipLaunchKernelGGL(k_chunk<4>, dim3(cb), dim3(256), 0, nullptr, src, dst, n_chunks, chunk); });
  time_it("chunk<8>", [&] { hipLaunchKernelGGL(k_chunk<8>, dim3(cb), dim3(256), 0, nullptr, src, dst, n_chunks, chunk); });
  time_it("chunk<12>", [&] { hipLaunchKernelGGL(k_chunk<12>, dim3(cb), dim3(256), 0, nullptr, src, dst, n_chunks, chunk); });
  time_it("chunk<8>, nt loads", [&] { hipLaunchKernelGGL((k_chunk<8, true, false>), dim3(cb), dim3(256), 0, nullptr, src, dst, n_chunks, chunk); });
  time_it("chunk<8>, nt stores", [&] { hipLaunchKernelGGL((k_chunk<8, false, true>), dim3(cb), dim3(256), 0, nullptr, src, dst, n_chunks, chunk); });
  time_it("chunk<8>, nt both", [&] { hipLaunchKernelGGL((k_chunk<8, true, true>), dim3(cb), dim3(256), 0, nullptr, src, dst, n_chunks, chunk); });
  time_it("chunk<8>, src+7", [&] { hipLaunchKernelGGL(k_chunk<8>, dim3(cb), dim3(256), 0, nullptr, src + 7, dst, n_chunks, chunk); });
  time_it("chunk<8>, dst+16", [&] { hipLaunchKernelGGL(k_chunk<8>, dim3(cb), dim3(256), 0, nullptr, src, dst + 16, n_chunks, chunk); });
  time_it("chunk<8>, src+7, dst+16", [&] { hipLaunchKernelGGL(k_chunk<8>, dim3(cb), dim3(256), 0, nullptr, src + 7, dst + 16, n_chunks, chunk); });
  time_it("lds, src+7, dst+16", [&] { hipLaunchKernelGGL(k_lds, dim3(cb), dim3(256), 0, nullptr, src + 16 + 7, dst + 16, n_chunks, chunk); });
  time_it("lds, aligned", [&] { hipLaunchKernelGGL(k_lds, dim3(cb), dim3(256), 0, nullptr, src, dst, n_chunks, chunk); });
  // verify the LDS form once
  CHK(hipMemset(dst, 0, n + 4096));
  std::vector<uint8_t> pat(1 << 20);
  for (size_t i = 0; i < pat.size(); ++i) pat[i] = (uint8_t)(i * 131 + (i >> 8));
  CHK(hipMemcpy(src, pat.data(), pat.size(), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_lds, dim3(16), dim3(256), 0, nullptr, src + 23, dst + 16, (uint64_t)64, chunk);
  std::vector<uint8_t> back(64 * chunk);
  CHK(hipMemcpy(back.data(), dst + 16, back.size(), hipMemcpyDeviceToHost));
  bool ok = true;
  for (size_t i = 0; i < back.size() && ok; ++i) ok = back[i] == pat[i + 23];
  std::printf("lds form copies correctly: %s\n", ok ? "yes" : "NO");
  return ok ? 0 : 1;
}
