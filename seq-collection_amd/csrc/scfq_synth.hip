// scfq_synth.hip — the synthetic FASTQ workloads of SURVEY.md §8(d) / BASELINE.json configs 2,3,5.
//
// The reference ships only tiny fixtures (tests/fastq/*, <= 1.4 KB); its headline configurations are
// synthetic, so the generator is part of the measurement harness, not of the counting path.
// Record i is a pure, integer-only function of (seed, i): the host code and the HIP kernel below are the
// same __host__ __device__ functions, so a 10 GB image can be produced directly in HBM and any slice of it
// reproduced on the host for the oracle.  Record shape of the Illumina workload follows the reference's
// tests/fastq/novaseq.fq:1-4 (NovaSeq header, '+' separator line, 4-valued binned qualities).
#include "../../include/sc_fqcount.h"

#include <hip/hip_runtime.h>
#include <cstring>        // (rocprim's texture iterator calls memset from host code)
#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

namespace synth {

#define HD __host__ __device__ __forceinline__

HD uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  uint64_t z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
HD uint64_t field_hash(uint64_t seed, uint64_t rec, uint64_t field) {
  return splitmix64(seed ^ (rec * 0x9E3779B97F4A7C15ull) ^ (field * 0xD1B54A32D192ED03ull));
}
HD uint32_t slice16(uint64_t h, unsigned k) { return (uint32_t)(h >> (16 * (k & 3))) & 0xFFFFu; }

HD int put_dec(char* dst, uint64_t v) {
  char tmp[24];
  int n = 0;
  do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
  for (int k = 0; k < n; ++k) dst[k] = tmp[n - 1 - k];
  return n;
}
HD int put_hex16(char* dst, uint64_t v) {
  for (int k = 0; k < 16; ++k) { unsigned d = (unsigned)(v >> (60 - 4 * k)) & 15u; dst[k] = (char)(d < 10 ? '0' + d : 'a' + d - 10); }
  return 16;
}
HD int put_str(char* dst, const char* s) { int n = 0; while (s[n]) { dst[n] = s[n]; ++n; } return n; }

constexpr int kMaxHeader = 160;

struct RecordShape {
  uint32_t header_len;   // including the '@' and the trailing '\n'
  uint32_t read_len;
};

// ---- Illumina 150 bp (config 2/3) ---------------------------------------------------------------
HD RecordShape illumina_header(uint64_t seed, uint64_t rec, char* hdr /* kMaxHeader or nullptr */) {
  const uint64_t h = field_hash(seed, rec, 0);
  const uint64_t lane = 1 + (h & 3), tile = 1101 + ((h >> 8) % 1578), x = 1000 + ((h >> 24) % 32000),
                 y = 1000 + ((h >> 44) % 49000);
  char local[kMaxHeader];
  char* p = hdr ? hdr : local;
  int n = put_str(p, "@A00156:217:HKJWGDSXX:");
  n += put_dec(p + n, lane); p[n++] = ':';
  n += put_dec(p + n, tile); p[n++] = ':';
  n += put_dec(p + n, x); p[n++] = ':';
  n += put_dec(p + n, y);
  n += put_str(p + n, " 1:N:0:AACGCTTA");
  p[n++] = '\n';
  return RecordShape{(uint32_t)n, 150u};
}
HD char illumina_base(uint64_t seed, uint64_t rec, uint32_t k) {
  const uint32_t v = slice16(field_hash(seed, rec, 1 + (k >> 2)), k);
  if (v < 131) return 'N';                      // p = 0.002
  const uint32_t w = v - 131;                   // 0 .. 65404 ; P(A,C,G,T) = (.295,.205,.205,.295)
  return w < 19294 ? 'A' : (w < 32702 ? 'C' : (w < 46110 ? 'G' : 'T'));
}
HD char illumina_qual(uint64_t seed, uint64_t rec, uint32_t k) {
  const uint32_t v = slice16(field_hash(seed, rec, 64 + (k >> 2)), k);
  return v < 58982 ? 'F' : (v < 62915 ? ':' : (v < 64881 ? ',' : '#'));   // .90 .06 .03 .01
}

// ---- Nanopore-style 500 bp .. 50 kb (config 5) ----------------------------------------------------
HD uint32_t nanopore_len(uint64_t h) {
  // log-uniform in [500, 50000): 500 * 2^(u * log2(100)), integer fixed point only
  const uint32_t u = (uint32_t)(h & 0xFFFF);
  const uint32_t e = (uint32_t)(((uint64_t)u * 435411u) >> 16);   // 16.16, in [0, 6.6439)
  const uint32_t k = e >> 16, f = e & 0xFFFF;
  const uint32_t m = 65536u + (uint32_t)(((uint64_t)f * (43011u + ((f * 22525u) >> 16))) >> 16);   // ~2^f, 16.16
  uint64_t L = ((uint64_t)500 * m << k) >> 16;
  if (L < 500) L = 500;
  if (L > 50000) L = 50000;
  return (uint32_t)L;
}
HD RecordShape nanopore_header(uint64_t seed, uint64_t rec, char* hdr) {
  const uint64_t h = field_hash(seed, rec, 0);
  char local[kMaxHeader];
  char* p = hdr ? hdr : local;
  int n = 0;
  p[n++] = '@';
  n += put_hex16(p + n, field_hash(seed, rec, 1));
  n += put_hex16(p + n, field_hash(seed, rec, 2));
  n += put_str(p + n, " runid=");
  n += put_hex16(p + n, field_hash(seed, ~0ull, 3));
  n += put_hex16(p + n, field_hash(seed, ~0ull, 4));
  n += put_hex16(p + n, field_hash(seed, ~0ull, 5)) - 8;   // 40 hex digits
  n += put_str(p + n, " read=");
  n += put_dec(p + n, rec);
  n += put_str(p + n, " ch=");
  n += put_dec(p + n, 1 + ((h >> 32) % 512));
  n += put_str(p + n, " start_time=2026-01-01T00:00:00Z");
  p[n++] = '\n';
  return RecordShape{(uint32_t)n, nanopore_len(h >> 16)};
}
HD char nanopore_base(uint64_t seed, uint64_t rec, uint32_t k) {
  const uint32_t v = slice16(field_hash(seed, rec, 16 + (k >> 2)), k);
  if (v < 7) return 'N';                        // p ~ 1e-4
  const uint32_t w = v - 7;                     // GC = 0.45
  return w < 18020 ? 'A' : (w < 32764 ? 'C' : (w < 47508 ? 'G' : 'T'));
}
HD char nanopore_qual(uint64_t seed, uint64_t rec, uint32_t k) {
  const uint32_t v = slice16(field_hash(seed, rec, 0x100000u + (k >> 2)), k);
  return (char)(34 + ((v * 40u) >> 16));         // uniform over bytes 34..73
}

HD RecordShape header(int kind, uint64_t seed, uint64_t rec, char* hdr) {
  return kind == SCFQ_SYNTH_NANOPORE ? nanopore_header(seed, rec, hdr) : illumina_header(seed, rec, hdr);
}
HD uint64_t record_bytes(const RecordShape& s) { return (uint64_t)s.header_len + 2ull * s.read_len + 4; }  // seq\n +\n qual\n
HD char base_at(int kind, uint64_t seed, uint64_t rec, uint32_t k) {
  return kind == SCFQ_SYNTH_NANOPORE ? nanopore_base(seed, rec, k) : illumina_base(seed, rec, k);
}
HD char qual_at(int kind, uint64_t seed, uint64_t rec, uint32_t k) {
  return kind == SCFQ_SYNTH_NANOPORE ? nanopore_qual(seed, rec, k) : illumina_qual(seed, rec, k);
}

// ---- device kernels -----------------------------------------------------------------------------------
__global__ void k_lengths(int kind, uint64_t seed, uint64_t first, uint64_t records, uint64_t* len) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= records) return;
  len[i] = record_bytes(header(kind, seed, first + i, nullptr));
}

// one wave per record: lane 0 formats the header into LDS, all lanes then emit bytes 64 at a time
__global__ __launch_bounds__(256) void k_write(int kind, uint64_t seed, uint64_t first, uint64_t records,
                                               const uint64_t* offset, uint8_t* dst, unsigned long long* tally) {
  __shared__ char hdr[4][kMaxHeader];
  __shared__ RecordShape shp[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __shared__ uint32_t tal[4][3];
  const uint64_t i = (uint64_t)blockIdx.x * 4 + w;
  if (lane < 3) tal[w][lane] = 0;
  if (i < records) {      // (wave-uniform; every wave of the block reaches the barrier below)
    const uint64_t rec = first + i;
    if (lane == 0) shp[w] = header(kind, seed, rec, hdr[w]);
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    const RecordShape s = shp[w];
    uint8_t* out = dst + offset[i];
    for (uint32_t k = lane; k < s.header_len; k += 64) out[k] = (uint8_t)hdr[w][k];
    uint8_t* seq = out + s.header_len;
    uint8_t* qual = seq + s.read_len + 3;
    uint32_t gc = 0, nn = 0;
    for (uint32_t k = lane; k < s.read_len; k += 64) {
      const char b = base_at(kind, seed, rec, k);
      seq[k] = (uint8_t)b;
      gc += (b == 'G' || b == 'C');
      nn += (b == 'N');
      qual[k] = (uint8_t)qual_at(kind, seed, rec, k);
    }
    if (lane == 0) {
      seq[s.read_len] = '\n'; seq[s.read_len + 1] = '+'; seq[s.read_len + 2] = '\n';
      qual[s.read_len] = '\n';
    }
    for (int off = 32; off; off >>= 1) { gc += __shfl_down(gc, off); nn += __shfl_down(nn, off); }
    // the tallies of the block's four records through LDS, then ONE atomic per counter and block (an atomic per record and counter
    // queued 1.7 G of them on three words for the 200 GB stream)
    if (lane == 0) { tal[w][0] = gc; tal[w][1] = nn; tal[w][2] = s.read_len; }
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const unsigned long long t = (unsigned long long)tal[0][threadIdx.x] + tal[1][threadIdx.x] + tal[2][threadIdx.x] + tal[3][threadIdx.x];
    if (t) atomicAdd(&tally[threadIdx.x], t);
  }
}

// ---- host generation ----------------------------------------------------------------------------------
void host_write_record(int kind, uint64_t seed, uint64_t rec, uint8_t* out, uint64_t* gc, uint64_t* nn, uint64_t* bases) {
  char hdr[kMaxHeader];
  const RecordShape s = header(kind, seed, rec, hdr);
  std::memcpy(out, hdr, s.header_len);
  uint8_t* seq = out + s.header_len;
  uint8_t* qual = seq + s.read_len + 3;
  for (uint32_t k = 0; k < s.read_len; ++k) {
    const char b = base_at(kind, seed, rec, k);
    seq[k] = (uint8_t)b;
    *gc += (b == 'G' || b == 'C');
    *nn += (b == 'N');
    qual[k] = (uint8_t)qual_at(kind, seed, rec, k);
  }
  seq[s.read_len] = '\n'; seq[s.read_len + 1] = '+'; seq[s.read_len + 2] = '\n';
  qual[s.read_len] = '\n';
  *bases += s.read_len;
}

}  // namespace synth

extern "C" {

int scfq_synth_plan(int kind, uint64_t seed, uint64_t first_record, uint64_t min_bytes, scfq_synth_info* info) {
  if (!info || info->struct_size != sizeof(scfq_synth_info) || (kind != 0 && kind != 1)) return SCFQ_EARG;
  uint64_t bytes = 0, n = 0;
  while (bytes < min_bytes) {
    bytes += synth::record_bytes(synth::header(kind, seed, first_record + n, nullptr));
    ++n;
  }
  std::memset(info, 0, sizeof(*info));
  info->struct_size = sizeof(*info);
  info->records = n;
  info->bytes = bytes;
  return SCFQ_OK;
}

int scfq_synth_locate(int kind, uint64_t seed, uint64_t offset, uint64_t* record, uint64_t* record_start) {
  if (!record || !record_start || (kind != 0 && kind != 1)) return SCFQ_EARG;
  uint64_t pos = 0, i = 0;
  for (;;) {
    const uint64_t len = synth::record_bytes(synth::header(kind, seed, i, nullptr));
    if (offset < pos + len) break;
    pos += len;
    ++i;
  }
  *record = i;
  *record_start = pos;
  return SCFQ_OK;
}

int scfq_synth_host(int kind, uint64_t seed, uint64_t first_record, uint64_t records, void* dst, uint64_t cap,
                    scfq_synth_info* info) {
  if (!info || info->struct_size != sizeof(scfq_synth_info) || (kind != 0 && kind != 1)) return SCFQ_EARG;
  unsigned nt = std::max(1u, std::min(std::thread::hardware_concurrency(), 32u));
  if (records < 4096) nt = 1;
  std::vector<uint64_t> part_bytes(nt, 0), part_lo(nt + 1, 0);
  for (unsigned t = 0; t <= nt; ++t) part_lo[t] = records * t / nt;
  {
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t)
      th.emplace_back([&, t] {
        uint64_t b = 0;
        for (uint64_t i = part_lo[t]; i < part_lo[t + 1]; ++i)
          b += synth::record_bytes(synth::header(kind, seed, first_record + i, nullptr));
        part_bytes[t] = b;
      });
    for (auto& x : th) x.join();
  }
  uint64_t total = 0;
  std::vector<uint64_t> part_off(nt, 0);
  for (unsigned t = 0; t < nt; ++t) { part_off[t] = total; total += part_bytes[t]; }
  std::memset(info, 0, sizeof(*info));
  info->struct_size = sizeof(*info);
  info->records = records;
  info->bytes = total;
  if (!dst) return SCFQ_OK;           // size query
  if (cap < total) return SCFQ_EARG;
  std::vector<uint64_t> gc(nt, 0), nn(nt, 0), bs(nt, 0);
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; ++t)
    th.emplace_back([&, t] {
      uint8_t* out = static_cast<uint8_t*>(dst) + part_off[t];
      for (uint64_t i = part_lo[t]; i < part_lo[t + 1]; ++i) {
        const uint64_t rec = first_record + i;
        synth::host_write_record(kind, seed, rec, out, &gc[t], &nn[t], &bs[t]);
        out += synth::record_bytes(synth::header(kind, seed, rec, nullptr));
      }
    });
  for (auto& x : th) x.join();
  for (unsigned t = 0; t < nt; ++t) { info->gc_bases += gc[t]; info->n_bases += nn[t]; info->bases += bs[t]; }
  return SCFQ_OK;
}

int scfq_synth_device(int kind, uint64_t seed, uint64_t first_record, uint64_t records, void* dst_device,
                      uint64_t cap, scfq_synth_info* info) {
  if (!info || info->struct_size != sizeof(scfq_synth_info) || (kind != 0 && kind != 1) || !dst_device) return SCFQ_EARG;
  std::memset(info, 0, sizeof(*info));
  info->struct_size = sizeof(*info);
  info->records = records;
  if (!records) return SCFQ_OK;
#define SYN_HIP(call) do { if ((call) != hipSuccess) { rc = SCFQ_EHIP; goto done; } } while (0)
  int rc = SCFQ_OK;
  uint64_t *d_len = nullptr, *d_off = nullptr;
  unsigned long long* d_tally = nullptr;
  void* d_tmp = nullptr;
  size_t tmp_bytes = 0;
  uint64_t last_len = 0, last_off = 0;
  unsigned long long tally[3] = {0, 0, 0};
  SYN_HIP(hipMalloc(&d_len, records * sizeof(uint64_t)));
  SYN_HIP(hipMalloc(&d_off, records * sizeof(uint64_t)));
  SYN_HIP(hipMalloc(&d_tally, 3 * sizeof(unsigned long long)));
  SYN_HIP(hipMemset(d_tally, 0, 3 * sizeof(unsigned long long)));
  hipLaunchKernelGGL(synth::k_lengths, dim3((unsigned)((records + 255) / 256)), dim3(256), 0, 0, kind, seed,
                     first_record, records, d_len);
  SYN_HIP(hipGetLastError());
  SYN_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, d_len, d_off, (uint64_t)0, (size_t)records, rocprim::plus<uint64_t>()));
  SYN_HIP(hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 16));
  SYN_HIP(rocprim::exclusive_scan(d_tmp, tmp_bytes, d_len, d_off, (uint64_t)0, (size_t)records, rocprim::plus<uint64_t>()));
  SYN_HIP(hipMemcpy(&last_len, d_len + records - 1, sizeof(uint64_t), hipMemcpyDeviceToHost));
  SYN_HIP(hipMemcpy(&last_off, d_off + records - 1, sizeof(uint64_t), hipMemcpyDeviceToHost));
  info->bytes = last_off + last_len;
  if (info->bytes > cap) { rc = SCFQ_EARG; goto done; }
  // (one wave per record; a launch is limited to 2^32 threads in total — 25 GB of 150 bp reads are 69.5 M waves = 4.4 G threads,
  // which the runtime wrapped to the low 32 bits without an error — so the records go out in launches of 8 M)
  for (uint64_t r0 = 0; r0 < records; r0 += (8ull << 20)) {
    const uint64_t nr = std::min<uint64_t>(8ull << 20, records - r0);
    hipLaunchKernelGGL(synth::k_write, dim3((unsigned)((nr + 3) / 4)), dim3(256), 0, 0, kind, seed, first_record + r0,
                       nr, d_off + r0, static_cast<uint8_t*>(dst_device), d_tally);
    SYN_HIP(hipGetLastError());
  }
  SYN_HIP(hipDeviceSynchronize());
  SYN_HIP(hipMemcpy(tally, d_tally, sizeof tally, hipMemcpyDeviceToHost));
  info->gc_bases = tally[0];
  info->n_bases = tally[1];
  info->bases = tally[2];
done:
  if (d_len) (void)hipFree(d_len);
  if (d_off) (void)hipFree(d_off);
  if (d_tally) (void)hipFree(d_tally);
  if (d_tmp) (void)hipFree(d_tmp);
#undef SYN_HIP
  return rc;
}

}  // extern "C"
