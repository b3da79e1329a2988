// scfq_dedup.hip — `sc fq-dedup` on the MI355X (gfx950): de-duplicate a FASTQ by read ID, keep the first record of
// every ID (reference: src/fq_dedup.nim:14-84; CLI sc.nim:118-122).
//
// The reference streams the file twice through one thread: pass 1 fills a Bloom filter (1e8 capacity, :29) plus a
// CountTable of suspects, pass 2 echoes every record whose header was not met before.  The Bloom filter is only a
// pre-filter (a false positive is still echoed on its first occurrence), so stdout is exactly "drop each record whose
// header line equals an earlier header line".  Here the whole (inflated) input sits in HBM (288 GB per GPU) and that
// definition is computed directly, with no probabilistic structure:
//   K5  line index            (fq_scan_kernels.hpp: fq_index_pos + fq_index_expand_pos, one pass over the input; it also says whether the input
//                             holds "\r\n" line ends at all: without them no kernel below looks behind a newline)
//   D1  header hashes         of every header line (line 4i, EOL stripped as Nim readLine does; scfq_hdrhash.hpp): computed by the index pass itself
//                             for the headers it has whole in LDS (r4), by dd_hash_listed for the few it leaves (dd_hash_headers: all of them,
//                             when the index pass could not: its mask form, SCFQ_DEDUP_FUSED_HASH=0)
//   D2  radix sort            (hash, record) pairs, rocprim::radix_sort_pairs; stable, so equal hashes stay in file order
//   D3  dd_mark_duplicates    a record is a duplicate iff an EARLIER record of its equal-hash run has the same bytes:
//                             exact string compare, so hash collisions cost time, never correctness (dd_count_marks: the statistics,
//                             summed per block — an atomic per wave on one counter was half of D3's time)
//   D4  dd_record_lengths     bytes each kept record echoes: for each of its lines, text + '\n' ("\r\n" comes out as "\n",
//                             a final line without '\n' gains one: `echo record`, fq_dedup.nim:59,67,71), summed per group of 32 records
//   D5  exclusive scan        output offset of every GROUP (rocprim::exclusive_scan)
//   D6  dd_gather             one wave per group: one contiguous copy, eight 16 B loads per lane in flight, when the group is kept verbatim
// Everything is integer / byte work bound by HBM traffic; there is no CPU fallback.
#include "../../include/sc_fqcount.h"
#include "scfq_hdrhash.hpp"

#include <cstring>        // (rocprim's texture iterator calls memset from host code)
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "scfq_index_aux.hpp"      // scfq_index_lines_ex2: the index, and the header hashes on its way
constexpr uint64_t kHashSeed = 0x5CF0DED0B1A5ull;

#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

namespace {

thread_local char g_derr[512] = "";

#define DCHK(call)                                                                                          \
  do {                                                                                                      \
    hipError_t e_ = (call);                                                                                 \
    if (e_ != hipSuccess) {                                                                                 \
      std::snprintf(g_derr, sizeof g_derr, "%s -> %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      if (std::getenv("SCFQ_VERBOSE")) std::fprintf(stderr, "scfq: %s\n", g_derr);                          \
      return SCFQ_EHIP;                                                                                     \
    }                                                                                                       \
  } while (0)

// Scratch comes from a stream-ordered memory pool OWNED BY THIS LIBRARY (one per device, release threshold = keep
// everything, destroyed by scfq_shutdown): after the first call a de-duplication allocates nothing from the driver
// (hipMalloc / hipFree of multi-GB buffers cost more than all kernels of the pipeline together), and the device's default
// pool — which belongs to the host application — is left as it was.
std::mutex g_pool_mu;
std::map<int, hipMemPool_t> g_pools;

int scratch_pool(hipMemPool_t* out) {
  int dev = 0;
  DCHK(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_pool_mu);
  auto it = g_pools.find(dev);
  if (it == g_pools.end()) {
    hipMemPoolProps props{};
    props.allocType = hipMemAllocationTypePinned;
    props.handleTypes = hipMemHandleTypeNone;
    props.location.type = hipMemLocationTypeDevice;
    props.location.id = dev;
    hipMemPool_t pool = nullptr;
    DCHK(hipMemPoolCreate(&pool, &props));
    uint64_t keep = UINT64_MAX;
    DCHK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep));
    it = g_pools.emplace(dev, pool).first;
  }
  *out = it->second;
  return SCFQ_OK;
}

// The call's private stream comes from a per-device list of idle ones and goes back to it (r4: creating and destroying a stream per
// call cost more host time than all the launches of the pipeline; scfq_shutdown destroys them).  A stream is only returned by a call
// that has waited for everything it put on it.
std::map<int, std::vector<hipStream_t>> g_idle_streams;      // under g_pool_mu

struct StreamLease {
  hipStream_t s = nullptr;
  int dev = -1;
  bool clean = false;            // set by the owner once nothing is pending on s: a stream with work in flight is destroyed instead
  int acquire() {
    DCHK(hipGetDevice(&dev));
    {
      std::lock_guard<std::mutex> lk(g_pool_mu);
      auto& v = g_idle_streams[dev];
      if (!v.empty()) { s = v.back(); v.pop_back(); return SCFQ_OK; }
    }
    DCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    return SCFQ_OK;
  }
  ~StreamLease() {
    if (!s) return;
    if (clean || hipStreamSynchronize(s) == hipSuccess) {
      std::lock_guard<std::mutex> lk(g_pool_mu);
      auto& v = g_idle_streams[dev];
      if (v.size() < 8) { v.push_back(s); return; }
    }
    (void)hipStreamDestroy(s);
  }
};

struct DevBuf {   // returns its memory to the pool on scope exit (stream-ordered)
  void* p = nullptr;
  hipStream_t s = nullptr;
  ~DevBuf() { if (p) (void)hipFreeAsync(p, s); }
  template <typename T> T* as() { return static_cast<T*>(p); }
  int alloc(size_t bytes, hipStream_t stream) {
    s = stream;
    hipMemPool_t pool;
    int rc = scratch_pool(&pool);
    if (rc) return rc;
    DCHK(hipMallocFromPoolAsync(&p, std::max<size_t>(bytes, 16), pool, stream));
    return SCFQ_OK;
  }
  void* release() { void* q = p; p = nullptr; return q; }
};

// the caller's stream (scfq_set_wait_stream) is ordered before this call's private stream
int wait_for_caller(hipStream_t stream) {
  int on = 0;
  void* ws = scfq_get_wait_stream(&on);
  if (!on) return SCFQ_OK;
  hipEvent_t ev;
  DCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  hipError_t e = hipEventRecord(ev, static_cast<hipStream_t>(ws));
  if (e == hipSuccess) e = hipStreamWaitEvent(stream, ev, 0);
  (void)hipEventDestroy(ev);
  DCHK(e);
  return SCFQ_OK;
}

// text of line j: [line_off[j], end) where end excludes the '\n' and a '\r' directly before a REAL '\n'
// (Nim 1.0.6 readLine; a final line without '\n' keeps a trailing '\r')
__device__ __forceinline__ void line_span(const uint8_t* base, uint64_t n, const uint64_t* line_off, uint64_t j,
                                          uint64_t& s, uint64_t& e, bool has_cr = true) {
  s = line_off[j];
  const uint64_t nlpos = line_off[j + 1] - 1;     // position of the (real or implied) '\n'
  e = nlpos;
  if (has_cr && nlpos < n && e > s && base[e - 1] == '\r') --e;      // (has_cr: kernel-uniform, from the index pass)
}

// Word k (bytes [8k, 8k + 8)) of the header that starts at s and is len bytes long, bytes past the header zeroed; addresses are
// clamped to stay inside the input.
__device__ __forceinline__ uint64_t load_head_word(const uint8_t* base, uint64_t n, uint64_t s, uint64_t len, uint32_t k) {
  uint64_t a = s + 8u * k;
  uint32_t shift = 0;
  if (a + 8 > n) { const uint64_t a2 = (n >= 8) ? n - 8 : 0; shift = (uint32_t)(a - a2) * 8; a = a2; }   // last bytes of the input
  uint64_t v = 0;
  if (8u * k < len && n >= 8) { __builtin_memcpy(&v, base + a, 8); v = (shift < 64) ? (v >> shift) : 0; }
  else if (8u * k < len) { for (uint64_t b = 0; b < 8 && a + b < n; ++b) v |= (uint64_t)base[a + b] << (8 * b); }
  const uint64_t valid = (len > 8u * k) ? len - 8u * k : 0;          // bytes of this word that belong to the header
  return (valid >= 8) ? v : (valid ? (v & ((1ull << (8 * valid)) - 1)) : 0);
}

// The first 64 bytes of a header as eight 64-bit words.  All eight loads are issued before any is used (one memory round trip
// instead of eight dependent ones: the compare kernel is latency-bound, every thread chases its own header).
__device__ __forceinline__ void load_head64(const uint8_t* base, uint64_t n, uint64_t s, uint64_t len, uint64_t w[8]) {
#pragma unroll
  for (int k = 0; k < 8; ++k) w[k] = load_head_word(base, n, s, len, (uint32_t)k);
}

// D1: one thread per header line, the first 64 bytes fetched COOPERATIVELY: the headers of a wave are ~360 B apart, so eight
// loads per thread touched 64 cache lines per instruction (512 line look-ups per wave for ~100 distinct lines).  Instead lane l
// fetches word l & 7 of header 8 j + (l >> 3) in step j — eight consecutive lanes read 64 consecutive bytes — and the words go
// through LDS back to the lane that owns the header.  Same hash as a per-thread fetch (scfq_hdrhash.hpp: a sum over the words).
// K: the type the sorted hash is kept in (uint32_t when SCFQ_DEDUP_HASH_BITS <= 32: 4 radix passes over 8-byte pairs; else uint64_t)
template <typename K>
__global__ __launch_bounds__(256) void dd_hash_headers(const uint8_t* base, uint64_t n, const uint64_t* line_off, uint64_t n_hdr,
                                                      uint64_t seed, uint32_t hash_bits, K* keys, uint32_t* idx, uint64_t* hdr, bool has_cr) {
  __shared__ uint64_t sh_s[4][64];
  __shared__ uint32_t sh_len[4][64];
  __shared__ uint64_t sh_w[4][64][9];                   // (9: the owner's eight 8-byte reads of consecutive lanes spread over the banks)
  const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const bool mine = i < n_hdr;
  uint64_t s = 0, e = 0;
  if (mine) line_span(base, n, line_off, 4 * i, s, e, has_cr);
  const uint64_t len = e - s;
  sh_s[wv][lane] = s;
  sh_len[wv][lane] = (uint32_t)(len < 64 ? len : 64);
  __syncthreads();
#pragma unroll
  for (uint32_t j = 0; j < 8; ++j) {
    const uint32_t h = 8u * j + (lane >> 3), k = lane & 7u;
    sh_w[wv][h][k] = load_head_word(base, n, sh_s[wv][h], sh_len[wv][h], k);      // (no load for a word past the header's end)
  }
  __syncthreads();
  if (!mine) return;
  hdr[i] = s | ((len < 0xFFFFFFull ? len : 0xFFFFFFull) << 40);      // start (40 bits: inputs up to 1 TiB) | length, saturated
  uint32_t A = 0, B = 0;
  const uint64_t n_words = (len + 7) / 8;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const uint64_t w = sh_w[wv][lane][k];
    if ((uint64_t)k < n_words) scfq_hdrhash::hh_word((uint32_t)w, (uint32_t)(w >> 32), (uint32_t)k, A, B);
  }
  for (uint64_t k = 8; k < n_words; ++k) {           // headers longer than 64 bytes: the rest, a word per step
    const uint64_t w = load_head_word(base, n, s, len, (uint32_t)k);
    scfq_hdrhash::hh_word((uint32_t)w, (uint32_t)(w >> 32), (uint32_t)k, A, B);
  }
  uint64_t h = scfq_hdrhash::hh_final(A, B, len, seed);
  if (hash_bits < 64) h &= (1ull << hash_bits) - 1;     // (32 by default; the tests force collisions with 4)
  keys[i] = (K)h;
  idx[i] = (uint32_t)i;
}

// D1b (the index pass has hashed most headers: fq_index_pos): the records it left with the all-ones key — headers that cross a tile
// of the index pass, very long ones, the input's first line — listed (2048 positions and ONE atomic per block, as dd_find_equal) ...
template <typename K>
__global__ __launch_bounds__(256) void dd_find_unknown(const K* keys, uint64_t n_hdr, uint32_t* list, uint32_t* n_list) {
  __shared__ uint32_t wave_cnt[8][4], block_base;
  const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint64_t base = (uint64_t)blockIdx.x * 2048;
  uint64_t bal[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const uint64_t p = base + 256u * j + threadIdx.x;
    const bool un = p < n_hdr && (p == 0 || keys[p] == (K)~(K)0);
    bal[j] = __builtin_amdgcn_ballot_w64(un);
    if (lane == 0) wave_cnt[j][w] = (uint32_t)__popcll(bal[j]);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t tot = 0;
    for (int j = 0; j < 8; ++j) for (int k = 0; k < 4; ++k) tot += wave_cnt[j][k];
    block_base = tot ? atomicAdd(n_list, tot) : 0u;
  }
  __syncthreads();
  uint32_t before = block_base;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    for (uint32_t k = 0; k < w; ++k) before += wave_cnt[j][k];
    if ((bal[j] >> lane) & 1ull) list[before + (uint32_t)__popcll(bal[j] & ((1ull << lane) - 1))] = (uint32_t)(base + 256u * j + threadIdx.x);
    for (uint32_t k = w; k < 4; ++k) before += wave_cnt[j][k];
  }
}

template <typename K>
__device__ __forceinline__ void hash_one(const uint8_t* base, uint64_t n, const uint64_t* line_off, uint64_t i, uint64_t n_hdr, uint64_t seed, uint32_t hash_bits,
                                         K* keys, uint32_t* idx, uint64_t* hdr, bool has_cr);

// ... and hashed, a thread each: all of a header's first eight words requested before any is used.  n_list == nullptr: the list is
// the index pass's own, four entries per tile, 0 = no record — and record 0, which it never lists, is thread 0's
template <typename K>
__global__ __launch_bounds__(256) void dd_hash_listed(const uint8_t* base, uint64_t n, const uint64_t* line_off, const uint32_t* list, const uint32_t* n_list,
                                                     uint64_t n_entries, uint64_t n_hdr, uint64_t seed, uint32_t hash_bits, K* keys, uint32_t* idx, uint64_t* hdr, bool has_cr) {
  const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (!n_list) {
    // a thread per TILE of the index pass: its four entries in one load, nearly always four zeros
    if (t > n_entries / 4) return;
    if (t == 0) { hash_one<K>(base, n, line_off, 0, n_hdr, seed, hash_bits, keys, idx, hdr, has_cr); return; }
    const uint4 q = reinterpret_cast<const uint4*>(list)[t - 1];
    if ((q.x | q.y | q.z | q.w) == 0) return;
    if (q.x) hash_one<K>(base, n, line_off, q.x, n_hdr, seed, hash_bits, keys, idx, hdr, has_cr);
    if (q.y) hash_one<K>(base, n, line_off, q.y, n_hdr, seed, hash_bits, keys, idx, hdr, has_cr);
    if (q.z) hash_one<K>(base, n, line_off, q.z, n_hdr, seed, hash_bits, keys, idx, hdr, has_cr);
    if (q.w) hash_one<K>(base, n, line_off, q.w, n_hdr, seed, hash_bits, keys, idx, hdr, has_cr);
    return;
  }
  if (t >= *n_list) return;
  hash_one<K>(base, n, line_off, list[t], n_hdr, seed, hash_bits, keys, idx, hdr, has_cr);
}

template <typename K>
__device__ __forceinline__ void hash_one(const uint8_t* base, uint64_t n, const uint64_t* line_off, uint64_t i, uint64_t n_hdr, uint64_t seed, uint32_t hash_bits,
                                         K* keys, uint32_t* idx, uint64_t* hdr, bool has_cr) {
  // (the index pass numbers the line BEHIND every newline: behind the input's last one there is none — "record" n_hdr of an input whose
  // line count is a multiple of four)
  if (i >= n_hdr) return;
  uint64_t s, e;
  line_span(base, n, line_off, 4 * i, s, e, has_cr);
  const uint64_t len = e - s;
  hdr[i] = s | ((len < 0xFFFFFFull ? len : 0xFFFFFFull) << 40);
  uint64_t w[8];
  load_head64(base, n, s, len, w);
  uint32_t A = 0, B = 0;
  const uint64_t n_words = (len + 7) / 8;
#pragma unroll
  for (int k = 0; k < 8; ++k) if ((uint64_t)k < n_words) scfq_hdrhash::hh_word((uint32_t)w[k], (uint32_t)(w[k] >> 32), (uint32_t)k, A, B);
  for (uint64_t k = 8; k < n_words; ++k) {
    const uint64_t v = load_head_word(base, n, s, len, (uint32_t)k);
    scfq_hdrhash::hh_word((uint32_t)v, (uint32_t)(v >> 32), (uint32_t)k, A, B);
  }
  uint64_t h = scfq_hdrhash::hh_final(A, B, len, seed);
  if (hash_bits < 64) h &= (1ull << hash_bits) - 1;
  keys[i] = (K)h;
  idx[i] = (uint32_t)i;
}

// header i is bytes [start, start + length): (start, length) packed by dd_hash_headers; a saturated length (a 16 MiB header)
// is looked up again through the line index
__device__ __forceinline__ void header_span(const uint8_t* base, uint64_t n, const uint64_t* line_off, const uint64_t* hdr, uint64_t r, bool has_cr,
                                            uint64_t& s, uint64_t& len) {
  const uint64_t h = hdr[r];
  s = h & ((1ull << 40) - 1);
  len = h >> 40;
  if (len == 0xFFFFFFull) { uint64_t e; line_span(base, n, line_off, 4 * r, s, e, has_cr); len = e - s; }
}

__device__ __forceinline__ bool same_header(const uint8_t* base, uint64_t n, const uint64_t* line_off, const uint64_t* hdr, uint64_t ra, uint64_t rb,
                                            bool has_cr) {
  uint64_t sa, la, sb, lb;
  header_span(base, n, line_off, hdr, ra, has_cr, sa, la);
  header_span(base, n, line_off, hdr, rb, has_cr, sb, lb);
  if (la != lb) return false;
  const uint64_t len = la;
  uint64_t wa[8], wb[8];
  load_head64(base, n, sa, len, wa);
  load_head64(base, n, sb, len, wb);
  uint64_t diff = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) diff |= wa[k] ^ wb[k];
  if (diff) return false;
  for (uint64_t k = 64; k < len; ++k)
    if (base[sa + k] != base[sb + k]) return false;
  return true;
}

// D3a: the sorted positions whose hash equals their predecessor's (the only records that can be duplicates), compacted:
// with one record in six a duplicate, most lanes of the compare kernel had nothing to do while the others waited on memory
template <typename K>
__global__ __launch_bounds__(256) void dd_find_equal(const K* keys_sorted, uint64_t n_hdr, uint32_t* cand, uint32_t* n_cand) {
  // a block takes 2048 positions (8 per thread, position = base + 256 j + thread) and ONE atomic: with a block per 256 positions
  // the 109 K atomics on the one counter were the kernel's whole time (1.0 ms for 445 MB of keys)
  __shared__ uint32_t wave_cnt[8][4], block_base;
  const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint64_t base = (uint64_t)blockIdx.x * 2048;
  uint64_t bal[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const uint64_t p = base + 256u * j + threadIdx.x;
    const bool eq = p > 0 && p < n_hdr && keys_sorted[p] == keys_sorted[p - 1];
    bal[j] = __builtin_amdgcn_ballot_w64(eq);
    if (lane == 0) wave_cnt[j][w] = (uint32_t)__popcll(bal[j]);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t tot = 0;
    for (int j = 0; j < 8; ++j) for (int k = 0; k < 4; ++k) tot += wave_cnt[j][k];
    block_base = tot ? atomicAdd(n_cand, tot) : 0u;
  }
  __syncthreads();
  uint32_t before = block_base;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    for (uint32_t k = 0; k < w; ++k) before += wave_cnt[j][k];
    if ((bal[j] >> lane) & 1ull) cand[before + (uint32_t)__popcll(bal[j] & ((1ull << lane) - 1))] = (uint32_t)(base + 256u * j + threadIdx.x);
    for (uint32_t k = w; k < 4; ++k) before += wave_cnt[j][k];
  }
}

// D3b: candidate position p of the sorted order: is an EARLIER record of its equal-hash run the same string?  Normally the
// neighbour p-1 decides (either it is the same string, or the run is a genuine collision and is a handful of entries long).
// The first 64 bytes of both headers of every candidate of a wave are fetched cooperatively, as in dd_hash_headers (lane l:
// word l & 7 of header 8 j + (l >> 3), 16 steps for 128 headers) and compared from LDS; longer headers and longer runs take
// the per-thread walk.
template <typename K>
__global__ __launch_bounds__(256) void dd_mark_duplicates(const uint8_t* base, uint64_t n, const uint64_t* line_off, const uint64_t* hdr,
                                                         const K* keys_sorted, const uint32_t* idx_sorted, uint32_t* cand /* in: sorted position; out: collisions met */,
                                                         const uint32_t* n_cand, uint8_t* dup, bool has_cr) {
  __shared__ uint64_t sh_s[4][128];
  __shared__ uint32_t sh_len[4][128];
  __shared__ uint64_t sh_w[4][128][9];
  const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const uint32_t total = *n_cand;
  if ((uint64_t)blockIdx.x * 256 >= total) return;            // (block-uniform: the barriers below are reached by all or none)
  const bool active = t < total;
  uint64_t p = 0;
  K key = 0;
  uint32_t me = 0, other = 0;
  uint64_t sa = 0, la = 0, sb = 0, lb = 0;
  if (active) {
    p = cand[t];
    key = keys_sorted[p];
    me = idx_sorted[p];
    other = idx_sorted[p - 1];                                // (a candidate has p >= 1 and keys[p - 1] == keys[p])
    header_span(base, n, line_off, hdr, me, has_cr, sa, la);
    header_span(base, n, line_off, hdr, other, has_cr, sb, lb);
  }
  sh_s[wv][2 * lane] = sa; sh_len[wv][2 * lane] = (uint32_t)(la < 64 ? la : 64);
  sh_s[wv][2 * lane + 1] = sb; sh_len[wv][2 * lane + 1] = (uint32_t)(lb < 64 ? lb : 64);
  __syncthreads();
#pragma unroll
  for (uint32_t j = 0; j < 16; ++j) {
    const uint32_t h = 8u * j + (lane >> 3), k = lane & 7u;
    sh_w[wv][h][k] = load_head_word(base, n, sh_s[wv][h], sh_len[wv][h], k);
  }
  __syncthreads();
  if (!active) return;
  bool is_dup = false;
  uint32_t collided = 0;
  {
    uint64_t diff = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) diff |= sh_w[wv][2 * lane][k] ^ sh_w[wv][2 * lane + 1][k];
    bool same = (la == lb) && diff == 0;
    for (uint64_t k = 64; same && k < la; ++k) same = base[sa + k] == base[sb + k];
    if (same) is_dup = true; else collided = 1;
  }
  if (!is_dup) {                                              // a collision: the rest of the run, one by one
    for (uint64_t q = p - 1; q > 0 && keys_sorted[q - 1] == key; --q) {
      if (same_header(base, n, line_off, hdr, idx_sorted[q - 1], me, has_cr)) { is_dup = true; break; }
      ++collided;
    }
  }
  // dup[] was zeroed: only the duplicates pay a scattered byte store.  They are COUNTED by dd_count_marks afterwards: an atomic per
  // wave on the one counter here (72 K of them behind each other at the L2) was most of this kernel's time (r4)
  if (is_dup) dup[me] = 1;
  cand[t] = collided;                  // (this thread's own entry, read above: summed by dd_count_marks like the marks)
}

// the number of marked records and of collisions met: a block per slice of dup[] (16 bytes per load) and of the candidates' collision
// counts, ONE atomic per block and counter
__global__ __launch_bounds__(256) void dd_count_marks(const uint8_t* dup, uint64_t n_hdr, const uint32_t* coll, const uint32_t* n_cand,
                                                     unsigned long long* counters /* [0] dups, [1] hash collisions */) {
  __shared__ uint32_t sum_sh[2][4];
  const uint64_t per_block = ((n_hdr + gridDim.x - 1) / gridDim.x + 15) & ~15ull;       // (dup + a multiple of 16 is 16-byte aligned)
  const uint64_t lo = std::min<uint64_t>((uint64_t)blockIdx.x * per_block, n_hdr), hi = std::min<uint64_t>(lo + per_block, n_hdr);
  uint32_t c = 0, k = 0;
  for (uint64_t p = lo + 16ull * threadIdx.x; p < hi; p += 16ull * 256) {
    if (p + 16 <= hi) {
      const uint4 v = *reinterpret_cast<const uint4*>(dup + p);                          // flags are 0 or 1: a byte sum is a popcount
      c += (uint32_t)__builtin_popcount(v.x) + (uint32_t)__builtin_popcount(v.y) + (uint32_t)__builtin_popcount(v.z) + (uint32_t)__builtin_popcount(v.w);
    } else {
      for (uint64_t q = p; q < hi; ++q) c += dup[q];
    }
  }
  const uint64_t nc = *n_cand;
  for (uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x; t < nc; t += (uint64_t)gridDim.x * 256) k += coll[t];
  for (int off = 32; off > 0; off >>= 1) { c += (uint32_t)__shfl_down((int)c, off, 64); k += (uint32_t)__shfl_down((int)k, off, 64); }
  if ((threadIdx.x & 63) == 0) { sum_sh[0][threadIdx.x >> 6] = c; sum_sh[1][threadIdx.x >> 6] = k; }
  __syncthreads();
  if (threadIdx.x < 2) {
    const uint32_t t = sum_sh[threadIdx.x][0] + sum_sh[threadIdx.x][1] + sum_sh[threadIdx.x][2] + sum_sh[threadIdx.x][3];
    if (t) atomicAdd(&counters[threadIdx.x], (unsigned long long)t);
  }
}

// D6 works on groups of kGatherGroup consecutive records, one wave each
constexpr uint64_t kGatherGroup = 32;

// D4: bytes record i echoes (0 when dropped): its lines 4i .. min(4i+3, lines-1), each text + '\n' — and their sum over every group of
// kGatherGroup records: the output offsets are a prefix sum over the GROUPS (870 K of them for 28 M records: the scan over the records,
// its 0.19 ms and its two arrays went with that, r4), a record's own offset is found by the wave that gathers its group
__global__ __launch_bounds__(256) void dd_record_lengths(const uint8_t* base, uint64_t n, const uint64_t* line_off, uint64_t lines,
                                                        uint64_t n_hdr, const uint8_t* dup, uint64_t* out_len, uint64_t* group_len, bool has_cr) {
  static_assert(kGatherGroup == 32, "the sum below runs over the two halves of a wave");
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  uint64_t total = 0;
  if (i < n_hdr && !dup[i]) {
    if (!has_cr) {
      // no "\r\n" anywhere: a record echoes exactly its own bytes (a final line without '\n' gains the one its sentinel implies)
      const uint64_t j1 = (4 * i + 4 < lines) ? 4 * i + 4 : lines;
      total = line_off[j1] - line_off[4 * i];
    } else {
      for (uint64_t j = 4 * i; j < 4 * i + 4 && j < lines; ++j) {
        uint64_t s, e;
        line_span(base, n, line_off, j, s, e);
        total += e - s + 1;
      }
    }
  }
  if (i < n_hdr) out_len[i] = total;
  uint64_t sum = total;
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) {
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)sum, off, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)(sum >> 32), off, 64);
    sum += (uint64_t)lo | ((uint64_t)hi << 32);
  }
  if ((threadIdx.x & 31) == 0 && i < n_hdr) group_len[i / kGatherGroup] = sum;
}

// D6: one wave per group of kGatherGroup consecutive records.  When the whole group is kept and echoed verbatim (same
// number of bytes in and out: no '\r' stripped, no '\n' added, nothing dropped) it is ONE contiguous copy of ~11 KB;
// a group that is dropped entirely costs two loads.  Mixed groups go record by record, a record that is not verbatim line by line.
//
// The copy keeps EIGHT 16-byte loads per lane in flight before the first store.  Measured r4 and left out again: non-temporal stores
// (8 % slower here, though 2 % faster in the aligned copy micro-benchmark), the body aligned to 64 or 128 bytes of the destination
// instead of 16 (no difference): profiles/r04/dedup_ab.txt, copy_shapes.txt — the gather runs at what a copy reaches on this device.
__device__ __forceinline__ void wave_copy(uint8_t* dst, const uint8_t* src, uint64_t len, uint32_t lane) {
  typedef uint32_t v4u __attribute__((ext_vector_type(4)));
  uint64_t head = (16 - ((uintptr_t)dst & 15)) & 15;
  if (head > len) head = len;
  if (lane < head) dst[lane] = src[lane];
  const uint64_t body = (len - head) / 16;
  const uint8_t* sp = src + head;
  uint8_t* dp = dst + head;
  auto put = [&](const v4u& v, uint64_t k) { *reinterpret_cast<v4u*>(dp + k * 16) = v; };
  uint64_t k = lane;
  for (; k + 7 * 64 < body; k += 8 * 64) {
    v4u v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) __builtin_memcpy(&v[u], sp + (k + 64u * u) * 16, 16);       // the source is arbitrarily aligned
#pragma unroll
    for (int u = 0; u < 8; ++u) put(v[u], k + 64u * u);
  }
  if (k + 3 * 64 < body) {
    v4u v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) __builtin_memcpy(&v[u], sp + (k + 64u * u) * 16, 16);
#pragma unroll
    for (int u = 0; u < 4; ++u) put(v[u], k + 64u * u);
    k += 4 * 64;
  }
  {   // at most four steps are left: all their loads first as well
    v4u v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) if (k + 64u * u < body) __builtin_memcpy(&v[u], sp + (k + 64u * u) * 16, 16);
#pragma unroll
    for (int u = 0; u < 4; ++u) if (k + 64u * u < body) put(v[u], k + 64u * u);
  }
  for (uint64_t t = head + body * 16 + lane; t < len; t += 64) dst[t] = src[t];
}

__global__ __launch_bounds__(256) void dd_gather(const uint8_t* base, uint64_t n, const uint64_t* line_off, uint64_t lines,
                                                uint64_t n_hdr, const uint64_t* group_off, const uint64_t* out_len, uint8_t* out) {
  const uint64_t g = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t i0 = g * kGatherGroup;
  if (i0 >= n_hdr) return;
  const uint64_t i1 = (i0 + kGatherGroup < n_hdr) ? i0 + kGatherGroup : n_hdr;
  const uint64_t o0 = group_off[g], o1 = group_off[g + 1];    // group_off has one entry more than there are groups
  if (o1 == o0) return;                                       // every record of the group was dropped
  const uint64_t jl = (4 * i1 < lines) ? 4 * i1 : lines;
  const uint64_t s0 = line_off[4 * i0], s1 = line_off[jl];
  if (s1 - s0 == o1 - o0 && s1 <= n) { wave_copy(out + o0, base + s0, o1 - o0, lane); return; }
  uint64_t at = o0;                                            // where the next kept record of the group goes
  for (uint64_t i = i0; i < i1; ++i) {
    const uint64_t len = out_len[i];
    if (!len) continue;
    uint8_t* dst = out + at;
    at += len;
    const uint64_t j0 = 4 * i, j1 = (j0 + 4 < lines) ? j0 + 4 : lines;
    const uint64_t r0 = line_off[j0], r1 = line_off[j1];
    if (r1 - r0 == len && r1 <= n) { wave_copy(dst, base + r0, len, lane); continue; }
    uint64_t w = 0;
    for (uint64_t j = j0; j < j1; ++j) {
      uint64_t s, e;
      line_span(base, n, line_off, j, s, e);
      for (uint64_t k = lane; k < e - s; k += 64) dst[w + k] = base[s + k];
      if (lane == 0) dst[w + (e - s)] = '\n';
      w += e - s + 1;
    }
  }
}

// user_out / user_cap: device memory of the caller to gather into directly (nullptr: the result gets its own buffer, *d_out,
// which the caller returns with hipFreeAsync on `stream`); sized_only: stop after the statistics (out_bytes is set)
int dedup_device(const uint8_t* d_in, uint64_t n, uint8_t* user_out, uint64_t user_cap, bool sized_only, uint8_t** d_out,
                 uint64_t* out_bytes, scfq_dedup_stats* st, hipStream_t stream) {
  *d_out = nullptr;
  *out_bytes = 0;
  uint64_t lines = 0;
  // SCFQ_DEDUP_TRACE=1: host-clock stage times on stderr (each mark synchronises the stream: diagnostic only)
  // (=2: the same marks WITHOUT the synchronisation: what the host spends enqueueing each stage)
  static const int trace = [] { const char* e = std::getenv("SCFQ_DEDUP_TRACE"); return e ? std::max(1, std::atoi(e)) : 0; }();
  auto t_last = std::chrono::steady_clock::now();
  auto mark = [&](const char* what) {
    if (!trace) return;
    if (trace == 1) (void)hipStreamSynchronize(stream);
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "scfq dedup: %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
  };
  auto mark2 = [&](const char* what) { if (trace == 2) mark(what); };      // (enqueue-only marks)
  // the line index in ONE pass (count and offsets together): its size is guessed first — a FASTQ line is rarely shorter
  // than 24 bytes on average — and only a wrong guess costs a second pass with the exact size
  int rc = SCFQ_OK;
  uint32_t index_flags = 1;
  DevBuf line_off, keys, keys2, idx, idx2, dup, out_len, group_len, group_off, counters, tmp, hdr, cand, unk;
  // 32 bits of hash in 32-bit keys: the radix sort makes 4 passes over 8-byte (key, record) pairs — 0.93 ms for 28 M records against
  // 1.29 ms for 40 bits in 64-bit keys — and the exact compare behind the sort makes collisions (n^2 / 2^33 pairs: 88 K among 28 M
  // records) a matter of time, never of correctness.  (r2 - r3 kept 40 bits: the colliding pairs cost the compare kernel 0.48 ms —
  // which was the atomic each of them did on one statistics counter, not the compares; r4 sums the statistics per block.)
  static const int hash_bits_env = [] { const char* e = std::getenv("SCFQ_DEDUP_HASH_BITS"); return e ? std::min(64, std::max(1, std::atoi(e))) : 0; }();
  const uint32_t hash_bits = hash_bits_env ? (uint32_t)hash_bits_env : 32u;
  // The header hashes ride on the index pass (fq_index_pos has every header in LDS: scfq_hdrhash.hpp, SCFQ_DEDUP_FUSED_HASH=0 keeps the
  // hash kernel for all of them): keys, numbers and (start, length) are then sized by the index's guess, before the line count is known
  static const bool fused_env = [] { const char* e = std::getenv("SCFQ_DEDUP_FUSED_HASH"); return e ? std::atoi(e) != 0 : true; }();
  const bool fused = fused_env && hash_bits <= 56;
  const uint32_t key_bytes = hash_bits <= 32 ? 4u : 8u;
  scfq_index_aux aux{};
  {
    uint64_t cap = n / 24 + 1024;
    for (int round = 0; round < 2; ++round) {
      if ((rc = line_off.alloc(cap * 8, stream))) return rc;
      if (fused) {
        const uint64_t cap_records = cap / 4 + 2;
        if ((rc = keys.alloc(cap_records * 8, stream)) || (rc = idx.alloc(cap_records * 4, stream)) || (rc = hdr.alloc(cap_records * 8, stream))) return rc;
        aux.keys = keys.p; aux.idx = idx.as<uint32_t>(); aux.hdr = hdr.as<uint64_t>();
        aux.cap_records = cap_records; aux.key_bytes = key_bytes; aux.hash_bits = hash_bits; aux.seed = kHashSeed;
        if (!unk.p) {
          aux.unk_tiles = n / 4096 + 2;      // (the index pass's tiles are 4 KiB, the first one may be a partial one)
          if ((rc = unk.alloc(aux.unk_tiles * 4 * sizeof(uint32_t), stream))) return rc;
          aux.unk = unk.as<uint32_t>();
        }
      }
      mark("alloc line offsets");
      DCHK(hipStreamSynchronize(stream));       // scfq_index_lines works on the library's own stream
      mark("wait for the caller's stream");
      rc = scfq_index_lines_ex2(d_in, n, line_off.as<uint64_t>(), cap, &lines, &index_flags, fused ? &aux : nullptr);
      if (rc) return rc;
      if (lines + 1 <= cap) break;
      // the guess was too small (lines shorter than 24 bytes on average): once more with the exact size
      (void)hipFreeAsync(line_off.release(), stream);
      if (fused) { (void)hipFreeAsync(keys.release(), stream); (void)hipFreeAsync(idx.release(), stream); (void)hipFreeAsync(hdr.release(), stream); }
      cap = lines + 1;
    }
  }
  const bool has_cr = (index_flags & 1u) != 0;
  st->total_reads = lines / 4;                       // n_reads = i div 4      src/fq_dedup.nim:49
  const uint64_t n_hdr = (lines + 3) / 4;            // header lines: 0-based index i mod 4 == 0 (:43,57)
  if (n_hdr >= (1ull << 31)) { std::snprintf(g_derr, sizeof g_derr, "more than 2^31 records in one input"); return SCFQ_EARG; }
  if (n_hdr == 0) return SCFQ_OK;
  mark("line index (K5, one pass)");
  const uint64_t n_groups = (n_hdr + kGatherGroup - 1) / kGatherGroup;
  if (!fused && ((rc = keys.alloc(n_hdr * 8, stream)) || (rc = idx.alloc(n_hdr * 4, stream)) || (rc = hdr.alloc(n_hdr * 8, stream)))) return rc;
  if ((rc = keys2.alloc(n_hdr * 8, stream)) ||
      (rc = idx2.alloc(n_hdr * 4, stream)) || (rc = dup.alloc(n_hdr, stream)) || (rc = out_len.alloc(n_hdr * 8, stream)) ||
      (rc = group_len.alloc((n_groups + 1) * 8, stream)) || (rc = group_off.alloc((n_groups + 1) * 8, stream)) || (rc = counters.alloc(32, stream)) ||
      (rc = cand.alloc(n_hdr * 4, stream)))
    return rc;
  mark("alloc scratch");
  DCHK(hipMemsetAsync(counters.p, 0, 32, stream));
  const unsigned blocks = (unsigned)((n_hdr + 255) / 256);
  size_t scan_bytes = 0;
  DCHK(rocprim::exclusive_scan(nullptr, scan_bytes, group_len.as<uint64_t>(), group_off.as<uint64_t>(), (uint64_t)0, (size_t)(n_groups + 1),
                               rocprim::plus<uint64_t>(), stream));
  uint32_t* n_cand = reinterpret_cast<uint32_t*>(counters.as<unsigned long long>() + 2);
  auto hash_sort_mark = [&](auto key_tag) -> int {
    using K = decltype(key_tag);
    if (fused && aux.filled && aux.unk_complete) {
      // (the index pass left few — ~2 % of the records of a 150 bp file: the headers that cross one of its 4 KiB tiles — and listed them)
      const uint64_t n_entries = aux.n_tiles * 4;
      hipLaunchKernelGGL(dd_hash_listed<K>, dim3((unsigned)((aux.n_tiles + 1 + 255) / 256)), dim3(256), 0, stream, d_in, n, line_off.as<uint64_t>(), unk.as<uint32_t>(),
                         (const uint32_t*)nullptr, n_entries, n_hdr, kHashSeed, hash_bits, keys.as<K>(), idx.as<uint32_t>(), hdr.as<uint64_t>(), has_cr);
    } else if (fused && aux.filled) {
      // (its list is not complete — a tile with five of them, or with 64+ lines: short reads — they are looked for)
      uint32_t* n_list = reinterpret_cast<uint32_t*>(counters.as<unsigned long long>() + 3);
      hipLaunchKernelGGL(dd_find_unknown<K>, dim3((unsigned)((n_hdr + 2047) / 2048)), dim3(256), 0, stream, keys.as<K>(), n_hdr, cand.as<uint32_t>(), n_list);
      DCHK(hipGetLastError());
      hipLaunchKernelGGL(dd_hash_listed<K>, dim3(blocks), dim3(256), 0, stream, d_in, n, line_off.as<uint64_t>(), cand.as<uint32_t>(), n_list,
                         (uint64_t)0, n_hdr, kHashSeed, hash_bits, keys.as<K>(), idx.as<uint32_t>(), hdr.as<uint64_t>(), has_cr);
    } else {
      hipLaunchKernelGGL(dd_hash_headers<K>, dim3(blocks), dim3(256), 0, stream, d_in, n, line_off.as<uint64_t>(), n_hdr,
                         kHashSeed, hash_bits, keys.as<K>(), idx.as<uint32_t>(), hdr.as<uint64_t>(), has_cr);
    }
    DCHK(hipGetLastError());
    mark("hash headers");
    size_t tmp_bytes = 0;
    DCHK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys.as<K>(), keys2.as<K>(), idx.as<uint32_t>(), idx2.as<uint32_t>(), (size_t)n_hdr, 0u, hash_bits, stream));
    if ((rc = tmp.alloc(std::max(tmp_bytes, scan_bytes), stream))) return rc;
    DCHK(rocprim::radix_sort_pairs(tmp.p, tmp_bytes, keys.as<K>(), keys2.as<K>(), idx.as<uint32_t>(), idx2.as<uint32_t>(), (size_t)n_hdr, 0u, hash_bits, stream));
    mark("radix sort");
    DCHK(hipMemsetAsync(dup.p, 0, n_hdr, stream));
    hipLaunchKernelGGL(dd_find_equal<K>, dim3((unsigned)((n_hdr + 2047) / 2048)), dim3(256), 0, stream, keys2.as<K>(), n_hdr, cand.as<uint32_t>(), n_cand);
    DCHK(hipGetLastError());
    // (the launch covers the worst case; blocks past the candidate count leave at once)
    hipLaunchKernelGGL(dd_mark_duplicates<K>, dim3(blocks), dim3(256), 0, stream, d_in, n, line_off.as<uint64_t>(), hdr.as<uint64_t>(),
                       keys2.as<K>(), idx2.as<uint32_t>(), cand.as<uint32_t>(), n_cand, dup.as<uint8_t>(), has_cr);
    DCHK(hipGetLastError());
    hipLaunchKernelGGL(dd_count_marks, dim3(512), dim3(256), 0, stream, dup.as<uint8_t>(), n_hdr, cand.as<uint32_t>(), n_cand, counters.as<unsigned long long>());
    DCHK(hipGetLastError());
    return SCFQ_OK;
  };
  if ((rc = hash_bits <= 32 ? hash_sort_mark(uint32_t{}) : hash_sort_mark(uint64_t{}))) return rc;
  mark("mark duplicates");
  hipLaunchKernelGGL(dd_record_lengths, dim3(blocks), dim3(256), 0, stream, d_in, n, line_off.as<uint64_t>(), lines, n_hdr,
                     dup.as<uint8_t>(), out_len.as<uint64_t>(), group_len.as<uint64_t>(), has_cr);
  DCHK(hipGetLastError());
  DCHK(hipMemsetAsync(group_len.as<uint64_t>() + n_groups, 0, 8, stream));       // the scan's last output is the total
  DCHK(rocprim::exclusive_scan(tmp.p, scan_bytes, group_len.as<uint64_t>(), group_off.as<uint64_t>(), (uint64_t)0, (size_t)(n_groups + 1),
                               rocprim::plus<uint64_t>(), stream));
  mark2("lengths + scan enqueued");
  uint64_t h[3] = {0, 0, 0};
  // a caller's buffer that holds the whole input holds any result: the gather goes out behind the scan at once, and the one
  // wait of the call is the one at the end (otherwise the size has to come back first — the result is allocated to fit)
  const bool gather_first = !sized_only && user_out && user_cap >= n;
  if (gather_first) {
    hipLaunchKernelGGL(dd_gather, dim3((unsigned)((n_groups + 3) / 4)), dim3(256), 0, stream, d_in, n, line_off.as<uint64_t>(), lines, n_hdr,
                       group_off.as<uint64_t>(), out_len.as<uint64_t>(), user_out);
    DCHK(hipGetLastError());
  }
  mark2("gather enqueued");
  DCHK(hipMemcpyAsync(&h[0], group_off.as<uint64_t>() + n_groups, 8, hipMemcpyDeviceToHost, stream));
  DCHK(hipMemcpyAsync(&h[1], counters.p, 16, hipMemcpyDeviceToHost, stream));
  mark2("readbacks enqueued");
  DCHK(hipStreamSynchronize(stream));
  mark(gather_first ? "lengths + scan + gather + readback" : "lengths + scan + readback");
  st->duplicates = h[1];
  st->hash_collisions = h[2];
  st->records_out = n_hdr - h[1];
  st->bytes_out = h[0];
  st->false_positive = 0;
  *out_bytes = h[0];
  if (sized_only || h[0] == 0 || gather_first) return SCFQ_OK;
  DevBuf own;
  uint8_t* o = user_out;
  if (!o) {
    if ((rc = own.alloc(h[0], stream))) return rc;
    o = own.as<uint8_t>();
  } else if (user_cap < h[0]) {
    return SCFQ_EARG;
  }
  hipLaunchKernelGGL(dd_gather, dim3((unsigned)((n_groups + 3) / 4)), dim3(256), 0, stream, d_in, n, line_off.as<uint64_t>(), lines, n_hdr,
                     group_off.as<uint64_t>(), out_len.as<uint64_t>(), o);
  DCHK(hipGetLastError());
  DCHK(hipStreamSynchronize(stream));
  mark("gather");
  if (!user_out) *d_out = static_cast<uint8_t*>(own.release());
  return SCFQ_OK;
}

int write_all(int fd, const uint8_t* p, uint64_t n) {
  while (n) {
    ssize_t w = write(fd, p, (size_t)std::min<uint64_t>(n, 1u << 30));
    if (w < 0 && errno == EINTR) continue;
    if (w < 0) {       // only a vanished reader is the quiet case (sc.nim:304); a full disk or an I/O error is an error
      const int e = errno;
      std::snprintf(g_derr, sizeof g_derr, "write: %s", std::strerror(e));
      return e == EPIPE ? SCFQ_EPIPE : SCFQ_EIO;
    }
    p += w;
    n -= (uint64_t)w;
  }
  return SCFQ_OK;
}

}  // namespace

// called by scfq_shutdown(): the pools go back to the driver
extern "C" void scfq_dedup_release_pools(void) {
  std::lock_guard<std::mutex> lk(g_pool_mu);
  for (auto& kv : g_idle_streams) { if (hipSetDevice(kv.first) == hipSuccess) for (hipStream_t st : kv.second) (void)hipStreamDestroy(st); }
  g_idle_streams.clear();
  for (auto& kv : g_pools) { if (hipSetDevice(kv.first) == hipSuccess) (void)hipMemPoolDestroy(kv.second); }
  g_pools.clear();
}

extern "C" {

const char* scfq_dedup_error_detail(void) { return g_derr; }

int scfq_dedup_buffer(const void* ptr, uint64_t n, int is_device, void* out, uint64_t out_cap, int out_is_device,
                      uint64_t* out_bytes, scfq_dedup_stats* st) {
  if ((!ptr && n) || !st || st->struct_size != sizeof(scfq_dedup_stats) || !out_bytes) return SCFQ_EARG;
  const uint64_t keep_size = st->struct_size;
  std::memset(st, 0, sizeof(*st));
  st->struct_size = keep_size;
  *out_bytes = 0;
  StreamLease lease;
  { const int lrc = lease.acquire(); if (lrc) return lrc; }
  const hipStream_t stream = lease.s;
  DevBuf staged;
  const uint8_t* d_in = static_cast<const uint8_t*>(ptr);
  if (is_device || (out && out_is_device)) { int rc = wait_for_caller(stream); if (rc) return rc; }
  if (!is_device && n) {
    int rc = staged.alloc(n, stream);
    if (rc) return rc;
    DCHK(hipMemcpyAsync(staged.p, ptr, n, hipMemcpyHostToDevice, stream));
    DCHK(hipStreamSynchronize(stream));
    d_in = staged.as<uint8_t>();
  }
  uint8_t* d_out = nullptr;
  uint64_t nb = 0;
  // device destination: gather straight into the caller's buffer; host destination: gather into pool memory, copy down
  const bool direct = out && out_is_device;
  int rc = dedup_device(d_in, n, direct ? static_cast<uint8_t*>(out) : nullptr, direct ? out_cap : 0, /*sized_only=*/!out, &d_out, &nb, st, stream);
  *out_bytes = nb;
  if (rc) return rc;
  if (!out || direct || nb == 0) { if (d_out) (void)hipFreeAsync(d_out, stream); return SCFQ_OK; }
  struct OutGuard { void* p; hipStream_t s; ~OutGuard() { if (p) (void)hipFreeAsync(p, s); } } og{d_out, stream};
  if (nb > out_cap) return SCFQ_EARG;      // caller sizes (out = NULL) and calls again
  DCHK(hipMemcpyAsync(out, d_out, nb, hipMemcpyDeviceToHost, stream));
  DCHK(hipStreamSynchronize(stream));
  return SCFQ_OK;
}

int scfq_dedup_file(const char* path, const scfq_opts* opts, int out_fd, scfq_dedup_stats* st) {
  if (!path || !st || st->struct_size != sizeof(scfq_dedup_stats)) return SCFQ_EARG;
  const uint64_t keep_size = st->struct_size;
  std::memset(st, 0, sizeof(*st));
  st->struct_size = keep_size;
  void* d_in = nullptr;
  uint64_t n = 0;
  int rc = scfq_stage_file(path, opts, &d_in, &n);      // whole (inflated) input into HBM
  if (rc) return rc;
  struct InGuard { void* p; ~InGuard() { if (p) (void)hipFree(p); } } ig{d_in};
  StreamLease lease;
  { const int lrc = lease.acquire(); if (lrc) return lrc; }
  const hipStream_t stream = lease.s;
  uint8_t* d_out = nullptr;
  uint64_t nb = 0;
  rc = dedup_device(static_cast<const uint8_t*>(d_in), n, nullptr, 0, /*sized_only=*/out_fd < 0, &d_out, &nb, st, stream);
  if (rc) return rc;
  struct OutGuard { void* p; hipStream_t s; ~OutGuard() { if (p) (void)hipFreeAsync(p, s); } } og{d_out, stream};
  if (out_fd < 0 || nb == 0) return SCFQ_OK;
  // HBM -> two pinned buffers -> write(): the copy of chunk k+1 overlaps the write of chunk k
  const uint64_t chunk = 32ull << 20;
  uint8_t* pin[2] = {nullptr, nullptr};
  hipEvent_t ev[2] = {nullptr, nullptr};
  for (int b = 0; b < 2; ++b) { DCHK(hipHostMalloc(&pin[b], chunk, hipHostMallocDefault)); DCHK(hipEventCreateWithFlags(&ev[b], hipEventDisableTiming)); }
  struct PinGuard { uint8_t** p; hipEvent_t* e; ~PinGuard() { for (int b = 0; b < 2; ++b) { if (p[b]) (void)hipHostFree(p[b]); if (e[b]) (void)hipEventDestroy(e[b]); } } } pg{pin, ev};
  const uint64_t n_chunks = (nb + chunk - 1) / chunk;
  auto issue = [&](uint64_t k) -> hipError_t {
    const uint64_t lo = k * chunk, len = std::min(chunk, nb - lo);
    hipError_t e = hipMemcpyAsync(pin[k & 1], d_out + lo, len, hipMemcpyDeviceToHost, stream);
    return e != hipSuccess ? e : hipEventRecord(ev[k & 1], stream);
  };
  DCHK(issue(0));
  for (uint64_t k = 0; k < n_chunks; ++k) {
    DCHK(hipEventSynchronize(ev[k & 1]));
    if (k + 1 < n_chunks) DCHK(issue(k + 1));
    const uint64_t lo = k * chunk, len = std::min(chunk, nb - lo);
    if ((rc = write_all(out_fd, pin[k & 1], len))) return rc;
  }
  return SCFQ_OK;
}

}  // extern "C"
