// scfq_api.hip — device side of the C ABI (include/sc_fqcount.h): per-device context, kernel
// launches, host->HBM staging with copy/compute overlap, file and gzip ingest.
//
// Boundary replaced in the reference: src/fq_count.nim:30-45 (open stream, line loop, counters).
// gzip input keeps the reference's inflate semantics (zlib gzread: gzip_stream.nim:16-17 — multi-member
// streams, transparent pass-through of non-gzip bytes) but runs it on the host into pinned buffers that a
// copy stream moves to HBM while the compute stream scans the previous chunk.
//
// There is NO CPU fallback in this library: without a HIP device every counting entry point returns
// SCFQ_EHIP.
#include "../../include/sc_fqcount.h"
#include "../../include/sc_fqcount_debug.h"
#include "fq_scan_kernels.hpp"
#include "scfq_bgzf.hpp"
#include "scfq_gzfast.hpp"
#include "scfq_pgz.hpp"
#include "bgzf_inflate_kernel.hpp"
#include "gz_inflate_kernels.hpp"
#include "scfq_arena.hpp"
#include "scfq_index_aux.hpp"

#include <fcntl.h>
#include <functional>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local char g_err[512] = "";

#define HIPCHK(call)                                                                          \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      std::snprintf(g_err, sizeof g_err, "%s -> %s (%s:%d)", #call, hipGetErrorString(e_),    \
                    __FILE__, __LINE__);                                                      \
      if (std::getenv("SCFQ_VERBOSE")) std::fprintf(stderr, "scfq: %s\n", g_err);             \
      return SCFQ_EHIP;                                                                       \
    }                                                                                         \
  } while (0)

// Where a call's time goes, as milliseconds since the library was loaded (process start, near enough).  Every stage mark is kept (the
// first 256 of a process: a mark is a clock read and a push) and handed out by scfq_debug_stages() — `sc fq-count --stats` prints them,
// bench.py's cold legs carry them —, and SCFQ_VERBOSE=1 also writes each one to stderr as it happens.
const std::chrono::steady_clock::time_point g_loaded = std::chrono::steady_clock::now();
inline int env_int_early(const char* name, int dflt) { const char* e = std::getenv(name); return e ? std::atoi(e) : dflt; }
inline bool trace_on() { static const bool v = std::getenv("SCFQ_VERBOSE") != nullptr; return v; }
struct StageLog { std::mutex mu; std::vector<std::pair<std::string, double>> v; };
inline StageLog& stage_log() { static StageLog* g = new StageLog; return *g; }      // (leaked on purpose: marks may come from exit paths)
inline void trace(const char* what) {
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - g_loaded).count();
  {
    StageLog& g = stage_log();
    std::lock_guard<std::mutex> lk(g.mu);
    if (g.v.size() < 256) g.v.emplace_back(what, ms);
  }
  if (trace_on()) std::fprintf(stderr, "scfq t+%8.1f ms  %s\n", ms, what);
}
// device memory the library holds for ingest (staging, inflate buffers, arenas): current and high-water, per process
std::atomic<uint64_t> g_dev_bytes{0}, g_dev_high{0};
inline void note_dev_bytes(int64_t delta) {
  const uint64_t now = g_dev_bytes.fetch_add((uint64_t)delta) + (uint64_t)delta;
  uint64_t h = g_dev_high.load();
  while (now > h && !g_dev_high.compare_exchange_weak(h, now)) {}
}

constexpr uint64_t kDefaultChunk = 64ull << 20;
constexpr int kStateWords = SCFQ_PARTIAL_WORDS + SCFQ_HIST_WORDS + scfq::kExtWords;   // partial | hist[4][256] | ext (K3 speculation)
constexpr int kExtAt = SCFQ_PARTIAL_WORDS + SCFQ_HIST_WORDS;

using scfq_arena::DevBuf;
using scfq_arena::SymPool;
struct GzSlot {            // one batch being decoded: symbols + its segment tables (device and pinned mirror: slices of GzDevBuffers' tables)
  SymPool sym;                                            // 16-bit symbols of every segment of the batch, markers in front: chunks of 512 MB
  uint8_t* d_meta = nullptr;
  uint8_t* h_meta = nullptr;
};
// Kept between calls, sized by what the batches turn out to need, and never one allocation of tens of GB (scfq_arena.hpp: that was
// 1.5 - 2.6 s of every first call in round 2).
constexpr int kGzMaxComp = 8, kGzMaxSlots = 6, kGzDecodeStreams = 3;
struct GzDevBuffers {
  DevBuf comp[kGzMaxComp];                                // compressed bytes of the batches in flight: slots + 2 (copy and search run ahead of the decode)
  GzSlot slot[kGzMaxSlots];                               // symbols of the batches being decoded / resolved (SCFQ_GZ_DEVICE_SLOTS)
  uint8_t* d_tables = nullptr;  uint64_t tables_cap = 0;        // every table of all slots: one device allocation ...
  uint8_t* h_tables = nullptr;  uint64_t htables_cap = 0;       // ... and one pinned one (+ the tile CRCs' mirror)
  uint8_t* d_pmeta[kGzMaxSlots] = {};                     // chain, work lists, gap searches of a batch
  uint8_t* h_pmeta[kGzMaxSlots] = {};
  DevBuf out;                                             // inflated bytes of one batch (kStagePad in front)
  DevBuf win;                                             // the window in front of every chain entry of one batch
  uint8_t* d_wcarry = nullptr; uint64_t wcarry_cap = 0;  // the window that crosses a batch border (two, alternating)
  DevBuf maps;                                            // window maps of the chain groups of one batch (16-bit symbols)
  DevBuf gwin;                                            // the window in front of every group
  DevBuf crc;                                             // status words, then the tile CRCs of the whole file
  uint8_t* h_crc = nullptr;
  std::vector<void*> retired;                             // buffers replaced by bigger ones during a call: freed when it ends
  bool copies_warmed = false;                             // the search stream has carried its first copies (a process's first ones cost 17 ms)
  bool decode_warmed = false;                             // the decode kernel's first (empty) launch has set the device's scratch up
  hipStream_t s_search = nullptr, s_gap = nullptr, s_decode[kGzDecodeStreams] = {};
  hipEvent_t ev_copy[kGzMaxComp] = {}, ev_found[kGzMaxComp] = {}, ev_dec[kGzMaxSlots] = {}, ev_post[kGzMaxSlots] = {};
  uint8_t* d_search[kGzMaxComp] = {};                     // from | found of a batch's block-start search (slices of the tables)
  uint8_t* h_search[kGzMaxComp] = {};
  uint8_t* h_ring = nullptr; uint64_t ring_piece = 0;     // the engine's own pinned ring (two pieces) the compressed bytes cross in: ingest_gz_device_batches
  hipEvent_t ev_ring[2] = {};
  uint8_t* fg_stage[2] = {nullptr, nullptr}; uint64_t fg_cap = 0;      // SCFQ_GZ_DEVICE_HOST_WRITES: fine-grained DEVICE memory the host's threads write a batch's bytes into
  hipEvent_t ev_fg[2] = {};
  uint64_t held() const {
    uint64_t t = out.cap + win.cap + maps.cap + gwin.cap + crc.cap + wcarry_cap + tables_cap + (fg_stage[0] ? 2 * fg_cap : 0);
    for (int b = 0; b < kGzMaxComp; ++b) t += comp[b].cap;
    for (int b = 0; b < kGzMaxSlots; ++b) t += slot[b].sym.bytes();
    return t;
  }
};

// The device gzip path has a small POOL of engines per device (buffers, streams, events: GzDevBuffers), SCFQ_GZ_DEVICE_ENGINES = 4 of
// them at most, made as sessions need them.  Round 2 had one: a small file's pipeline is a chain — copy, block-start search, decode,
// walk, windows, bytes, scan — that leaves the device idle most of the time, so `sc fq-count --jobs=N *.fq.gz` (and the default
// loop, which keeps up to four files in flight) only paid for N contexts and then queued.  With buffers sized by what a file needs
// (scfq_gzdev.hpp) an engine for a 64 MB file holds about a GB, and file k + 1's copy and search run under file k's decode.
// Files of more than 1 GiB compressed take the device one at a time (`big`): their decode fills it anyway, and four of them would
// hold four times the memory.
struct GzShared {
  std::mutex mu;                       // guards the pool's bookkeeping
  std::condition_variable cv;
  static constexpr int kMax = 8;
  GzDevBuffers buf[kMax];
  bool busy[kMax] = {false, false, false, false, false, false, false, false};
  int n_busy = 0;
  bool big_running = false;
  int big_waiting = 0;                 // files of more than 1 GiB waiting for the pool to drain: no small file is admitted meanwhile
};
inline GzShared& gz_shared(int dev) { static GzShared g[64]; return g[dev & 63]; }

struct Ctx {
  int dev = -1;
  int n_cu = 256;
  hipStream_t compute = nullptr, copy = nullptr;
  uint64_t n_sessions = 0;              // sessions begun on this context so far
  bool bgzf_crc_consts = false;         // bgzf_crc_consts_init has been queued on this context's compute stream
  bool copy_is_alias = false;           // `copy` IS the compute stream (no second chunk has had to move yet): want_copy_stream()
  uint64_t* d_partials = nullptr;
  uint64_t cap_ranges = 0;
  uint32_t* d_hist_partials = nullptr;
  uint8_t* d_range_phase = nullptr;   // [cap] rel phase per range, then [cap] start phase per level-1 block
  uint64_t cap_hist_ranges = 0;
  // K3 speculative form: per-range guess / todo flags, per-workgroup verdict and quality histogram
  uint8_t* d_guess = nullptr;
  uint8_t* d_todo = nullptr;
  uint8_t* d_wg_ok = nullptr;
  uint32_t* d_hist_wg = nullptr;
  bool from_start = false;             // the session begins at the start of the input: header lines are class 0
  uint32_t last_tpr = 0;               // geometry of the last scan launch (K5 re-walks the same ranges)
  uint64_t last_ranges = 0;
  uint64_t* d_first_ord = nullptr;     // K5: per range, ordinal of the first line start it emits
  uint64_t cap_first_ord = 0;
  uint64_t* d_block_partials = nullptr;   // level-1 fold output
  uint64_t cap_blocks = 0;
  uint64_t* d_state = nullptr;   // [32 partial words][1024 hist words]
  uint64_t* h_state = nullptr;   // pinned mirror
  std::vector<hipEvent_t> ev_pool;     // timing events, 3 per scan launch of the current session
  size_t ev_used = 0;
  std::vector<uint64_t> ev_bytes;      // bytes of each timed launch
  std::vector<hipEvent_t> cp_pool;     // timing events around each H2D copy (2 per chunk)
  size_t cp_used = 0;
  uint32_t* d_ticket = nullptr;        // arrival counter of the fused fold (re-armed by the kernel)
  bool fresh = true;                   // no scan folded into d_state yet in this session
  // staging (host buffers / files)
  uint8_t* d_stage[2] = {nullptr, nullptr};
  // device-side BGZF inflate: compressed chunk, block table (device + pinned), error word
  uint8_t* d_comp[2] = {nullptr, nullptr};
  scfq_dinflate::Block* d_blk[2] = {nullptr, nullptr};
  scfq_dinflate::Block* h_blk[2] = {nullptr, nullptr};
  bool bgzf_warmed = false;                        // bgzf_inflate's first (empty) launch has set the device's scratch up
  hipEvent_t ev_piece[2] = {nullptr, nullptr};     // pieces of compressed bytes crossing the pinned ring (device inflate paths)
  unsigned piece_it = 0;
  // ... or written by the host's threads into one of two FINE-GRAINED device buffers and moved on by a device-to-device copy (copy_through_ring)
  uint8_t* fg_stage[2] = {nullptr, nullptr};
  uint64_t fg_cap = 0;
  hipEvent_t ev_fg[2] = {nullptr, nullptr};
  unsigned fg_it = 0;
  bool fg_refused = false;
  uint8_t* d_inf[2] = {nullptr, nullptr};      // inflated chunks (kStagePad + inf_cap each)
  uint32_t* d_dstatus = nullptr;
  uint64_t comp_cap = 0, inf_cap = 0;
  // the FIRST chunk of a BGZF file has small buffers of its own (64 MiB inflated): a process's first decode starts behind four small
  // allocations, and the big double buffers above are allocated on a helper thread under it (ensure_bgzf_first / ingest_bgzf_device)
  uint8_t* d_comp_first = nullptr;
  uint8_t* d_inf_first = nullptr;
  uint64_t comp_first_cap = 0, inf_first_cap = 0;
  uint8_t* h_pin[2] = {nullptr, nullptr};
  uint64_t stage_cap = 0;
  hipEvent_t ev_copied[2] = {nullptr, nullptr}, ev_scanned[2] = {nullptr, nullptr};
  hipEvent_t ev_caller = nullptr;      // orders the caller's stream (scfq_opts.wait_stream) before the private ones
  scfq_timing timing{};
  std::mutex mu;   // one counting session at a time per device context
};

// A session owns one context for its whole life. Every device has a small POOL of contexts (own streams, pinned
// and device buffers), so several host threads can run sessions on the same GPU concurrently — e.g. `sc fq-count
// --jobs=8 *.fq.gz` inflates 8 files at once while their scans share the device.
struct SessionLock {
  std::unique_lock<std::mutex> lk;
  void acquire(Ctx*) {}   // the context returned by get_ctx() is already locked into this object
};

int env_int(const char* name, int dflt);
}  // namespace
extern "C" void scfq_dedup_release_pools(void);      // scfq_dedup.hip
namespace {

std::mutex g_mu;
std::map<int, std::vector<std::unique_ptr<Ctx>>> g_ctx;
thread_local scfq_timing g_last_timing{};
thread_local uint64_t g_hist_stats[2] = {0, 0};   // ranges of the last session taken from the fast K3 form / redone exactly

int new_ctx(int dev, std::unique_ptr<Ctx>* out);
int want_copy_stream(Ctx* c);
std::map<int, int> g_ctx_creating;      // (under g_mu) contexts of a device being created right now
std::condition_variable g_ctx_cv;       // a context joined a pool, or a creation ended (with or without one)

int get_ctx(Ctx** out, SessionLock& sl) {
  int dev = 0;
  { static std::once_flag once; std::call_once(once, [] { trace("first device call of the process (runtime initialisation starts)"); }); }
  HIPCHK(hipGetDevice(&dev));
  { static std::once_flag once; std::call_once(once, [] { trace("runtime initialised (hipGetDevice returned)"); }); }
  static const int max_ctx = std::max(1, std::min(64, env_int("SCFQ_MAX_SESSIONS", 16)));
  Ctx* wait_on = nullptr;
  {
    std::unique_lock<std::mutex> lk(g_mu);
    for (;;) {
      auto& pool = g_ctx[dev];
      for (auto& c : pool) {
        std::unique_lock<std::mutex> try_lk(c->mu, std::try_to_lock);
        if (try_lk.owns_lock()) { sl.lk = std::move(try_lk); *out = c.get(); return SCFQ_OK; }
      }
      if ((int)pool.size() + g_ctx_creating[dev] < max_ctx) {
        // a new context is made OUTSIDE the registry's lock (its streams and pinned words take tens of milliseconds): the sessions of
        // `--jobs=N` set their contexts up side by side, not one after the other
        ++g_ctx_creating[dev];
        lk.unlock();
        std::unique_ptr<Ctx> c;
        const int rc = new_ctx(dev, &c);
        lk.lock();
        --g_ctx_creating[dev];
        if (rc) { g_ctx_cv.notify_all(); return rc; }
        sl.lk = std::unique_lock<std::mutex>(c->mu);
        *out = c.get();
        g_ctx[dev].push_back(std::move(c));
        g_ctx_cv.notify_all();
        return SCFQ_OK;
      }
      if (!pool.empty()) break;
      // every allowed context is still being created by another thread: wait for the first to appear — or for the creators to give up
      // (out of memory, no stream), in which case the loop comes round to try a creation of its own and returns ITS error
      g_ctx_cv.wait(lk, [&] { return !g_ctx[dev].empty() || g_ctx_creating[dev] == 0; });
    }
    static unsigned rr = 0;
    wait_on = g_ctx[dev][rr++ % g_ctx[dev].size()].get();
  }
  sl.lk = std::unique_lock<std::mutex>(wait_on->mu);   // every context busy: queue behind one of them
  *out = wait_on;
  return SCFQ_OK;
}

int new_ctx(int dev, std::unique_ptr<Ctx>* out) {
  trace("first use of the device: creating a context");
  auto c = std::make_unique<Ctx>();
  c->dev = dev;
  int n_cu = 0;
  HIPCHK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
  c->n_cu = n_cu > 0 ? n_cu : 256;
  trace("  device attributes read");
  // ONE stream to begin with: a stream costs 8 - 16 ms to create, and an input of one chunk (a gzip file of one batch, a BGZF or
  // plain file of one chunk) is a chain — its copy and its kernels have nothing to overlap with.  `copy` is the compute stream
  // under another name until a path has a second chunk to move (want_copy_stream()).
  // (r4, measured and dropped: the second stream and the pinned ring created on helper threads beside the first stream — the
  // runtime serialises the three, the context came up in 64 ms instead of 56; profiles/r04/cold_stages_parallel_bringup.jsonl)
  {
    // (SCFQ_COMPUTE_PRIORITY=1: the context's compute stream above the default priority — an A/B knob of the device gzip path, whose
    // window / resolve / scan kernels share the device with decode kernels on streams of default priority)
    int least = 0, greatest = 0;
    if (env_int_early("SCFQ_COMPUTE_PRIORITY", 0) && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest)
      HIPCHK(hipStreamCreateWithPriority(&c->compute, hipStreamNonBlocking, greatest));
    else
      HIPCHK(hipStreamCreateWithFlags(&c->compute, hipStreamNonBlocking));
  }
  c->copy = c->compute;
  c->copy_is_alias = true;
  trace("  compute stream created");
  HIPCHK(hipMalloc(&c->d_state, kStateWords * sizeof(uint64_t) + 64));
  c->d_ticket = reinterpret_cast<uint32_t*>(c->d_state + kStateWords);      // (the fold's arrival counter lives behind the state words)
  HIPCHK(hipHostMalloc(&c->h_state, kStateWords * sizeof(uint64_t), hipHostMallocDefault));
  trace("  state words allocated (device + pinned)");
  HIPCHK(hipMemsetAsync(c->d_ticket, 0, sizeof(uint32_t), c->compute));
  for (int b = 0; b < 2; ++b) {
    HIPCHK(hipEventCreateWithFlags(&c->ev_copied[b], hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ev_scanned[b], hipEventDisableTiming));
  }
  *out = std::move(c);
  trace("context up");
  return SCFQ_OK;
}

// a copy stream of its own, from now on (idempotent).  What was queued on the alias stays ordered: it is on the compute stream,
// and every consumer of a copy waits for the event recorded behind it, whichever stream that was.
int want_copy_stream(Ctx* c) {
  if (!c->copy_is_alias) return SCFQ_OK;
  hipStream_t st = nullptr;
  HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  c->copy = st;
  c->copy_is_alias = false;
  trace("copy stream created (a second chunk to move)");
  return SCFQ_OK;
}

int ensure_partials(Ctx* c, uint64_t n_ranges, bool hist) {
  if (n_ranges > c->cap_ranges) {
    HIPCHK(hipStreamSynchronize(c->compute));
    if (c->d_partials) HIPCHK(hipFree(c->d_partials));
    c->d_partials = nullptr;
    uint64_t cap = std::max<uint64_t>(n_ranges + n_ranges / 4, 4096);
    HIPCHK(hipMalloc(&c->d_partials, cap * scfq::kPartialWords * sizeof(uint64_t)));
    c->cap_ranges = cap;
    if (c->d_block_partials) HIPCHK(hipFree(c->d_block_partials));
    c->d_block_partials = nullptr;
    c->cap_blocks = cap / scfq::kFold1 + 2;
    HIPCHK(hipMalloc(&c->d_block_partials, c->cap_blocks * scfq::kPartialWords * sizeof(uint64_t)));
  }
  if (hist && n_ranges > c->cap_hist_ranges) {
    HIPCHK(hipStreamSynchronize(c->compute));
    if (c->d_hist_partials) HIPCHK(hipFree(c->d_hist_partials));
    if (c->d_range_phase) HIPCHK(hipFree(c->d_range_phase));
    if (c->d_guess) HIPCHK(hipFree(c->d_guess));
    if (c->d_hist_wg) HIPCHK(hipFree(c->d_hist_wg));
    c->d_hist_partials = nullptr;
    c->d_range_phase = nullptr;
    c->d_guess = nullptr;
    c->d_hist_wg = nullptr;
    uint64_t cap = std::max<uint64_t>(n_ranges + n_ranges / 4, 4096);
    HIPCHK(hipMalloc(&c->d_hist_partials, cap * 1024 * sizeof(uint32_t)));
    HIPCHK(hipMalloc(&c->d_range_phase, 2 * cap));
    HIPCHK(hipMalloc(&c->d_guess, 3 * cap));          // guess | todo | wg_ok
    c->d_todo = c->d_guess + cap;
    c->d_wg_ok = c->d_todo + cap;
    HIPCHK(hipMalloc(&c->d_hist_wg, (cap / scfq::kQWaves + 1) * 256 * sizeof(uint32_t)));
    c->cap_hist_ranges = cap;
  }
  return SCFQ_OK;
}

int ensure_staging(Ctx* c, uint64_t chunk, bool pinned) {
  if (chunk > c->stage_cap) {
    HIPCHK(hipStreamSynchronize(c->compute));
    HIPCHK(hipStreamSynchronize(c->copy));
    for (int b = 0; b < 2; ++b) {
      if (c->d_stage[b]) HIPCHK(hipFree(c->d_stage[b]));
      c->d_stage[b] = nullptr;
    }
    if (c->h_pin[0]) HIPCHK(hipHostFree(c->h_pin[0]));      // (both halves are one allocation)
    c->h_pin[0] = c->h_pin[1] = nullptr;
    note_dev_bytes(-(int64_t)(2 * c->stage_cap));      // (what was freed above)
    c->stage_cap = 0;
    for (int b = 0; b < 2; ++b) HIPCHK(hipMalloc(&c->d_stage[b], chunk));
    note_dev_bytes((int64_t)(2 * chunk));
    c->stage_cap = chunk;
  }
  if (pinned && !c->h_pin[0]) {
    // (ONE allocation for the two halves: a pinned allocation costs milliseconds before its first byte)
    HIPCHK(hipHostMalloc(&c->h_pin[0], 2 * c->stage_cap, hipHostMallocDefault));
    c->h_pin[1] = c->h_pin[0] + c->stage_cap;
    trace("pinned staging ring allocated");
  }
  return SCFQ_OK;
}

uint32_t pick_tiles_per_range(const Ctx* c, uint64_t n_tiles) {
  if (const char* e = std::getenv("SCFQ_TILES_PER_RANGE")) {
    long v = std::atol(e);
    if (v >= 1) return (uint32_t)std::min<long>(v, scfq::kMaxTilesPerRange);
  }
  // enough ranges that every CU sees many waves come and go (dynamic balance from block dispatch),
  // few enough that the ordered fold stays negligible
  const uint64_t target_ranges = (uint64_t)c->n_cu * 12 * 8;
  uint64_t tpr = (n_tiles + target_ranges - 1) / target_ranges;
  tpr = std::max<uint64_t>(tpr, 16);
  tpr = std::min<uint64_t>(tpr, scfq::kMaxTilesPerRange);
  return (uint32_t)tpr;
}

// the scan kernel instance and range geometry of this thread's last launch (scfq_debug_last_scan_kernel: bench.py attaches committed PMC
// traffic only to a run of the very kernel that was profiled)
thread_local char g_last_scan[128] = "";

template <bool S, int H, int RING, bool NT, bool GUESS = false>
void launch_scan(const scfq::ScanArgs& a, unsigned blocks, hipStream_t st) {
  if (!GUESS)
    std::snprintf(g_last_scan, sizeof g_last_scan, "fq_scan_tiles<%s, %d, %d, %s, %s> tiles_per_range=%u", S ? "true" : "false", H, RING, NT ? "true" : "false",
                  GUESS ? "true" : "false", a.tiles_per_range);
  constexpr int waves = (H == 1) ? scfq::kHistWaves : (H == 2) ? scfq::kQWaves : scfq::kWavesPerBlock;
  unsigned lds = waves * RING * scfq::kTile;
  if (H == 1) lds += waves * scfq::kHistWords * sizeof(uint32_t);
  if (H == 2) lds += scfq::kQWords * sizeof(uint32_t);
  (void)blocks;
  const unsigned grid = (unsigned)((a.n_ranges + waves - 1) / waves);
  hipLaunchKernelGGL((scfq::fq_scan_tiles<S, H, RING, NT, GUESS>), dim3(grid), dim3(64 * waves), lds, st, a);
}

// tuning knobs (defaults are the measured best): SCFQ_RING = 2|3|4 LDS ring slots per wave, SCFQ_NT = 0|1
int env_int(const char* name, int dflt) {
  const char* e = std::getenv(name);
  return e ? std::atoi(e) : dflt;
}

// Enqueue scan + fold of one device-resident range onto the compute stream. State accumulates.
int scan_async(Ctx* c, const uint8_t* dptr, uint64_t n, int prev_byte, uint32_t flags, bool timing) {
  if (n == 0) return SCFQ_OK;
  const bool hist = flags & SCFQ_QUAL_HIST, strct = flags & SCFQ_STRUCT_CHECK;
  const uint64_t B = (uint64_t)(uintptr_t)dptr, A0 = B & ~(uint64_t)(scfq::kTile - 1);
  const uint64_t NT = (B + n - A0 + scfq::kTile - 1) / scfq::kTile;
  if (NT >= (1ull << 32)) { std::snprintf(g_err, sizeof g_err, "a single scan launch covers at most 16 TiB"); return SCFQ_EARG; }
  const uint32_t tpr = pick_tiles_per_range(c, NT);
  const uint64_t n_ranges = (NT + tpr - 1) / tpr;
  int rc = ensure_partials(c, n_ranges, hist);
  if (rc) return rc;
  c->last_tpr = tpr;
  c->last_ranges = n_ranges;
  scfq::ScanArgs a;
  a.base = dptr;
  a.n = n;
  a.prev_byte = prev_byte;
  a.tiles_per_range = tpr;
  a.n_ranges = n_ranges;
  a.partials = c->d_partials;
  a.hist_partials = c->d_hist_partials;
  const unsigned blocks = (unsigned)((n_ranges + scfq::kWavesPerBlock - 1) / scfq::kWavesPerBlock);
  hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
  if (timing) {
    while (c->ev_pool.size() < c->ev_used + 3) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); c->ev_pool.push_back(e); }
    for (int k = 0; k < 3; ++k) ev[k] = c->ev_pool[c->ev_used + k];
    c->ev_used += 3;
    c->ev_bytes.push_back(n);
    HIPCHK(hipEventRecord(ev[0], c->compute));
  }
  // K3: speculative form unless SCFQ_HIST_EXACT asks for the exact 4-class histogram (env SCFQ_HIST_MODE=exact|fast overrides)
  static const int hist_mode_env = [] { const char* e = std::getenv("SCFQ_HIST_MODE"); return !e ? 0 : (e[0] == 'e' ? 1 : 2); }();
  const bool spec = hist && (hist_mode_env ? hist_mode_env == 2 : !(flags & SCFQ_HIST_EXACT));
  a.todo = nullptr;
  a.guess = c->d_guess;
  a.hist_wg = c->d_hist_wg;
  a.guess_out = c->d_guess;
  a.guess_cap_tiles = (uint32_t)std::max(1, env_int("SCFQ_GUESS_TILES", 64));
  if (spec) {
    launch_scan<true, 0, 2, false, true>(a, blocks, c->compute);   // guess pass: first tiles of every range
    if (strct) launch_scan<true, 2, 2, true>(a, blocks, c->compute);
    else launch_scan<false, 2, 2, true>(a, blocks, c->compute);
  }
  else if (hist && strct) launch_scan<true, 1, 2, true>(a, blocks, c->compute);
  else if (hist) launch_scan<false, 1, 2, true>(a, blocks, c->compute);
  else if (strct) launch_scan<true, 0, 2, true>(a, blocks, c->compute);
  else {
    // measured on MI355X (10 GB Illumina): ring 2 + nt 6.36 TB/s, ring 3 + nt 6.23, ring 2 5.91, ring 3 5.87
    static const int ring = env_int("SCFQ_RING", 2), nt = env_int("SCFQ_NT", 1);
    if (ring == 2 && nt) launch_scan<false, 0, 2, true>(a, blocks, c->compute);
    else if (ring == 2) launch_scan<false, 0, 2, false>(a, blocks, c->compute);
    else if (ring == 4 && nt) launch_scan<false, 0, 4, true>(a, blocks, c->compute);
    else if (ring == 4) launch_scan<false, 0, 4, false>(a, blocks, c->compute);
    else if (nt) launch_scan<false, 0, 3, true>(a, blocks, c->compute);
    else launch_scan<false, 0, 3, false>(a, blocks, c->compute);
  }
  HIPCHK(hipGetLastError());
  if (timing) HIPCHK(hipEventRecord(ev[1], c->compute));
  {
    const uint64_t n_blocks = (n_ranges + scfq::kFold1 - 1) / scfq::kFold1;
    uint8_t* rel_phase = hist ? c->d_range_phase : nullptr;
    uint8_t* block_phase = hist ? c->d_range_phase + c->cap_hist_ranges : nullptr;
    if (hist && c->fresh)
      HIPCHK(hipMemsetAsync(c->d_state + SCFQ_PARTIAL_WORDS, 0, (SCFQ_HIST_WORDS + scfq::kExtWords) * sizeof(uint64_t), c->compute));
    hipLaunchKernelGGL(scfq::fq_fold_fused, dim3((unsigned)n_blocks), dim3(scfq::kFold1), 0, c->compute, c->d_partials,
                       n_ranges, c->d_block_partials, c->d_ticket, c->d_state, c->fresh ? 1 : 0, rel_phase, block_phase,
                       dptr, n);
    HIPCHK(hipGetLastError());
    c->fresh = false;
    unsigned long long* state_hist = (unsigned long long*)(c->d_state + SCFQ_PARTIAL_WORDS);
    if (spec) {
      // verify the guesses against the exact phases, redo what did not verify with the exact kernel, fold both kinds
      uint64_t* ext = c->d_state + kExtAt;
      const uint64_t n_wg = (n_ranges + scfq::kQWaves - 1) / scfq::kQWaves;
      hipLaunchKernelGGL(scfq::fq_hist_verify, dim3((unsigned)((n_wg + 255) / 256)), dim3(256), 0, c->compute, c->d_guess, rel_phase, block_phase, n_ranges,
                         c->from_start ? 0 : -1, ext, c->d_todo, c->d_wg_ok, c->d_hist_wg);
      HIPCHK(hipGetLastError());
      a.todo = c->d_todo;
      if (strct) launch_scan<true, 1, 2, true>(a, blocks, c->compute);
      else launch_scan<false, 1, 2, true>(a, blocks, c->compute);
      HIPCHK(hipGetLastError());
      hipLaunchKernelGGL(scfq::fq_fold_hist, dim3((unsigned)n_blocks), dim3(256), 0, c->compute, c->d_hist_partials,
                         rel_phase, block_phase, n_ranges, c->d_todo, state_hist);
      hipLaunchKernelGGL(scfq::fq_fold_hist_wg, dim3((unsigned)((n_wg + scfq::kFoldWgPer - 1) / scfq::kFoldWgPer)), dim3(256), 0, c->compute, c->d_hist_wg,
                         c->d_wg_ok, n_wg, ext, state_hist);
      HIPCHK(hipGetLastError());
    } else if (hist) {
      hipLaunchKernelGGL(scfq::fq_fold_hist, dim3((unsigned)n_blocks), dim3(256), 0, c->compute, c->d_hist_partials,
                         rel_phase, block_phase, n_ranges, (const uint8_t*)nullptr, state_hist);
      HIPCHK(hipGetLastError());
    }
  }
  if (timing) HIPCHK(hipEventRecord(ev[2], c->compute));   // resolved in end_session, after the one stream sync
  return SCFQ_OK;
}

int begin_session(Ctx* c, bool from_start) {
  ++c->n_sessions;
  c->from_start = from_start;
  c->timing = scfq_timing{};
  c->timing.struct_size = sizeof(scfq_timing);
  c->ev_used = 0;
  c->cp_used = 0;
  c->ev_bytes.clear();
  c->fresh = true;   // the first fold of the session overwrites d_state (no memset launch)
  return SCFQ_OK;
}

int end_session(Ctx* c, bool hist, scfq_partial* out, uint64_t* hist_out) {
  const size_t words = hist ? kStateWords : SCFQ_PARTIAL_WORDS;
  if (c->fresh) {   // nothing was scanned (empty input)
    std::memset(c->h_state, 0, words * sizeof(uint64_t));
  } else {
    HIPCHK(hipMemcpyAsync(c->h_state, c->d_state, words * sizeof(uint64_t), hipMemcpyDeviceToHost, c->compute));
    HIPCHK(hipStreamSynchronize(c->compute));
  }
  for (size_t k = 0; k + 2 < c->ev_used + 0 && k < c->ev_used; k += 3) {
    float ms1 = 0, ms2 = 0;
    HIPCHK(hipEventElapsedTime(&ms1, c->ev_pool[k], c->ev_pool[k + 1]));
    HIPCHK(hipEventElapsedTime(&ms2, c->ev_pool[k + 1], c->ev_pool[k + 2]));
    c->timing.scan_kernel_ms += ms1;
    c->timing.fold_kernel_ms += ms2;
    c->timing.scan_bytes += c->ev_bytes[k / 3];
    c->timing.scan_launches += 1;
  }
  for (size_t k = 0; k + 1 < c->cp_used; k += 2) {
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, c->cp_pool[k], c->cp_pool[k + 1]));
    c->timing.h2d_ms += ms;
  }
  trace("session folded");
  std::memcpy(out, c->h_state, sizeof(scfq_partial));
  out->hist_class = 0;
  g_hist_stats[0] = g_hist_stats[1] = 0;
  if (hist) {
    // speculative K3: only the class the fast form took to be the quality line is complete (0 = all four are)
    const uint64_t h1 = c->h_state[kExtAt + scfq::kExtH];
    out->hist_class = h1 ? (((h1 - 1) + 3) & 3u) + 1 : 0;
    g_hist_stats[0] = c->h_state[kExtAt + scfq::kExtFast];
    g_hist_stats[1] = c->h_state[kExtAt + scfq::kExtRedo];
  }
  if (hist && hist_out) std::memcpy(hist_out, c->h_state + SCFQ_PARTIAL_WORDS, SCFQ_HIST_WORDS * sizeof(uint64_t));
  g_last_timing = c->timing;
  return SCFQ_OK;
}

uint32_t opt_flags(const scfq_opts* o) { return o ? o->flags : 0u; }
uint64_t opt_chunk(const scfq_opts* o) {
  uint64_t ch = (o && o->chunk_bytes) ? o->chunk_bytes : kDefaultChunk;
  ch = (ch + scfq::kTile - 1) & ~(uint64_t)(scfq::kTile - 1);
  return std::max<uint64_t>(ch, scfq::kTile);
}
// the caller's opts as a current-size struct (v1 callers pass 40 bytes: everything after chunk_bytes reads as zero)
scfq_opts opts_copy(const scfq_opts* o) {
  scfq_opts c{};
  if (o) std::memcpy(&c, o, std::min<uint64_t>(o->struct_size, sizeof c));
  c.struct_size = sizeof c;
  return c;
}
thread_local void* g_wait_stream = nullptr;       // scfq_set_wait_stream(): this host thread's default caller stream
thread_local bool g_wait_enabled = false;
bool opt_wait_stream(const scfq_opts* o, void** ws) {
  if (o && o->struct_size >= sizeof(scfq_opts) && (o->flags & SCFQ_WAIT_STREAM)) { *ws = o->wait_stream; return true; }
  *ws = g_wait_stream;
  return g_wait_enabled;
}
// Device-pointer arguments are used on the context's private streams; opts->wait_stream (the caller's stream) is ordered
// before them by an event (include/sc_fqcount.h: scfq_opts.wait_stream).
int wait_for_caller(Ctx* c, const scfq_opts* o) {
  void* ws = nullptr;
  if (!opt_wait_stream(o, &ws)) return SCFQ_OK;
  if (!c->ev_caller) HIPCHK(hipEventCreateWithFlags(&c->ev_caller, hipEventDisableTiming));
  HIPCHK(hipEventRecord(c->ev_caller, static_cast<hipStream_t>(ws)));
  HIPCHK(hipStreamWaitEvent(c->compute, c->ev_caller, 0));
  HIPCHK(hipStreamWaitEvent(c->copy, c->ev_caller, 0));
  return SCFQ_OK;
}
int check_opts(const scfq_opts* o) {
  if (o && o->struct_size != sizeof(scfq_opts) && o->struct_size != SCFQ_OPTS_V1_SIZE) return SCFQ_EARG;
  if (o && (o->flags & ~(SCFQ_QUAL_HIST | SCFQ_STRUCT_CHECK | SCFQ_TIMING | SCFQ_PREV_IN_MEMORY | SCFQ_HIST_EXACT | SCFQ_WAIT_STREAM))) return SCFQ_EARG;
  if (o && o->n_devices < 0) return SCFQ_EARG;
  if (o && o->n_devices > 0 && !o->device_ids) return SCFQ_EARG;
  return SCFQ_OK;
}

// ---- C1 inside one process: communicators of the multi-device path (scfq_opts.n_devices > 1) -------------------------
// One RCCL communicator per listed device (ncclCommInitAll), kept until scfq_shutdown.  RCCL cannot form a communicator
// that names a device twice, so a list with repeats (several ingest sessions sharing one GPU to overlap file reads) folds
// its partials on the host, as does SCFQ_EXCHANGE=host (a host that does not want librccl mapped for 256 bytes per rank).
std::mutex g_comm_mu;
// A set is shared by the registry and by every session that is using it: a set that failed is taken OUT of the registry at once (the next
// count builds a fresh one) but its communicators are only destroyed when the last session that still holds it lets go — another thread
// may be parked on the set's mutex or inside its own exchange with copies of the same pointers.
struct CommSet {
  std::vector<scfq_comm*> comms;
  std::mutex one_at_a_time;             // exchanges of concurrent sessions on the same communicators must not interleave
  bool retired = false;                 // (under one_at_a_time) an exchange on this set failed: whoever comes next does not use it
  ~CommSet() { for (scfq_comm* c : comms) if (c) scfq_comm_destroy(c); }
};
std::map<std::vector<int>, std::shared_ptr<CommSet>> g_comms;

void release_comms() {
  std::map<std::vector<int>, std::shared_ptr<CommSet>> gone;
  { std::lock_guard<std::mutex> lk(g_comm_mu); gone.swap(g_comms); }
  // (destroyed here, outside the registry's lock, unless a session still holds a set: then when that session ends)
}

bool exchange_on_host(const int32_t* ids, int nd) {
  const char* e = std::getenv("SCFQ_EXCHANGE");
  if (e && e[0] == 'h') return true;
  for (int a = 0; a < nd; ++a) for (int b = a + 1; b < nd; ++b) if (ids[a] == ids[b]) return true;
  return false;
}

// parts[d] (+ hists[d]) of the devices in list order -> their rank-ordered fold
int fold_device_partials(const scfq_opts& o, int nd, const std::vector<scfq_partial>& parts, const std::vector<std::vector<uint64_t>>& hists,
                         bool want_hist, scfq_partial* p, uint64_t* hist) {
  if (exchange_on_host(o.device_ids, nd)) {
    scfq_partial_identity(p, want_hist ? hist : nullptr);
    for (int d = 0; d < nd; ++d) scfq_partial_combine(p, &parts[d], want_hist ? hist : nullptr, want_hist ? hists[d].data() : nullptr);
    return SCFQ_OK;
  }
  std::shared_ptr<CommSet> set, given_up;      // (given_up: destroyed — if this is its last holder — after the registry's lock is released)
  std::vector<int> key(o.device_ids, o.device_ids + nd);
  {
    std::lock_guard<std::mutex> lk(g_comm_mu);
    auto it = g_comms.find(key);
    if (it != g_comms.end()) {
      // a set with a broken member (an exchange timed out) is given up here, so that one stuck collective does not fail
      // every later multi-device count of the process: the next lines build a fresh set (the old one goes when its last user does)
      bool broken = false;
      for (scfq_comm* c : it->second->comms) broken = broken || scfq_comm_is_broken(c);
      if (broken) { given_up = std::move(it->second); g_comms.erase(it); it = g_comms.end(); }
    }
    if (it == g_comms.end()) {
      auto fresh = std::make_shared<CommSet>();
      fresh->comms.assign((size_t)nd, nullptr);
      const int rc = scfq_comm_init_all(nd, o.device_ids, 0, fresh->comms.data());
      if (rc) { std::snprintf(g_err, sizeof g_err, "%s", scfq_comm_error_detail()); return rc; }
      it = g_comms.emplace(key, std::move(fresh)).first;
    }
    set = it->second;
  }
  if (!p) return SCFQ_OK;               // scfq_prepare(): only the communicators were wanted
  const std::vector<scfq_comm*>& comms = set->comms;
  int rc = SCFQ_OK;
  {
    std::lock_guard<std::mutex> lk(set->one_at_a_time);
    // (a set that broke while this session waited for its turn: not used, the caller's count fails like the one that broke it)
    if (set->retired) rc = SCFQ_ERCCL;
    for (scfq_comm* c : comms) if (scfq_comm_is_broken(c)) rc = SCFQ_ERCCL;
    if (rc) std::snprintf(g_err, sizeof g_err, "exchange: the communicator set was broken by an earlier exchange");
    int started = 0;
    for (int d = 0; d < nd && !rc; ++d) {
      rc = scfq_comm_exchange_start(comms[d], &parts[d], want_hist ? hists[d].data() : nullptr, 0);
      if (!rc) ++started;
      else std::snprintf(g_err, sizeof g_err, "%s", scfq_comm_error_detail());
    }
    std::vector<uint64_t> h2(want_hist ? SCFQ_HIST_WORDS : 0);
    for (int d = 0; d < started; ++d) {
      scfq_partial q;
      const int r = scfq_comm_exchange_finish(comms[d], d == 0 ? p : &q, want_hist ? (d == 0 ? hist : h2.data()) : nullptr, 0);
      if (r && !rc) { rc = r; std::snprintf(g_err, sizeof g_err, "%s", scfq_comm_error_detail()); }
      if (!r && !rc && d > 0 && (std::memcmp(&q, p, sizeof q) != 0 || (want_hist && std::memcmp(h2.data(), hist, SCFQ_HIST_WORDS * sizeof(uint64_t)) != 0))) {
        std::snprintf(g_err, sizeof g_err, "exchange: rank %d folded a different result than rank 0", d);
        rc = SCFQ_ERCCL;
      }
    }
    if (rc) set->retired = true;
  }
  if (rc) {
    // whatever went wrong, this set is not handed out again (a started exchange that was never finished would answer the next one);
    // it is destroyed when the last session holding it — this one, or one still waiting for its turn above — drops it
    {
      std::lock_guard<std::mutex> lk2(g_comm_mu);
      auto it = g_comms.find(key);
      if (it != g_comms.end() && it->second == set) g_comms.erase(it);
    }
    set.reset();
  }
  return rc;
}

#include "scfq_sources.hpp"   // Source, MemSource, FdSource, GzSource, FastGzSource, BgzfSource, open_gz_source

// (r4, measured and dropped: filling in the page tables of the mapped file on a helper thread — MADV_POPULATE_READ — while the runtime
// comes up.  The copies into the pinned ring ran no faster: 10.7 against 11.7 ms for a 121 MB file, profiles/r04/cold_stages_*.jsonl.)
// `n` bytes of host (pageable: a mapped file) memory to device memory through the context's pinned ring, in pieces: a piece
// crosses PCIe while the next one is copied into the other pinned buffer, so the bytes are on the device one piece after the last
// of them reached pinned memory — and the ring is 2 x 16 MiB, not two buffers of the size of a chunk (pinning memory costs
// 160 ms per GB: 170 of the 310 ms of a cold `sc fq-count` over a 2 GB BGZF file went into two pinned buffers of 0.5 GB).
int copy_through_ring(Ctx* c, uint8_t* dst_device, const FileBytes& src, uint64_t src_off, uint64_t n) {
  // (r5) Chunks of 8 MiB and more do not go through the pinned ring at all: the host's threads pread them into one of two 64 MiB buffers of
  // fine-grained device memory (mapped into the process: posted writes over PCIe, 43 GB/s from 8 threads on, one pass over host memory),
  // and a device-to-device copy moves a piece to where the kernels read it — the device gzip path's feed (scfq_gzdev.hpp).  Through the
  // ring's 16 MiB halves the same threads reach 17 GB/s (profiles/r05/h2d_rate.txt), and the copy engine was idle for three quarters of a
  // warm BGZF call (profiles/r05/bgzf_timeline.txt).  SCFQ_BGZF_HOST_WRITES=0: the ring.
  static const bool host_writes = env_int("SCFQ_BGZF_HOST_WRITES", 1) != 0;
  if (host_writes && n >= (8ull << 20) && !c->fg_refused) {
    // (a context's first session: 32 MiB pieces — the first write to a page of such a buffer is a fault, 8 ms per 64 MiB, and a process pays for
    // every page it touches once: profiles/r05/cold_staging_pieces_ab.txt; later sessions: 64 MiB.  SCFQ_BGZF_STAGING_MB = n: always n MiB)
    static const int staging_mb = env_int("SCFQ_BGZF_STAGING_MB", 0);
    const uint64_t want_piece = staging_mb > 0 ? ((uint64_t)std::max(8, staging_mb) << 20) : (c->n_sessions <= 1 ? (32ull << 20) : (64ull << 20));
    if (c->fg_stage[0] && c->fg_cap < want_piece) {
      // (a bigger piece than the last session's: the streams are idle between sessions)
      HIPCHK(hipStreamSynchronize(c->copy));
      for (int b = 0; b < 2; ++b) { (void)hipFree(c->fg_stage[b]); c->fg_stage[b] = nullptr; }
      note_dev_bytes(-(int64_t)(2 * c->fg_cap));
      c->fg_cap = 0;
    }
    if (!c->fg_stage[0]) {
      const uint64_t bytes = want_piece;
      for (int b = 0; b < 2 && !c->fg_refused; ++b)
        if (hipExtMallocWithFlags(reinterpret_cast<void**>(&c->fg_stage[b]), bytes, hipDeviceMallocFinegrained) != hipSuccess) { (void)hipGetLastError(); c->fg_stage[b] = nullptr; c->fg_refused = true; }
      if (c->fg_refused) { for (int b = 0; b < 2; ++b) { if (c->fg_stage[b]) (void)hipFree(c->fg_stage[b]); c->fg_stage[b] = nullptr; } }
      else { c->fg_cap = bytes; note_dev_bytes((int64_t)(2 * bytes)); }
    }
    if (c->fg_stage[0]) {
      for (int b = 0; b < 2; ++b) if (!c->ev_fg[b]) HIPCHK(hipEventCreateWithFlags(&c->ev_fg[b], hipEventDisableTiming));
      for (uint64_t o = 0; o < n; o += c->fg_cap, ++c->fg_it) {
        const int sb = (int)(c->fg_it & 1);
        const uint64_t len = std::min(c->fg_cap, n - o);
        if (c->fg_it >= 2) HIPCHK(hipEventSynchronize(c->ev_fg[sb]));      // the device-to-device copy of the piece before last has read this buffer
        copy_file_bytes(src, src_off + o, c->fg_stage[sb], len, true);
        std::atomic_thread_fence(std::memory_order_seq_cst);
        HIPCHK(hipMemcpyAsync(dst_device + o, c->fg_stage[sb], (size_t)len, hipMemcpyDeviceToDevice, c->copy));
        HIPCHK(hipEventRecord(c->ev_fg[sb], c->copy));
      }
      return SCFQ_OK;
    }
  }
  int rc = ensure_staging(c, c->stage_cap ? c->stage_cap : (16ull << 20), true);
  if (rc) return rc;
  const uint64_t piece = std::min<uint64_t>(c->stage_cap, 64ull << 20);
  for (int b = 0; b < 2; ++b) if (!c->ev_piece[b]) HIPCHK(hipEventCreateWithFlags(&c->ev_piece[b], hipEventDisableTiming));
  for (uint64_t o = 0; o < n; o += piece, ++c->piece_it) {
    const int pb = (int)(c->piece_it & 1);
    const uint64_t len = std::min(piece, n - o);
    if (c->piece_it >= 2) HIPCHK(hipEventSynchronize(c->ev_piece[pb]));
    copy_file_bytes(src, src_off + o, c->h_pin[pb], len);
    HIPCHK(hipMemcpyAsync(dst_device + o, c->h_pin[pb], (size_t)len, hipMemcpyHostToDevice, c->copy));
    HIPCHK(hipEventRecord(c->ev_piece[pb], c->copy));
  }
  return SCFQ_OK;
}


// Chunked ingest with copy/compute overlap: the host thread fills pinned buffer b (pread / inflate)
// while the copy stream moves buffer b^1 to HBM and the compute stream scans the chunk before it.
int ingest(Ctx* c, Source& src, int prev_byte, uint32_t flags, uint64_t chunk, bool timing) {
  int rc = ensure_staging(c, chunk, true);
  if (rc) return rc;
  int prev = prev_byte;
  using clk = std::chrono::steady_clock;
  const auto t_begin = clk::now();
  double fill_ms = 0;
  struct Fin {   // record wall time on every exit path
    Ctx* c; clk::time_point t0; double* fill;
    ~Fin() { c->timing.host_fill_ms += *fill; c->timing.ingest_wall_ms += std::chrono::duration<double, std::milli>(clk::now() - t0).count(); }
  } fin{c, t_begin, &fill_ms};
  for (unsigned it = 0;; ++it) {
    const int b = it & 1;
    if (it >= 2) HIPCHK(hipEventSynchronize(c->ev_copied[b]));   // pinned buffer b is free again
    const auto tf = clk::now();
    int64_t got = src.fill(c->h_pin[b], chunk);
    if (it == 1 && got > 0 && (rc = want_copy_stream(c))) return rc;      // a second chunk: its copy runs under the first one's scan
    fill_ms += std::chrono::duration<double, std::milli>(clk::now() - tf).count();
    if (got < 0) return (int)got;
    if (got == 0) break;
    c->timing.h2d_bytes += (uint64_t)got;
    if (it >= 2) HIPCHK(hipStreamWaitEvent(c->copy, c->ev_scanned[b], 0));   // device buffer b consumed
    if (timing) {
      while (c->cp_pool.size() < c->cp_used + 2) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); c->cp_pool.push_back(e); }
      HIPCHK(hipEventRecord(c->cp_pool[c->cp_used], c->copy));
    }
    HIPCHK(hipMemcpyAsync(c->d_stage[b], c->h_pin[b], (size_t)got, hipMemcpyHostToDevice, c->copy));
    if (timing) { HIPCHK(hipEventRecord(c->cp_pool[c->cp_used + 1], c->copy)); c->cp_used += 2; }
    HIPCHK(hipEventRecord(c->ev_copied[b], c->copy));
    HIPCHK(hipStreamWaitEvent(c->compute, c->ev_copied[b], 0));
    rc = scan_async(c, c->d_stage[b], (uint64_t)got, prev, flags & ~SCFQ_PREV_IN_MEMORY, timing);
    if (rc) return rc;
    HIPCHK(hipEventRecord(c->ev_scanned[b], c->compute));
    prev = c->h_pin[b][got - 1];
    if ((uint64_t)got < chunk) {
      // short fill: could still be followed by data for gz streams; loop again (fill returns 0 at end)
    }
  }
  return SCFQ_OK;
}

// ---- BGZF with device-side inflate: the host only walks the member headers and moves COMPRESSED bytes ---------------
constexpr uint32_t kMaxBlocksPerChunk = 1u << 18;
// Which symbol loop the device inflate kernels run (bgzf_inflate_kernel.hpp): the boundary-first one (2, symbol_loop_dense) unless
// SCFQ_INFLATE_LOOP says lanes (0, symbol_loop_lanes: r2's) or serial (1, r1's) — both kept for A/B measurements and covered by
// tests/test_gpu_bgzf_device.py::test_the_other_symbol_loops
inline uint32_t inflate_serial_loop() {
  static const uint32_t v = [] { const char* e = std::getenv("SCFQ_INFLATE_LOOP"); return !e ? 2u : (e[0] == 's' ? 1u : (e[0] == 'l' ? 0u : 2u)); }();
  return v;
}
constexpr int kFallbackToHost = 1;        // ingest_bgzf_device: could not set up, nothing queued
constexpr int kFallbackRest = 3;          // device gzip (scfq_gzdev.hpp), internal: a batch without room — the host takes the file from there
constexpr int kNotPureBgzf = 2;           // ingest_bgzf_device: the file holds something other than BGZF members <= 64 KiB (found on the
                                          // way: the header walk runs chunk by chunk under the device's work); queue drained, session to restart
constexpr uint64_t kStagePad = 4096;      // inflated chunks start one tile into their buffer: byte [-1] carries the look-behind

static int64_t bgzf_plan(const uint8_t* img, uint64_t n, uint64_t pos, uint64_t out_cap, uint64_t comp_cap, uint32_t max_blocks,
                         scfq_dinflate::Block* blocks, uint32_t* n_blocks, uint64_t* out_bytes);

bool bgzf_device_enabled() {
  static const bool v = [] { const char* e = std::getenv("SCFQ_BGZF_DEVICE"); return e ? e[0] != '0' : true; }();
  return v;
}

// true when [img, img + n) is nothing but BGZF members of at most 64 KiB (what the device kernel takes)
bool bgzf_is_pure(const uint8_t* img, uint64_t n) {
  uint64_t p = 0;
  while (p < n) {
    uint32_t hl = 0;
    const uint32_t bs = scfq_bgzf::block_size(img + p, n - p, &hl);
    if (!bs || p + bs > n || scfq_bgzf::rd32(img + p + bs - 4) > (1u << 16)) return false;
    p += bs;
  }
  return n > 0;
}

// block tables (device + pinned) of both slots and the status word: small, allocated once per context
bool ensure_bgzf_tables(Ctx* c) {
  bool ok = true;
  for (int b = 0; b < 2 && ok; ++b) {
    if (!c->d_blk[b]) ok = hipMalloc(&c->d_blk[b], kMaxBlocksPerChunk * sizeof(scfq_dinflate::Block)) == hipSuccess;
    if (ok && !c->h_blk[b]) ok = hipHostMalloc(&c->h_blk[b], kMaxBlocksPerChunk * sizeof(scfq_dinflate::Block), hipHostMallocDefault) == hipSuccess;
  }
  if (ok && !c->d_dstatus) ok = hipMalloc(&c->d_dstatus, sizeof(uint32_t)) == hipSuccess;
  if (!ok) (void)hipGetLastError();
  return ok;
}

// The first chunk's own buffers: kBgzfFirstInflated of inflated bytes (1024 members of the maximum size) and as much of compressed
// ones (a member that does not compress is a stored block: its deflate data is a few bytes longer than what it holds).
constexpr uint64_t kBgzfFirstInflated = 64ull << 20;
int ensure_bgzf_first(Ctx* c) {
  if (c->d_comp_first && c->d_inf_first) return SCFQ_OK;
  bool ok = ensure_bgzf_tables(c);
  if (ok && !c->d_comp_first) ok = hipMalloc(&c->d_comp_first, kBgzfFirstInflated + 64) == hipSuccess;
  if (ok && !c->d_inf_first) ok = hipMalloc(&c->d_inf_first, kBgzfFirstInflated + kStagePad) == hipSuccess;
  if (!ok) {
    (void)hipGetLastError();
    if (c->d_comp_first) (void)hipFree(c->d_comp_first);
    if (c->d_inf_first) (void)hipFree(c->d_inf_first);
    c->d_comp_first = c->d_inf_first = nullptr;
    return kFallbackToHost;
  }
  c->comp_first_cap = c->inf_first_cap = kBgzfFirstInflated;
  note_dev_bytes((int64_t)(2 * kBgzfFirstInflated));
  trace("BGZF first-chunk buffers allocated");
  return SCFQ_OK;
}

// One wave inflates one member and a member is slow on its own (a serial bit stream): the kernel needs thousands of
// members per launch to fill 256 CUs, so the device path works in large inflated chunks (SCFQ_BGZF_DEVICE_CHUNK_MB) whatever
// the staging chunk of the host path is; compressed chunks are a third to a quarter of that.
// (r5: 640 MiB — two rounds of the kernel's 5120 wave slots — instead of 1 GiB: a launch of one round runs at 120 GB/s, of two at 149,
// of three at 179, but the call ends one launch after the last byte arrived: 2 GB file 25 -> 22 ms warm, 6 GB the same either way)
// (scfq_dinflate::Block keeps 32-bit offsets into the chunk: the knob is clamped so that a chunk stays below 4 GiB)
inline void bgzf_wanted_caps(uint64_t fsize, uint64_t* want_inf, uint64_t* want_comp) {
  static const uint64_t max_inf = (uint64_t)std::min(2048, std::max(64, env_int("SCFQ_BGZF_DEVICE_CHUNK_MB", 640))) << 20;
  *want_inf = std::min<uint64_t>(max_inf, std::max<uint64_t>(64ull << 20, (fsize * 5 + 4095) & ~4095ull));
  *want_comp = std::min<uint64_t>(*want_inf / 2, std::max<uint64_t>(32ull << 20, (fsize + 4095) & ~4095ull));
}
// the caps the big buffers WILL have for this file once ensure_bgzf_device_buffers has run (the member walk runs ahead of it)
inline void bgzf_caps_for(const Ctx* c, uint64_t fsize, uint64_t* inf, uint64_t* comp) {
  uint64_t want_inf = 0, want_comp = 0;
  bgzf_wanted_caps(fsize, &want_inf, &want_comp);
  const bool grow = c->comp_cap < want_comp || c->inf_cap < want_inf;
  *inf = grow ? want_inf : c->inf_cap;
  *comp = grow ? want_comp : c->comp_cap;
}

// buffers of the device-inflate path; kFallbackToHost when they cannot be had (nothing queued yet)
int ensure_bgzf_device_buffers(Ctx* c, uint64_t fsize, bool on_helper = false) {
  uint64_t want_inf = 0, want_comp = 0;
  bgzf_wanted_caps(fsize, &want_inf, &want_comp);
  if (c->comp_cap < want_comp || c->inf_cap < want_inf) {
    // (on the helper thread of ingest_bgzf_device the streams carry the first chunk of the NEW call, which does not touch these
    // buffers; the call before ended with both streams idle)
    if (!on_helper) {
      HIPCHK(hipStreamSynchronize(c->compute));
      HIPCHK(hipStreamSynchronize(c->copy));
    }
    for (int b = 0; b < 2; ++b) {
      if (c->d_comp[b]) HIPCHK(hipFree(c->d_comp[b]));
      if (c->d_inf[b]) HIPCHK(hipFree(c->d_inf[b]));
      c->d_comp[b] = nullptr; c->d_inf[b] = nullptr;
    }
    note_dev_bytes(-(int64_t)(2 * (c->comp_cap + c->inf_cap)));      // (what was freed above)
    c->comp_cap = c->inf_cap = 0;
    bool ok = true;
    for (int b = 0; b < 2 && ok; ++b)
      ok = hipMalloc(&c->d_comp[b], want_comp + 64) == hipSuccess && hipMalloc(&c->d_inf[b], want_inf + kStagePad) == hipSuccess;
    ok = ok && ensure_bgzf_tables(c);
    if (!ok) {
      // not enough device / pinned memory for the big chunks (many concurrent sessions): nothing was queued yet, so the
      // caller can still take the host inflate path
      (void)hipGetLastError();
      for (int b = 0; b < 2; ++b) {
        if (c->d_comp[b]) (void)hipFree(c->d_comp[b]);
        if (c->d_inf[b]) (void)hipFree(c->d_inf[b]);
        c->d_comp[b] = nullptr; c->d_inf[b] = nullptr;
      }
      return kFallbackToHost;
    }
    note_dev_bytes((int64_t)(2 * (want_comp + want_inf)));
    c->comp_cap = want_comp;
    c->inf_cap = want_inf;
    trace("BGZF device buffers allocated");
  }
  return SCFQ_OK;
}

// Members per launch.  BGZF members are (almost) all the same size, so the waves of one launch finish in rounds: with
// 16448 members on 4096 wave slots the 64 members of a fifth round ran alone for a fifth of the kernel's time (measured:
// 24.7 ms for 16448 members, 20.7 ms for 14192).  The member count of a full chunk is therefore a multiple of the slots
// (CUs x resident waves of bgzf_inflate, asked of the runtime); the byte caps of the chunk still apply.
// `launch`: 0, 1, 2 ... from the start of the file; the first launches are short (one round of members, one, two), so that the
// device starts after a quarter of the first copy and each of the next chunks is copied while the one before it is inflated: a
// full chunk straight behind the first short one left the device idle for 7 ms (its fill and copy take 11, the short launch 3).
static uint32_t bgzf_members_per_launch(uint64_t inflated_chunk, int launch = 3) {
  static const uint32_t slots = [] {
    int dev = 0, cus = 0, wgs = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&wgs, scfq_dinflate::bgzf_inflate, 64 * scfq_dinflate::kWavesPerWg,
                                                     scfq_dinflate::kWavesPerWg * scfq_dinflate::kWaveLdsBytes) != hipSuccess || cus <= 0 || wgs <= 0) {
      (void)hipGetLastError();
      return 0u;
    }
    return (uint32_t)(cus * wgs * scfq_dinflate::kWavesPerWg);
  }();
  const uint64_t fit = inflated_chunk >> 16;           // members of the maximum size (64 KiB) that fit the chunk
  if (!slots || fit < slots) return kMaxBlocksPerChunk;
  const uint64_t full = std::min<uint64_t>(kMaxBlocksPerChunk, fit / slots * slots);
  if (launch <= 1) return slots;
  if (launch == 2) return (uint32_t)std::min<uint64_t>(full, 2ull * slots);
  return (uint32_t)full;
}

// the powers of x bgzf_crc32_members multiplies with: made once per context, on its compute stream (in front of the kernels that read them)
inline int bgzf_crc_ready(Ctx* c) {
  if (c->bgzf_crc_consts) return SCFQ_OK;
  hipLaunchKernelGGL(scfq_dinflate::bgzf_crc_consts_init, dim3(1), dim3(256), 0, c->compute);
  HIPCHK(hipGetLastError());
  c->bgzf_crc_consts = true;
  return SCFQ_OK;
}

// first_prev: the byte in front of the first inflated byte (0..255), or -1 when the members start the input (a rank of a sharded
// BGZF file starts in the middle of the inflated stream: scfq_count_file_sharded)
// bytes_left_host (optional): called once when the last compressed byte has left the mapping (the launches behind it still run): the
// caller's moment to unmap the file — tearing down the page tables the member walk filled costs 8 ms for a 1.5 GB file, which then
// pass under the last chunk's inflate instead of behind it
int ingest_bgzf_device(Ctx* c, const uint8_t* img, uint64_t fsize, uint32_t flags, uint64_t /*chunk*/, bool timing, int first_prev = -1, int fd = -1, uint64_t fd_off = 0,
                       const std::function<void()>& bytes_left_host = std::function<void()>()) {
  FileBytes fbytes;
  fbytes.img = img; fbytes.fd = fd; fbytes.fd_off = fd_off;
  // (bgzf_inflate keeps 72 bytes of scratch per lane and the runtime sets the device's scratch up inside the first launch of such
  // a kernel — 35 - 50 ms on the launching thread: an empty launch on a helper thread, under the buffers' allocation and the first copy)
  struct Helper { std::thread th; ~Helper() { if (th.joinable()) th.join(); } };
  Helper warm, alloc;          // (declared before everything that they touch: joined last)
  int alloc_rc = SCFQ_OK;
  if (!c->bgzf_warmed) {
    c->bgzf_warmed = true;
    hipStream_t st = c->compute;
    const int dev = c->dev;
    warm.th = std::thread([st, dev] {
      if (hipSetDevice(dev) != hipSuccess) return;
      hipLaunchKernelGGL(scfq_dinflate::bgzf_inflate, dim3(1), dim3(64 * scfq_dinflate::kWavesPerWg), scfq_dinflate::kWavesPerWg * scfq_dinflate::kWaveLdsBytes, st,
                         (const uint8_t*)nullptr, (const scfq_dinflate::Block*)nullptr, 0u, (uint8_t*)nullptr, (uint32_t*)nullptr, 0u);
      (void)hipGetLastError();
      trace("bgzf_inflate's empty first launch returned (device scratch set up)");
    });
  }
  // The first chunk is SMALL and has buffers of its own (64 MiB inflated: four allocations of tens of MiB), so that the first members are
  // being inflated a few milliseconds into the call; the big double buffers (up to 2 x 1.5 GiB) are allocated on a helper thread under
  // that chunk's copy and decode, and only a file that goes on behind its first chunk waits for them.  Round 3 allocated everything up
  // front: whatever a big allocation costs on a given box at a given moment (the driver's cold BGZF process of BENCH_r03 took 3.4 s
  // before its first kernel where the builder's took 0.1 s) was paid before the first byte moved.
  int rc = ensure_bgzf_first(c);
  if (rc) return rc;
  HIPCHK(hipMemsetAsync(c->d_dstatus, 0, sizeof(uint32_t), c->compute));
  using clk = std::chrono::steady_clock;
  const auto t_begin = clk::now();
  double fill_ms = 0;
  struct Fin {
    Ctx* c; clk::time_point t0; double* fill;
    ~Fin() { c->timing.host_fill_ms += *fill; c->timing.ingest_wall_ms += std::chrono::duration<double, std::milli>(clk::now() - t0).count(); }
  } fin{c, t_begin, &fill_ms};
  uint64_t pos = 0;
  const uint8_t* prev_base = nullptr;
  uint64_t prev_n = 0;
  double plan_ms = 0, wait_ms = 0, copy_ms = 0;     // (SCFQ_VERBOSE: where the orchestrating thread spent the call)
  unsigned n_chunks = 0;
  auto ms_since = [](clk::time_point t) { return std::chrono::duration<double, std::milli>(clk::now() - t).count(); };
  // (r5) The member walk runs AHEAD on a thread of its own.  It reads two pages of the mapped file per member — a header, a trailer —
  // and pays a page fault for each: 23 ms for the 92 000 members of a 6 GB file, which the orchestrating thread used to spend between
  // two chunks' copies with the device idle (profiles/r05/bgzf_timeline.txt).  The plans are the same ones, made in the same order.
  struct Plan { int64_t used = 0; uint32_t nb = 0; uint64_t ob = 0; std::vector<scfq_dinflate::Block> blocks; };
  struct Walker {
    std::mutex mu; std::condition_variable cv; std::deque<Plan> q; bool done = false, stop = false; std::thread th;
    ~Walker() { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv.notify_all(); if (th.joinable()) th.join(); }
  } walker;
  {
    uint64_t big_inf = 0, big_comp = 0;
    bgzf_caps_for(c, fsize, &big_inf, &big_comp);
    (void)bgzf_members_per_launch(big_inf);            // (asks the runtime for the kernel's occupancy once, on this thread)
    const uint64_t first_inf = c->inf_first_cap, first_comp = c->comp_first_cap;
    walker.th = std::thread([&walker, img, fsize, big_inf, big_comp, first_inf, first_comp] {
      uint64_t at = 0;
      for (unsigned it = 0; at < fsize; ++it) {
        const bool first = it == 0;
        const uint64_t chunk = first ? first_inf : big_inf, comp_chunk = first ? first_comp : big_comp;
        const uint32_t members = first ? (uint32_t)(kBgzfFirstInflated >> 16) : bgzf_members_per_launch(chunk, (int)std::min(it, 3u));
        Plan pl;
        pl.blocks.resize(std::min<uint64_t>(members, kMaxBlocksPerChunk));
        pl.used = bgzf_plan(img, fsize, at, chunk, comp_chunk, (uint32_t)pl.blocks.size(), pl.blocks.data(), &pl.nb, &pl.ob);
        const int64_t used = pl.used;
        {
          std::unique_lock<std::mutex> lk(walker.mu);
          walker.cv.wait(lk, [&] { return walker.stop || walker.q.size() < 4; });
          if (walker.stop) return;
          walker.q.push_back(std::move(pl));
        }
        walker.cv.notify_all();
        if (used <= 0) break;
        at += (uint64_t)used;
      }
      { std::lock_guard<std::mutex> lk(walker.mu); walker.done = true; }
      walker.cv.notify_all();
    });
  }
  for (unsigned it = 0; pos < fsize; ++it) {
    const int b = it & 1;
    { const auto tw = clk::now(); if (it >= 2) HIPCHK(hipEventSynchronize(c->ev_copied[b])); wait_ms += ms_since(tw); }      // pinned table b is free again
    if (it == 1 && alloc.th.joinable()) {
      alloc.th.join();
      if (alloc_rc) {
        // no room for the big buffers (many concurrent sessions): the first chunk is in flight — drained here, and the caller starts
        // the session again on the host path (kNotPureBgzf's contract)
        (void)hipStreamSynchronize(c->copy);
        (void)hipStreamSynchronize(c->compute);
        return alloc_rc == kFallbackToHost ? kNotPureBgzf : alloc_rc;
      }
    }
    const bool first = it == 0;
    if (it == 1 && (rc = want_copy_stream(c))) return rc;      // a second chunk: its copy runs under the first one's inflate and scan
    const uint64_t chunk = first ? c->inf_first_cap : c->inf_cap, comp_chunk = first ? c->comp_first_cap : c->comp_cap;
    uint8_t* const d_comp = first ? c->d_comp_first : c->d_comp[b];
    uint8_t* const d_inf = first ? c->d_inf_first : c->d_inf[b];
    const auto tf = clk::now();
    uint32_t nb = 0;
    uint64_t ob = 0;
    int64_t used = 0;
    {
      Plan pl;
      {
        std::unique_lock<std::mutex> lk(walker.mu);
        walker.cv.wait(lk, [&] { return !walker.q.empty() || walker.done; });
        if (walker.q.empty()) break;                   // (the walk ended exactly at the file's end)
        pl = std::move(walker.q.front());
        walker.q.pop_front();
      }
      walker.cv.notify_all();
      used = pl.used; nb = pl.nb; ob = pl.ob;
      if (nb) std::memcpy(c->h_blk[b], pl.blocks.data(), (size_t)nb * sizeof(scfq_dinflate::Block));
    }
    (void)chunk; (void)comp_chunk;
    plan_ms += ms_since(tf);
    ++n_chunks;
    if (used < 0) {                                    // not a BGZF member, or a truncated one: the host path decides what it is
      HIPCHK(hipStreamSynchronize(c->copy));
      HIPCHK(hipStreamSynchronize(c->compute));
      return kNotPureBgzf;
    }
    if (used == 0) break;
    if (first && (uint64_t)used < fsize) {
      // more than one chunk: the big buffers, sized by the file, on a helper thread under this chunk's copy and decode
      const int dev = c->dev;
      alloc.th = std::thread([c, fsize, dev, &alloc_rc] {
        if (hipSetDevice(dev) != hipSuccess) { alloc_rc = SCFQ_EHIP; return; }
        alloc_rc = ensure_bgzf_device_buffers(c, fsize, true);
      });
    }
    c->timing.h2d_bytes += (uint64_t)used;
    // device buffers b were consumed.  Chunk 0 had compressed / inflated buffers of its own but shares the block table d_blk[0] with
    // chunk 2: the wait holds from the third chunk on (chunk 0's inflate finished long before — the wait costs nothing — but nothing
    // else orders chunk 2's table copy behind chunk 0's kernel when the compute stream is held up by other sessions)
    if (it >= 2) HIPCHK(hipStreamWaitEvent(c->copy, c->ev_scanned[b], 0));
    if (timing) {
      while (c->cp_pool.size() < c->cp_used + 2) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); c->cp_pool.push_back(e); }
      HIPCHK(hipEventRecord(c->cp_pool[c->cp_used], c->copy));
    }
    { const auto tc = clk::now(); if ((rc = copy_through_ring(c, d_comp, fbytes, pos, (uint64_t)used))) return rc; copy_ms += ms_since(tc); }
    fill_ms += std::chrono::duration<double, std::milli>(clk::now() - tf).count();
    HIPCHK(hipMemcpyAsync(c->d_blk[b], c->h_blk[b], nb * sizeof(scfq_dinflate::Block), hipMemcpyHostToDevice, c->copy));
    if (timing) { HIPCHK(hipEventRecord(c->cp_pool[c->cp_used + 1], c->copy)); c->cp_used += 2; }
    HIPCHK(hipEventRecord(c->ev_copied[b], c->copy));
    if (first) trace("BGZF: first chunk's compressed bytes queued for the device");
    HIPCHK(hipStreamWaitEvent(c->compute, c->ev_copied[b], 0));
    uint8_t* base = d_inf + kStagePad;
    if (prev_base) HIPCHK(hipMemcpyAsync(base - 1, prev_base + prev_n - 1, 1, hipMemcpyDeviceToDevice, c->compute));
    if (warm.th.joinable()) warm.th.join();
    if (nb) {
      hipLaunchKernelGGL(scfq_dinflate::bgzf_inflate, dim3((nb + scfq_dinflate::kWavesPerWg - 1) / scfq_dinflate::kWavesPerWg),
                         dim3(64 * scfq_dinflate::kWavesPerWg), scfq_dinflate::kWavesPerWg * scfq_dinflate::kWaveLdsBytes, c->compute,
                         d_comp, c->d_blk[b], nb, base, c->d_dstatus, inflate_serial_loop());
      HIPCHK(hipGetLastError());
      // every member's CRC-32 trailer against its bytes (a kernel of its own since r5: bgzf_inflate_kernel.hpp)
      if ((rc = bgzf_crc_ready(c))) return rc;
      hipLaunchKernelGGL(scfq_dinflate::bgzf_crc32_members, dim3(nb), dim3(256), 0, c->compute, base, c->d_blk[b], nb, c->d_dstatus);
      HIPCHK(hipGetLastError());
      if (first) trace("BGZF: first inflate kernel queued");
      // measurement aid: one line per bgzf_inflate dispatch (members, compressed bytes, inflated bytes), in dispatch order, so
      // that a rocprofv3 kernel trace of the same run can be priced in GB/s per dispatch (scripts/gpu_profile_inflate.sh)
      if (const char* lp = std::getenv("SCFQ_BGZF_LAUNCH_LOG")) {
        if (FILE* lf = std::fopen(lp, "a")) { std::fprintf(lf, "%u\t%llu\t%llu\n", nb, (unsigned long long)used, (unsigned long long)ob); std::fclose(lf); }
      }
    }
    if (ob) {
      rc = scan_async(c, base, ob, prev_base ? -2 : first_prev, flags & ~SCFQ_PREV_IN_MEMORY, timing);
      if (rc) return rc;
      prev_base = base;
      prev_n = ob;
    }
    HIPCHK(hipEventRecord(c->ev_scanned[b], c->compute));
    pos += (uint64_t)used;
  }
  uint32_t st = 0;
  HIPCHK(hipMemcpyAsync(c->h_state + kStateWords - 1, c->d_dstatus, sizeof(uint32_t), hipMemcpyDeviceToHost, c->compute));
  const auto t_drain = clk::now();
  if (bytes_left_host) {
    { std::unique_lock<std::mutex> lk(walker.mu); walker.cv.wait(lk, [&] { return walker.done; }); }      // (the walk reads the mapping; it has ended when the last plan was taken)
    bytes_left_host();
  }
  HIPCHK(hipStreamSynchronize(c->compute));
  if (trace_on())
    std::fprintf(stderr, "scfq bgzf: %u chunk(s), %.2f GB compressed; orchestrating thread: waits for the member walk %.1f ms, bytes to the device %.1f ms, waits for a table %.1f ms, "
                         "drain behind the last launch %.1f ms; wall %.1f ms\n", n_chunks, (double)pos / 1e9, plan_ms, copy_ms, wait_ms, ms_since(t_drain), ms_since(t_begin));
  std::memcpy(&st, c->h_state + kStateWords - 1, sizeof st);
  if (st) { std::snprintf(g_err, sizeof g_err, "device inflate: error mask 0x%x (2 = corrupt deflate data, 4 = length, 8 = CRC-32)", st); return SCFQ_EGZ; }
  return SCFQ_OK;
}

#include "scfq_gzdev.hpp"      // ingest_gz_device: ordinary gzip members inflated on the device

int partial_on_current_device(const void* ptr, uint64_t n, int is_device, int prev_byte, const scfq_opts* opts,
                              scfq_partial* out, uint64_t* hist) {
  Ctx* c = nullptr;
  SessionLock sl;
  int rc = get_ctx(&c, sl);
  if (rc) return rc;
  const uint32_t flags = opt_flags(opts);
  const bool timing = flags & SCFQ_TIMING;
  rc = begin_session(c, prev_byte == -1 && !(flags & SCFQ_PREV_IN_MEMORY));   // -1: the range starts the input
  if (rc) return rc;
  if (is_device && (rc = wait_for_caller(c, opts))) return rc;
  if (is_device) {
    const int prev = (flags & SCFQ_PREV_IN_MEMORY) ? -2 : prev_byte;
    rc = scan_async(c, static_cast<const uint8_t*>(ptr), n, prev, flags, timing);
  } else {
    int prev = prev_byte;
    if (flags & SCFQ_PREV_IN_MEMORY) prev = static_cast<const uint8_t*>(ptr)[-1];
    MemSource src(static_cast<const uint8_t*>(ptr), n);
    rc = ingest(c, src, prev, flags, std::min<uint64_t>(opt_chunk(opts), std::max<uint64_t>((n + 4095) & ~4095ull, 4096)), timing);
  }
  if (rc) return rc;
  return end_session(c, flags & SCFQ_QUAL_HIST, out, hist);
}

}  // namespace

extern "C" {

const char* scfq_last_error_detail(void) { return g_err; }

int scfq_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    std::snprintf(g_err, sizeof g_err, "hipGetDeviceCount -> %s", hipGetErrorString(e));
    return SCFQ_EHIP;
  }
  return n;
}

int scfq_set_wait_stream(void* hip_stream, int enable) { g_wait_stream = enable ? hip_stream : nullptr; g_wait_enabled = enable != 0; return SCFQ_OK; }
void* scfq_get_wait_stream(int* enabled) { if (enabled) *enabled = g_wait_enabled ? 1 : 0; return g_wait_stream; }

int scfq_debug_hist_stats(uint64_t* fast_ranges, uint64_t* redone_ranges) {
  if (fast_ranges) *fast_ranges = g_hist_stats[0];
  if (redone_ranges) *redone_ranges = g_hist_stats[1];
  return SCFQ_OK;
}

// the stage marks of this process so far as one JSON array of [name, ms since the library was loaded]; returns the length written
// (without the terminator), or the length needed when cap is too small (nothing is written then)
int64_t scfq_debug_stages(char* buf, uint64_t cap) {
  std::string js = "[";
  {
    StageLog& g = stage_log();
    std::lock_guard<std::mutex> lk(g.mu);
    for (size_t k = 0; k < g.v.size(); ++k) {
      char num[48];
      std::snprintf(num, sizeof num, "\", %.2f]", g.v[k].second);
      js += k ? ", [\"" : "[\"";
      for (char ch : g.v[k].first) if (ch != '"' && ch != '\\' && (unsigned char)ch >= 32) js += ch;
      js += num;
    }
  }
  js += "]";
  if (buf && cap > js.size()) std::memcpy(buf, js.c_str(), js.size() + 1);
  return (int64_t)js.size();
}
void scfq_debug_stage_mark(const char* what) { if (what) trace(what); }
int64_t scfq_debug_last_scan_kernel(char* buf, uint64_t cap) {
  const size_t n = std::strlen(g_last_scan);
  if (buf && cap > n) std::memcpy(buf, g_last_scan, n + 1);
  return (int64_t)n;
}

uint64_t scfq_device_bytes_high_water(void) { return g_dev_high.load(); }
uint64_t scfq_device_bytes_now(void) { return g_dev_bytes.load(); }

int scfq_last_timing(scfq_timing* t) {
  if (!t || t->struct_size != sizeof(scfq_timing)) return SCFQ_EARG;
  *t = g_last_timing;
  t->struct_size = sizeof(scfq_timing);
  return SCFQ_OK;
}

int scfq_partial_buffer(const void* ptr, uint64_t n, int is_device, int prev_byte, const scfq_opts* opts,
                        scfq_partial* out, uint64_t* hist) {
  if (!out || (!ptr && n)) return SCFQ_EARG;
  if (prev_byte < -1 || prev_byte > 255) return SCFQ_EARG;
  int rc = check_opts(opts);
  if (rc) return rc;
  return partial_on_current_device(ptr, n, is_device, prev_byte, opts, out, hist);
}

// A shard whose speculative histogram turns out to have backed the wrong class (possible only for a malformed input cut
// into shards: SCFQ_ESPEC from combine / finalize) is counted again with the exact histogram kernel.
static int count_buffer_once(const void* ptr, uint64_t n, int is_device, const scfq_opts* opts, scfq_counts* out);
static int count_file_once(const char* path, const scfq_opts* opts, scfq_counts* out);
static int count_file_partial(const char* path, const scfq_opts* opts, scfq_partial* p_out, uint64_t* hist_out);

int scfq_count_buffer(const void* ptr, uint64_t n, int is_device, const scfq_opts* opts, scfq_counts* out) {
  int rc = count_buffer_once(ptr, n, is_device, opts, out);
  if (rc == SCFQ_ESPEC) {
    scfq_opts o = opts_copy(opts);     // ESPEC implies opts with SCFQ_QUAL_HIST
    o.flags |= SCFQ_HIST_EXACT;
    rc = count_buffer_once(ptr, n, is_device, &o, out);
  }
  return rc;
}

int scfq_count_file(const char* path, const scfq_opts* opts, scfq_counts* out) {
  int rc = count_file_once(path, opts, out);
  if (rc == SCFQ_ESPEC) {
    scfq_opts o = opts_copy(opts);
    o.flags |= SCFQ_HIST_EXACT;
    rc = count_file_once(path, &o, out);
  }
  return rc;
}

static int count_buffer_once(const void* ptr, uint64_t n, int is_device, const scfq_opts* opts, scfq_counts* out) {
  if (!out || out->struct_size != sizeof(scfq_counts) || (!ptr && n)) return SCFQ_EARG;
  int rc = check_opts(opts);
  if (rc) return rc;
  scfq_opts o = opts_copy(opts);
  o.flags &= ~SCFQ_PREV_IN_MEMORY;
  const bool want_hist = o.flags & SCFQ_QUAL_HIST;
  std::vector<uint64_t> hist(want_hist ? SCFQ_HIST_WORDS : 0);
  scfq_partial p;
  if (!is_device && o.n_devices > 1 && n >= (uint64_t)o.n_devices * scfq::kTile) {
    // byte-range shards, one per device, arbitrary (unaligned) cut points; ordered host fold
    const int nd = o.n_devices;
    std::vector<scfq_partial> parts(nd);
    std::vector<std::vector<uint64_t>> hists(nd, std::vector<uint64_t>(want_hist ? SCFQ_HIST_WORDS : 0));
    std::vector<int> rcs(nd, 0);
    std::vector<std::thread> th;
    const uint8_t* base = static_cast<const uint8_t*>(ptr);
    for (int d = 0; d < nd; ++d) {
      th.emplace_back([&, d] {
        if (hipSetDevice(o.device_ids[d]) != hipSuccess) { rcs[d] = SCFQ_EHIP; return; }
        const uint64_t lo = n * (uint64_t)d / nd, hi = n * (uint64_t)(d + 1) / nd;
        scfq_opts od = o;
        od.n_devices = 0;
        rcs[d] = partial_on_current_device(base + lo, hi - lo, 0, lo ? base[lo - 1] : -1, &od, &parts[d],
                                           want_hist ? hists[d].data() : nullptr);
      });
    }
    for (auto& t : th) t.join();
    for (int d = 0; d < nd; ++d) if (rcs[d]) return rcs[d];
    rc = fold_device_partials(o, nd, parts, hists, want_hist, &p, want_hist ? hist.data() : nullptr);
    if (rc) return rc;
  } else {
    if (!is_device && o.n_devices >= 1) HIPCHK(hipSetDevice(o.device_ids[0]));
    rc = partial_on_current_device(ptr, n, is_device, -1, &o, &p, want_hist ? hist.data() : nullptr);
    if (rc) return rc;
  }
  return scfq_partial_finalize(&p, want_hist ? hist.data() : nullptr, out);
}

static int count_file_once(const char* path, const scfq_opts* opts, scfq_counts* out) {
  trace("scfq_count_file entered");
  if (!path || !out || out->struct_size != sizeof(scfq_counts)) return SCFQ_EARG;
  int rc = check_opts(opts);
  if (rc) return rc;
  const bool want_hist = opt_flags(opts) & SCFQ_QUAL_HIST;
  std::vector<uint64_t> hist(want_hist ? SCFQ_HIST_WORDS : 0);
  scfq_partial p;
  rc = count_file_partial(path, opts, &p, want_hist ? hist.data() : nullptr);
  if (rc) return rc;
  return scfq_partial_finalize(&p, want_hist ? hist.data() : nullptr, out);
}

// The partial of a whole file folded from its first byte (source selection: plain pread / BGZF on the device or the host /
// the library's gzip readers), before finalisation.  hist_out: uint64_t[SCFQ_HIST_WORDS] when SCFQ_QUAL_HIST is set.
static int count_file_partial(const char* path, const scfq_opts* opts, scfq_partial* p_out, uint64_t* hist_out) {
  int rc = SCFQ_OK;
  scfq_opts o = opts_copy(opts);
  o.flags &= ~SCFQ_PREV_IN_MEMORY;
  const bool want_hist = o.flags & SCFQ_QUAL_HIST;
  const bool timing = o.flags & SCFQ_TIMING;
  std::vector<uint64_t> hist(want_hist ? SCFQ_HIST_WORDS : 0);
  scfq_partial p;
  struct Hand {      // hands the folded partial to the caller on every successful return
    scfq_partial* p; std::vector<uint64_t>* h; scfq_partial* po; uint64_t* ho;
    ~Hand() { *po = *p; if (ho && !h->empty()) std::memcpy(ho, h->data(), SCFQ_HIST_WORDS * sizeof(uint64_t)); }
  } hand{&p, &hist, p_out, hist_out};
  scfq_partial_identity(&p, nullptr);
  const size_t plen = std::strlen(path);
  // fastq[^3 .. ^1] == ".gz"      src/fq_count.nim:31 (case-sensitive, last three bytes)
  const bool is_gz = plen >= 3 && std::memcmp(path + plen - 3, ".gz", 3) == 0;
  if (is_gz) {
    {   // BGZF (bgzip) files: block-parallel inflate; every other gzip layout: serial gzread below
      const int bfd = open(path, O_RDONLY);
      struct stat bsb;
      if (bfd >= 0 && fstat(bfd, &bsb) == 0 && S_ISREG(bsb.st_mode) && !std::getenv("SCFQ_NO_BGZF") && scfq_bgzf::probe(bfd)) {
        if (o.n_devices >= 1 && hipSetDevice(o.device_ids[0]) != hipSuccess) { close(bfd); return SCFQ_EHIP; }
        void* m = MAP_FAILED;
        if (bgzf_device_enabled() && bsb.st_size > 0) {
          m = mmap(nullptr, (size_t)bsb.st_size, PROT_READ, MAP_PRIVATE, bfd, 0);
          if (m != MAP_FAILED) (void)madvise(m, (size_t)bsb.st_size, MADV_SEQUENTIAL);
        }
        Ctx* c = nullptr;
        SessionLock sl;
        rc = get_ctx(&c, sl);
        if (!rc) rc = begin_session(c, true);
        bool on_device = false;
        if (!rc && bgzf_device_enabled() && bsb.st_size > 0) {
          // pure BGZF (every member <= 64 KiB with its size in the header): compressed bytes over PCIe, inflate on the device
          if (m != MAP_FAILED) {
            // (no purity walk up front: touching every member header of a mapped 1 GB file costs 15 ms of page faults;
            // the chunk planner walks them anyway, under the device's work, and reports what it cannot take)
            rc = ingest_bgzf_device(c, static_cast<const uint8_t*>(m), (uint64_t)bsb.st_size, o.flags, opt_chunk(&o), timing, -1, bfd, 0,
                                    [&] { munmap(m, (size_t)bsb.st_size); m = MAP_FAILED; });
            on_device = (rc != kFallbackToHost && rc != kNotPureBgzf);
            if (rc == kNotPureBgzf) rc = begin_session(c, true);      // drop what the device path accumulated
            else if (!on_device) rc = SCFQ_OK;
          }
        }
        if (m != MAP_FAILED) munmap(m, (size_t)bsb.st_size);
        if (!rc && !on_device) {
          BgzfSource src(bfd, (uint64_t)bsb.st_size);
          rc = ingest(c, src, -1, o.flags, opt_chunk(&o), timing);
        }
        close(bfd);
        if (rc) return rc;
        rc = end_session(c, want_hist, &p, want_hist ? hist.data() : nullptr);
        if (rc) return rc;
        return SCFQ_OK;
      }
      if (bfd >= 0) close(bfd);
    }
    // ordinary gzip members: inflate on the device (compressed bytes over PCIe); anything the device path cannot prove
    // consistent — and small files, FIFOs — goes to the host readers below, which are gzread byte for byte
    if (gz_device_enabled()) {
      const int gfd = open(path, O_RDONLY);
      struct stat gsb;
      static const uint64_t min_bytes = (uint64_t)std::max(0, env_int("SCFQ_GZ_DEVICE_MIN_MB", 4)) << 20;
      if (gfd >= 0 && fstat(gfd, &gsb) == 0 && S_ISREG(gsb.st_mode) && (uint64_t)gsb.st_size >= std::max<uint64_t>(min_bytes, 64)) {
        void* m = mmap(nullptr, (size_t)gsb.st_size, PROT_READ, MAP_PRIVATE, gfd, 0);
        if (m != MAP_FAILED) {
          (void)madvise(m, (size_t)gsb.st_size, MADV_SEQUENTIAL);
          struct Unmap { void* m; size_t n; int fd; ~Unmap() { munmap(m, n); close(fd); } } um{m, (size_t)gsb.st_size, gfd};
          if (o.n_devices >= 1 && hipSetDevice(o.device_ids[0]) != hipSuccess) return SCFQ_EHIP;
          Ctx* c = nullptr;
          SessionLock sl;
          rc = get_ctx(&c, sl);
          if (!rc) rc = begin_session(c, true);
          if (!rc) rc = ingest_gz_device(c, static_cast<const uint8_t*>(m), (uint64_t)gsb.st_size, o.flags, timing, nullptr, gfd, 0);
          if (rc == SCFQ_OK) return end_session(c, want_hist, &p, want_hist ? hist.data() : nullptr);
          if (rc != kFallbackToHost) return rc;
          (void)hipStreamSynchronize(c->compute);
          (void)hipStreamSynchronize(c->copy);
          rc = SCFQ_OK;
        } else {
          close(gfd);
        }
      } else if (gfd >= 0) {
        close(gfd);
      }
    }
    gzFile f = nullptr;
    std::unique_ptr<Source> gsrc = open_gz_source(path, opt_chunk(&o), &f);
    if (!gsrc) return SCFQ_EOPEN;
    struct GzCloser { gzFile* f; std::unique_ptr<Source>* s; ~GzCloser() { s->reset(); if (*f) gzclose(*f); } } gzc{&f, &gsrc};
    if (o.n_devices >= 1 && hipSetDevice(o.device_ids[0]) != hipSuccess) return SCFQ_EHIP;
    Ctx* c = nullptr;
    SessionLock sl;
    rc = get_ctx(&c, sl);
    if (!rc) rc = begin_session(c, true);
    if (!rc) rc = ingest(c, *gsrc, -1, o.flags, opt_chunk(&o), timing);
    if (rc) return rc;
    rc = end_session(c, want_hist, &p, want_hist ? hist.data() : nullptr);
    if (rc) return rc;
    return SCFQ_OK;
  }
  int fd = open(path, O_RDONLY);
  if (fd < 0) return SCFQ_EOPEN;
  struct stat sb;
  if (fstat(fd, &sb) != 0 || S_ISDIR(sb.st_mode)) { close(fd); return SCFQ_EOPEN; }
  const bool regular = S_ISREG(sb.st_mode);
  const uint64_t size = regular ? (uint64_t)sb.st_size : 0;
  // SCFQ_EXCHANGE_AT_1=1 (rehearsal on a one-GPU box): a one-device list takes the sharded path too, exchange included
  static const bool at1 = env_int("SCFQ_EXCHANGE_AT_1", 0) != 0;
  const int nd = (regular && o.n_devices > 1 && size >= (uint64_t)o.n_devices * (1u << 20)) ? o.n_devices : 1;
  if (nd > 1 || (at1 && regular && o.n_devices == 1)) {
    std::vector<scfq_partial> parts(nd);
    std::vector<std::vector<uint64_t>> hists(nd, std::vector<uint64_t>(want_hist ? SCFQ_HIST_WORDS : 0));
    std::vector<int> rcs(nd, 0);
    std::vector<std::thread> th;
    for (int d = 0; d < nd; ++d) {
      th.emplace_back([&, d] {
        if (hipSetDevice(o.device_ids[d]) != hipSuccess) { rcs[d] = SCFQ_EHIP; return; }
        const uint64_t lo = size * (uint64_t)d / nd, hi = size * (uint64_t)(d + 1) / nd;
        int prev = -1;
        if (lo) { uint8_t pb; if (pread(fd, &pb, 1, (off_t)(lo - 1)) != 1) { rcs[d] = SCFQ_EIO; return; } prev = pb; }
        Ctx* c = nullptr;
        SessionLock sl;
        int r = get_ctx(&c, sl);
        if (!r) r = begin_session(c, lo == 0);
        if (!r) { FdSource src(fd, lo, hi); r = ingest(c, src, prev, o.flags, opt_chunk(&o), timing); }
        if (!r) r = end_session(c, want_hist, &parts[d], want_hist ? hists[d].data() : nullptr);
        rcs[d] = r;
      });
    }
    for (auto& t : th) t.join();
    close(fd);
    for (int d = 0; d < nd; ++d) if (rcs[d]) return rcs[d];
    rc = fold_device_partials(o, nd, parts, hists, want_hist, &p, want_hist ? hist.data() : nullptr);
    if (rc) return rc;
    return SCFQ_OK;
  }
  if (o.n_devices >= 1 && hipSetDevice(o.device_ids[0]) != hipSuccess) { close(fd); return SCFQ_EHIP; }
  Ctx* c = nullptr;
  SessionLock sl;
  rc = get_ctx(&c, sl);
  if (!rc) rc = begin_session(c, true);
  if (!rc) {
    if (regular) {
      FdSource src(fd, 0, size);
      rc = ingest(c, src, -1, o.flags, std::min<uint64_t>(opt_chunk(&o), std::max<uint64_t>((size + 4095) & ~4095ull, 4096)), timing);
    } else {
      // FIFO / character device: sequential read()
      struct SeqSource : Source {
        int fd;
        explicit SeqSource(int f) : fd(f) {}
        int64_t fill(uint8_t* dst, uint64_t cap) override {
          uint64_t got = 0;
          while (got < cap) {
            ssize_t r = read(fd, dst + got, cap - got);
            if (r < 0) return SCFQ_EIO;
            if (r == 0) break;
            got += (uint64_t)r;
          }
          return (int64_t)got;
        }
      } src(fd);
      rc = ingest(c, src, -1, o.flags, opt_chunk(&o), timing);
    }
  }
  close(fd);
  if (rc) return rc;
  rc = end_session(c, want_hist, &p, want_hist ? hist.data() : nullptr);
  if (rc) return rc;
  return SCFQ_OK;
}

}  // extern "C"
namespace {
// BGZF shards.  A BGZF file is cut where its members are: the first position at or after `from` where eight members follow one
// another (or a shorter run that ends exactly with the file) — every rank finds the SAME positions with this rule, from the bytes
// alone.  n when there is none.
uint64_t bgzf_boundary(const uint8_t* img, uint64_t n, uint64_t from) {
  for (uint64_t p = from; p + 18 <= n;) {
    const void* hit = std::memchr(img + p, 0x1f, (size_t)(n - 17 - p));
    if (!hit) break;
    p = (uint64_t)(static_cast<const uint8_t*>(hit) - img);
    uint64_t q = p;
    int k = 0;
    while (k < 8 && q < n) {
      uint32_t hl = 0;
      const uint32_t bs = scfq_bgzf::block_size(img + q, n - q, &hl);
      if (!bs || q + bs > n || scfq_bgzf::rd32(img + q + bs - 4) > (1u << 16)) break;
      q += bs;
      ++k;
    }
    if (k == 8 || (k > 0 && q == n)) return p;
    ++p;
  }
  return n;
}
// Where rank r's members start (r > 0) and the byte in front of its first inflated byte: the boundary rule gives a member; the rank's
// range starts BEHIND the first non-empty member from there, which the rank inflates on the host (one member of at most 64 KiB)
// for its last byte.  Both neighbours compute the same cut.  false: a member that does not inflate (the file is damaged).
bool bgzf_cut(const uint8_t* img, uint64_t n, uint64_t from, uint64_t* cut, int* prev) {
  uint64_t p = bgzf_boundary(img, n, from);
  *prev = -1;
  while (p < n) {
    uint32_t hl = 0;
    const uint32_t bs = scfq_bgzf::block_size(img + p, n - p, &hl);
    if (!bs || p + bs > n) { *cut = n; return false; }
    const uint32_t isize = scfq_bgzf::rd32(img + p + bs - 4);
    if (isize == 0) { p += bs; continue; }               // (an empty member — the end-of-file marker — has no last byte)
    if (isize > (1u << 16)) { *cut = n; return false; }
    std::vector<scfq_bgzf::Block> one{{p, bs, hl, isize, scfq_bgzf::rd32(img + p + bs - 8), 0}};
    std::vector<uint8_t> out(isize);
    if (scfq_bgzf::inflate_blocks(img, one, 0, 1, out.data())) { *cut = n; return false; }
    *prev = out[isize - 1];
    *cut = p + bs;
    return true;
  }
  *cut = n;
  return true;
}
// ---- ordinary gzip shards: a file of SEVERAL members is cut where members start ------------------------------------------------
// (`cat a.fq.gz b.fq.gz`, `pigz -i`, per-lane or per-tile members of a sequencer's writer.)  Unlike BGZF a member does not say how long
// it is, so a rank that starts in the middle of the file can only LOOK for a member start: the three magic bytes with no reserved flag
// bit set, a header that parses, and deflate data that inflates cleanly for its first 64 KiB — a block header made of chance bits
// survives the Huffman-code tests about once in 4000 tries and then dies within a few hundred symbols.  That makes a false start
// unlikely, not impossible; what PROVES a cut is the rank before it: its members must end — trailer, CRC-32 and ISIZE checked —
// exactly where the next rank began (gz_shard_fold below), or every rank falls back to rank 0 reading the whole file.
//
// gz_member_here: true when a member demonstrably starts at p.  *first_byte: the first byte it (or, when it is empty, a member
// behind it, inside [p, stop)) inflates to; -1 when there is none in that stretch.
bool gz_member_here(const uint8_t* img, uint64_t n, uint64_t p, uint64_t stop, int* first_byte) {
  *first_byte = -1;
  std::vector<uint8_t> buf;
  bool first = true;
  while (p < n && (first || p < stop)) {
    if (n - p < 18 || img[p] != 0x1f || img[p + 1] != 0x8b || img[p + 2] != 8 || (img[p + 3] & 0xE0)) return !first;
    const long h = scfq_gzfast::member_header(img + p, (size_t)(n - p));
    if (h <= 0) return !first;
    const uint64_t sample = std::min<uint64_t>(n - (p + (uint64_t)h), 64u << 10);
    if (buf.empty()) buf.resize(scfq_gzfast::kWindow + (2u << 20));
    auto dec = std::unique_ptr<scfq_inflate::Decoder>(new scfq_inflate::Decoder());
    dec->begin(img + p + h, img + p + h + sample);
    uint8_t* o = buf.data() + scfq_gzfast::kWindow;
    const int r = dec->run(o, buf.data() + buf.size());
    const uint64_t got = (uint64_t)(o - (buf.data() + scfq_gzfast::kWindow));
    if (r == scfq_inflate::kErrData) return !first;
    if (r == scfq_inflate::kErrTruncated && sample == n - (p + (uint64_t)h)) return !first;      // (the FILE ends inside the member: damaged)
    if (got) { *first_byte = buf[scfq_gzfast::kWindow]; return true; }
    if (r != scfq_inflate::kStreamEnd) return true;            // (no byte yet and no end either: a long run of empty stored blocks; rare, harmless)
    // an empty member: the first byte is a later member's
    const uint8_t* t = dec->end_of_stream();
    first = false;
    p = (uint64_t)(t - img) + 8;
  }
  return true;
}
// the first demonstrable member start at or after `from`; n when there is none.  *first_byte as above (stop: the end of the rank's stretch)
// (limit: only starts in front of this offset are looked for — a rank that only wants to know what its own share of the file holds does
// not walk a 25 GB member to its end)
uint64_t gz_member_boundary(const uint8_t* img, uint64_t n, uint64_t from, uint64_t stop, int* first_byte, uint64_t limit = ~0ull) {
  *first_byte = -1;
  const uint64_t last = std::min<uint64_t>(limit, n >= 17 ? n - 17 : 0);       // first offset that is no candidate any more
  for (uint64_t p = from; p < last;) {
    const void* hit = std::memchr(img + p, 0x1f, (size_t)(last - p));
    if (!hit) break;
    p = (uint64_t)(static_cast<const uint8_t*>(hit) - img);
    if (img[p + 1] == 0x8b && img[p + 2] == 8 && !(img[p + 3] & 0xE0) && gz_member_here(img, n, p, std::max(stop, p + 1), first_byte)) return p;
    ++p;
  }
  return n;
}

// A shard that was scanned as if it began the input (no byte before it is known when its scan starts: that byte is the LAST one the
// member before inflates to), put right once that byte is known.  Two things depend on it: a '\n' at the shard's first position ends
// a line whose '\r' — if the byte before is one — is not part of that line (len, and the quality histogram's '\r' bin, are taken back
// exactly as a range of the device path takes back a '\r' that lies in the range before it: u64 modular); and the shard's first byte
// starts a line only when the byte before is a '\n' (K4's line starts, and what they begin with).
void gz_shard_fix(scfq_partial* p, uint64_t* hist, int true_prev, int first_byte, uint32_t flags) {
  if (p->bytes == 0 || true_prev < 0 || first_byte < 0) return;
  if (first_byte == '\n' && true_prev == '\r') {
    p->len[0] -= 1;
    if (hist && (p->hist_class == 0 || p->hist_class == 1)) hist[0 * 256 + 13] -= 1;
  }
  if ((flags & SCFQ_STRUCT_CHECK) && true_prev != '\n') {
    p->starts[0] -= 1;
    if (first_byte == '@') p->first_at[0] -= 1;
    if (first_byte == '+') p->first_plus[0] -= 1;
  }
}
// ---- ONE member over several ranks: the deflate stream is cut where BLOCKS start -----------------------------------------------------
// gz_block_boundary: the first bit at or after byte `from` where a dynamic-Huffman block demonstrably starts (scfq_pgz.hpp's test: a
// header that parses — complete code-length, literal/length and distance codes — and 4096 symbols that decode cleanly); 0 when there is
// none within `span` bytes.  Both neighbours of a cut compute it from the same bytes.  What proves it is the rank before: its chain must
// arrive at exactly this bit (GzStretch: stop_bit), or every rank falls back to rank 0 reading the whole file.
uint64_t gz_block_boundary(const uint8_t* img, uint64_t n, uint64_t from, uint64_t span) {
  const uint64_t to = std::min<uint64_t>(from + span, n > 16 ? n - 16 : 0);
  if (from >= to) return 0;
  auto d = std::unique_ptr<scfq_inflate::Decoder>(new scfq_inflate::Decoder());
  std::vector<uint16_t> scratch(scfq_pgz::kWindow + scfq_pgz::kTrialSymbols + 2 * scfq_inflate::kOutSlack);
  for (uint32_t i = 0; i < scfq_pgz::kWindow; ++i) scratch[i] = (uint16_t)(0x8000u | i);
  for (uint64_t b = from * 8; b < to * 8; ++b)
    if (scfq_pgz::plausible_block(*d, img, img + n, b, scratch)) return b;
  return 0;
}
}  // namespace
extern "C" {

// fq_count of one file by all ranks of a communicator (include/sc_fqcount.h): byte-range shard -> K1/K2 -> exchange -> fold
int scfq_count_file_sharded(const char* path, const scfq_opts* opts, scfq_comm* comm, scfq_counts* out) {
  if (!path || !comm || !out || out->struct_size != sizeof(scfq_counts)) return SCFQ_EARG;
  int rc = check_opts(opts);
  if (rc) return rc;
  scfq_opts o = opts_copy(opts);
  o.flags &= ~SCFQ_PREV_IN_MEMORY;
  const bool want_hist = o.flags & SCFQ_QUAL_HIST;
  const bool timing = o.flags & SCFQ_TIMING;
  const int world = scfq_comm_world(comm), rank = scfq_comm_rank(comm);
  std::vector<uint64_t> hist(want_hist ? SCFQ_HIST_WORDS : 0), hist_all(want_hist ? SCFQ_HIST_WORDS : 0);
  scfq_partial mine, all;
  scfq_partial_identity(&mine, want_hist ? hist.data() : nullptr);
  const size_t plen = std::strlen(path);
  const bool is_gz = plen >= 3 && std::memcmp(path + plen - 3, ".gz", 3) == 0;      // src/fq_count.nim:31
  int local = SCFQ_OK;          // a rank that fails still takes part in the exchange (with the identity) so nobody hangs
  if (is_gz) {
    // BGZF (bgzip) input shards where its members are: rank r takes the members that start in its byte range, inflates them on its
    // device and scans them; the partials fold as for a plain file.  The ranks first AGREE (one all-gather of a word) that every one
    // of them found its cuts and saw nothing but BGZF members of at most 64 KiB in its range — otherwise, and for every other gzip
    // layout (one deflate stream has no shards), rank 0 inflates and scans all of it and the others contribute the identity.
    static const bool shard_bgzf = env_int("SCFQ_SHARD_BGZF", 1) != 0;
    const int fd = shard_bgzf && world > 1 ? open(path, O_RDONLY) : -1;
    struct stat sb;
    const uint8_t* img = nullptr;
    uint64_t size = 0, b_lo = 0, b_hi = 0;
    int prev = -1;
    uint64_t mine_ok = 0;
    if (fd >= 0 && fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 0 && bgzf_device_enabled() && !std::getenv("SCFQ_NO_BGZF") && scfq_bgzf::probe(fd)) {
      void* m = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
      if (m != MAP_FAILED) {
        img = static_cast<const uint8_t*>(m);
        size = (uint64_t)sb.st_size;
        bool ok = true;
        int prev_hi = -1;
        if (rank > 0) ok = bgzf_cut(img, size, size / (uint64_t)world * (uint64_t)rank, &b_lo, &prev);
        if (rank + 1 < world) ok = bgzf_cut(img, size, size / (uint64_t)world * (uint64_t)(rank + 1), &b_hi, &prev_hi) && ok;
        else b_hi = size;
        if (b_hi < b_lo) b_hi = b_lo;
        ok = ok && (b_lo == b_hi || bgzf_is_pure(img + b_lo, b_hi - b_lo));
        mine_ok = ok ? 1 : 0;
      }
    }
    // An ordinary gzip file (bit 1 of the word the ranks exchange): rank r's members are those that start in [g_lo, g_hi), where a cut is
    // the first demonstrable member start at or after size * r / world (gz_member_boundary; both neighbours find the same one from the
    // bytes alone).  A file of ONE member gives every cut but rank 0's as "none": rank 0 has all of it, as before.
    static const bool shard_gz = env_int("SCFQ_SHARD_GZ", 1) != 0;
    uint64_t g_lo = 0, g_hi = 0;
    int g_first = -1;
    if (!mine_ok && !img && fd >= 0 && shard_gz && gz_device_enabled() && fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size >= 64) {
      void* m = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
      if (m != MAP_FAILED) {
        img = static_cast<const uint8_t*>(m);
        size = (uint64_t)sb.st_size;
        const uint64_t nom_lo = size / (uint64_t)world * (uint64_t)rank, nom_hi = rank + 1 < world ? size / (uint64_t)world * (uint64_t)(rank + 1) : size;
        // (bit 1: an ordinary gzip file as far as this rank can tell without walking it — the cuts themselves are looked for once the ranks
        // have agreed on a scheme; a rank that then finds none says so in the gathered rows)
        int fb = -1;
        if (rank != 0 || gz_member_here(img, size, 0, 1, &fb)) mine_ok = 2;
        // (bit 2: at most ONE member starts inside this rank's share of the file — rank 0: none behind the file's first.  When every rank
        // says so the members are big ones — one, or a few: `cat lane1.gz lane2.gz` — and the ranks cut the deflate streams where BLOCKS
        // start, a member start being a cut of its own: no rank is left without work, as the member scheme leaves the ranks in whose share
        // no member starts.  The search stays inside the rank's share.)
        static const bool shard_blocks = env_int("SCFQ_SHARD_GZ_BLOCKS", 1) != 0;
        if (mine_ok == 2 && shard_blocks && size >= (uint64_t)world * (8ull << 20)) {
          const uint64_t m1 = gz_member_boundary(img, size, rank == 0 ? 1 : nom_lo, size, &fb, nom_hi);
          const uint64_t m2 = m1 < nom_hi ? gz_member_boundary(img, size, m1 + 1, size, &fb, nom_hi) : size;
          if (rank == 0 ? m1 >= nom_hi : m2 >= nom_hi) mine_ok = 6;
        }
      }
    }
    std::vector<uint64_t> oks((size_t)world, 0);
    bool sharded = false, sharded_gz = false, sharded_blk = false;
    if (world > 1 && shard_bgzf) {
      // (every rank takes part in this all-gather whatever it found: a rank that cannot even open the file says 0)
      rc = scfq_comm_allgather_u64(comm, &mine_ok, 1, oks.data(), 0);
      if (rc) { if (img) munmap(const_cast<uint8_t*>(img), (size_t)size); if (fd >= 0) close(fd); std::snprintf(g_err, sizeof g_err, "%s", scfq_comm_error_detail()); return rc; }
      sharded = sharded_gz = sharded_blk = true;
      for (int r = 0; r < world; ++r) {
        sharded = sharded && oks[(size_t)r] == 1;
        sharded_gz = sharded_gz && (oks[(size_t)r] & 2) != 0;
        sharded_blk = sharded_blk && oks[(size_t)r] == 6;
      }
    }
    if (sharded_gz && !sharded_blk) {
      // (the member scheme's cuts: the first demonstrable member start at or after size * r / world and the one after the next rank's)
      int f_hi = -1;
      const uint64_t nom_lo = size / (uint64_t)world * (uint64_t)rank, nom_hi = size / (uint64_t)world * (uint64_t)(rank + 1);
      g_hi = rank + 1 < world ? gz_member_boundary(img, size, nom_hi, nom_hi, &f_hi) : size;
      if (rank == 0) g_lo = 0, (void)gz_member_here(img, size, 0, std::max<uint64_t>(g_hi, 1), &g_first);
      else g_lo = gz_member_boundary(img, size, nom_lo, g_hi, &g_first);
      if (g_hi < g_lo) g_hi = g_lo;
    }
    if (sharded_gz) {
      // every rank inflates and scans the members of its stretch as if they were a file of their own (device path; the host's decoder
      // when that declines), scanned as if they began the input; the partials — with each stretch's first byte and whether its members
      // ended exactly where the next rank's begin — are gathered and folded here, in rank order, with the byte before each stretch
      // (the last byte of the stretch before it) put right first: gz_shard_fix.
      if (o.n_devices >= 1 && hipSetDevice(o.device_ids[0]) != hipSuccess) local = SCFQ_EHIP;
      uint64_t end_off = 0;
      uint64_t blk_crc_raw = 0, blk_len = 0;      // block scheme: this stretch's raw CRC-32 and length
      struct BlkGroup { uint64_t first, last, end_byte; };
      std::vector<BlkGroup> blk_groups;          // ... and the members: the ranks that hold one, the offset just behind its trailer
      if (sharded_blk) {
        // ---- big members: rank r's stretch is [cut_r, cut_{r+1}) — a cut is the member start inside the rank's share of the file when there
        // is one, else the first block start at or behind the share's first byte; a stretch never crosses a member's end ----
        struct Cut { bool found = false, member = false; uint64_t byte = 0, bit = 0; };
        auto cut_of = [&](int r) -> Cut {
          Cut ct;
          if (r <= 0) { ct.found = true; ct.member = true; return ct; }
          if (r >= world) { ct.found = true; ct.member = true; ct.byte = size; ct.bit = size * 8; return ct; }
          const uint64_t lo = size / (uint64_t)world * (uint64_t)r, hi = r + 1 < world ? size / (uint64_t)world * (uint64_t)(r + 1) : size;
          int fb = -1;
          const uint64_t m = gz_member_boundary(img, size, lo, size, &fb, hi);
          if (m < hi) { ct.found = true; ct.member = true; ct.byte = m; ct.bit = m * 8; return ct; }
          const uint64_t bb = gz_block_boundary(img, size, lo, std::min<uint64_t>(16ull << 20, hi - lo));
          if (bb && (bb >> 3) < hi) { ct.found = true; ct.bit = bb; ct.byte = bb >> 3; }
          return ct;
        };
        const Cut c_lo = cut_of(rank), c_hi = cut_of(rank + 1);
        const long h0 = scfq_gzfast::member_header(img, (size_t)size);
        // the stretch as the pipeline sees it: an image that begins at the member's start (a member cut) or at the file's (a block cut:
        // bit positions are the file's), and ends where the next member starts (a member cut) or with the file
        const uint64_t base = c_lo.member ? c_lo.byte : 0;
        const uint64_t image_end = c_hi.member ? c_hi.byte : size;
        GzStretch sx;
        sx.start_bit = c_lo.member ? 0 : c_lo.bit;
        sx.stop_bit = c_hi.member ? 0 : c_hi.bit - 8 * base;
        const bool cuts_ok = h0 > 0 && c_lo.found && c_hi.found && image_end > base + 64 && (c_hi.member || c_hi.bit > (c_lo.member ? c_lo.byte * 8 : c_lo.bit));
        // What every rank learns of every stretch — [proven, cut kinds and positions, bytes, where its member ended, the map] — and what it
        // makes of it: the window in front of its own stretch = the maps of the stretches before it IN THE SAME MEMBER, applied in order to
        // the member's (empty) start.
        const uint32_t kMapWords = (uint32_t)(scfq_gzfast::kWindow / 4), kHead = 8, w1 = kHead + kMapWords;
        std::vector<uint8_t> window(scfq_gzfast::kWindow, 0);
        bool exchanged = false, agree = false;
        int comm_rc = SCFQ_OK;
        auto exchange_fn = [&](GzStretch& x, bool proven) -> int {
          exchanged = true;
          std::vector<uint64_t> row1(w1, 0), rows1((size_t)world * w1, 0);
          // (a stretch that ends at a member cut must have ended WITH its member, exactly at the cut: nothing but members in between)
          proven = proven && cuts_ok && !local && x.map.size() == scfq_gzfast::kWindow && x.member_ended == c_hi.member &&
                   (!c_hi.member || c_hi.byte == size || base + x.end_byte == c_hi.byte);
          row1[0] = proven ? 1 : 0;
          row1[1] = c_lo.member; row1[2] = c_lo.bit; row1[3] = c_hi.member; row1[4] = c_hi.bit;
          row1[5] = x.out_bytes; row1[6] = base + x.end_byte;
          if (proven) std::memcpy(row1.data() + kHead, x.map.data(), 2 * scfq_gzfast::kWindow);
          comm_rc = scfq_comm_allgather_u64(comm, row1.data(), w1, rows1.data(), 0);
          if (comm_rc) return 1;
          agree = true;
          for (int r = 0; r < world; ++r) {
            const uint64_t* rr = rows1.data() + (size_t)r * w1;
            const uint64_t* nx = r + 1 < world ? rows1.data() + (size_t)(r + 1) * w1 : nullptr;
            agree = agree && rr[0] == 1 && (nx ? (rr[3] == nx[1] && rr[4] == nx[2]) : (rr[3] == 1 && rr[4] == size * 8));      // a stretch ends where the next begins
          }
          if (!agree) return 1;
          // the members' ends, for the CRC check at the fold: [first rank, last rank, offset just behind the trailer]
          blk_groups.clear();
          for (int r = 0, a0 = 0; r < world; ++r) {
            const uint64_t* rr = rows1.data() + (size_t)r * w1;
            if (rr[3] == 1) { blk_groups.push_back({(uint64_t)a0, (uint64_t)r, rr[6]}); a0 = r + 1; }
          }
          int first_of_member = rank;
          while (first_of_member > 0 && rows1[(size_t)first_of_member * w1 + 1] == 0) --first_of_member;
          std::vector<uint8_t> next(scfq_gzfast::kWindow, 0);
          uint64_t before_bytes = 0;
          for (int r = first_of_member; r < rank; ++r) {
            const uint16_t* m = reinterpret_cast<const uint16_t*>(rows1.data() + (size_t)r * w1 + kHead);
            for (uint32_t i = 0; i < scfq_gzfast::kWindow; ++i) next[i] = (m[i] & 0x8000u) ? window[m[i] & 0x7FFFu] : (uint8_t)m[i];
            window.swap(next);
            before_bytes += rows1[(size_t)r * w1 + 5];
          }
          x.window = window.data();
          x.valid = (uint32_t)std::min<uint64_t>(scfq_gzfast::kWindow, before_bytes);
          return 0;
        };
        Ctx* c = nullptr;
        SessionLock sl;
        if (!local) local = get_ctx(&c, sl);
        // ONE pass (the default): the stretch's proven symbols are kept — two bytes per inflated byte — while its map goes out and the window
        // comes back (GzStretch::exchange), then they become bytes.  SCFQ_SHARD_GZ_KEEP=0: two passes, the second decoding again (what a
        // device short of memory would want: nothing is kept between them).
        static const bool keep_env = env_int("SCFQ_SHARD_GZ_KEEP", 1) != 0;
        bool keep_on = keep_env;
        if (keep_on && !local) {
          // (the store of kept symbols: two bytes per inflated byte — taken as 12 per compressed byte of the stretch, FASTQ compresses 3 - 5
          // times — must leave the pipeline its own 12 GB or so: a rank whose device is short of that goes over its stretch twice instead;
          // the ranks need not agree on this, the exchange in the middle is the same)
          size_t free_b = 0, total_b = 0;
          const uint64_t lo_b = c_lo.byte, hi_b = c_hi.byte;
          if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = 0; }
          // (SCFQ_TEST_DEVICE_FREE_GB: the tests' way of putting a rank on a device that is short of memory)
          static const int test_free_gb = env_int("SCFQ_TEST_DEVICE_FREE_GB", -1);
          if (test_free_gb >= 0) free_b = std::min<size_t>(free_b, (size_t)test_free_gb << 30);
          if ((hi_b > lo_b ? hi_b - lo_b : 0) * 12 + (16ull << 30) > (uint64_t)free_b) keep_on = false;
        }
        uint64_t out1 = 0;
        if (keep_on) {
          if (!local && cuts_ok) local = begin_session(c, rank == 0);
          if (!local && cuts_ok) {
            sx.exchange = exchange_fn;
            const int r1 = ingest_gz_device(c, img + base, image_end - base, o.flags, timing, nullptr, fd, base, &sx);
            if (r1 == kFallbackToHost) { if (agree) local = SCFQ_EGZ; } else if (r1) local = r1;
          }
          if (!exchanged) (void)exchange_fn(sx, false);
          out1 = sx.out_bytes;
        } else {
          GzStretch sx1 = sx;
          sx1.map_only = true;
          bool proven = false;
          if (!local && cuts_ok) {
            const int r1 = ingest_gz_device(c, img + base, image_end - base, o.flags, false, nullptr, fd, base, &sx1);
            if (r1 == SCFQ_OK) proven = true; else if (r1 != kFallbackToHost) local = r1;
          }
          (void)exchange_fn(sx1, proven);
          out1 = sx1.out_bytes;
          if (agree) {
            sx.window = sx1.window;
            sx.valid = sx1.valid;
            if (!local) local = begin_session(c, rank == 0);
            if (!local) {
              const int r2 = ingest_gz_device(c, img + base, image_end - base, o.flags, timing, nullptr, fd, base, &sx);
              local = r2 == kFallbackToHost ? SCFQ_EGZ : r2;
            }
          }
        }
        if (comm_rc) { if (img) munmap(const_cast<uint8_t*>(img), (size_t)size); if (fd >= 0) close(fd); std::snprintf(g_err, sizeof g_err, "%s", scfq_comm_error_detail()); return local ? local : comm_rc; }
        if (agree) {
          if (!local && sx.out_bytes != out1) local = SCFQ_EGZ;
          if (!local) local = end_session(c, want_hist, &mine, want_hist ? hist.data() : nullptr);
          g_first = sx.first_byte;
          blk_crc_raw = sx.crc_raw;
          blk_len = sx.out_bytes;
        } else {
          local = SCFQ_EGZ;      // (not an error of this rank: the rows below send every rank to the fall-back)
          std::snprintf(g_err, sizeof g_err, "the block cuts of a one-member file did not join up");
        }
      } else if (!local && g_hi > g_lo) {
        Ctx* c = nullptr;
        SessionLock sl;
        local = get_ctx(&c, sl);
        if (!local) local = begin_session(c, rank == 0);
        if (!local) {
          const uint64_t len = g_hi - g_lo;
          static const uint64_t min_bytes = (uint64_t)std::max(0, env_int("SCFQ_GZ_DEVICE_MIN_MB", 4)) << 20;
          local = len >= std::max<uint64_t>(min_bytes, 64) ? ingest_gz_device(c, img + g_lo, len, o.flags, timing, &end_off, fd, g_lo) : kFallbackToHost;
          if (local == kFallbackToHost) {
            // (small stretches, and whatever the device path declines: the host's decoder over the same bytes — Resume from the first
            // block of the stretch's first member, an empty window, no prefix)
            (void)hipStreamSynchronize(c->compute);
            (void)hipStreamSynchronize(c->copy);
            local = begin_session(c, rank == 0);
            const long h = scfq_gzfast::member_header(img + g_lo, (size_t)len);
            if (!local && h <= 0) local = SCFQ_EGZ;
            if (!local) {
              struct RangeSource : Source {
                scfq_gzfast::Resume rs;
                int64_t fill(uint8_t* dst, uint64_t cap) override { const int64_t r = rs.next_chunk(dst, cap); return r < 0 ? (int64_t)SCFQ_EGZ : r; }
              } src;
              const std::vector<uint8_t> no_window(scfq_gzfast::kWindow, 0);
              src.rs.open(img + g_lo, (size_t)len, (uint64_t)h * 8, no_window.data(), 0, 0, 0);
              local = ingest(c, src, -1, o.flags, opt_chunk(&o), timing);
              end_off = src.rs.end_offset();
            }
          }
          // the stretch must be members and nothing else, up to the very byte the next stretch starts at (only the file's last stretch may
          // have the trailing bytes gzread ignores behind it)
          if (!local && end_off != len && g_hi < size) { local = SCFQ_EGZ; std::snprintf(g_err, sizeof g_err, "shard %d: its members end at byte %llu, the next shard begins at %llu", rank, (unsigned long long)(g_lo + end_off), (unsigned long long)g_hi); }
        }
        if (!local) local = end_session(c, want_hist, &mine, want_hist ? hist.data() : nullptr);
      }
      if (img) munmap(const_cast<uint8_t*>(img), (size_t)size);
      if (fd >= 0) close(fd);
      if (local) scfq_partial_identity(&mine, want_hist ? hist.data() : nullptr);
      mine.reserved[0] = (uint64_t)(int64_t)local;
      mine.reserved[1] = (uint64_t)(g_first + 1);          // 0: this stretch holds no byte
      mine.reserved[2] = blk_crc_raw;                      // (block scheme) raw CRC-32 and length of this stretch of the one member
      mine.reserved[3] = blk_len;
      const uint32_t words = SCFQ_PARTIAL_WORDS + (want_hist ? SCFQ_HIST_WORDS : 0);
      std::vector<uint64_t> row(words), rows((size_t)world * words);
      std::memcpy(row.data(), &mine, sizeof mine);
      if (want_hist) std::memcpy(row.data() + SCFQ_PARTIAL_WORDS, hist.data(), SCFQ_HIST_WORDS * sizeof(uint64_t));
      rc = scfq_comm_allgather_u64(comm, row.data(), words, rows.data(), 0);
      if (rc) { std::snprintf(g_err, sizeof g_err, "%s", scfq_comm_error_detail()); return local ? local : rc; }
      bool all_ok = true;
      for (int r = 0; r < world; ++r) all_ok = all_ok && rows[(size_t)r * words + offsetof(scfq_partial, reserved) / 8] == 0;
      if (all_ok && sharded_blk) {
        // every member's CRC-32 and ISIZE against the join of its stretches' (x^(8 |part|), as between the batches of one stretch)
        const int tfd = open(path, O_RDONLY);
        for (const BlkGroup& gpm : blk_groups) {
          uint32_t raw = 0;
          uint64_t len = 0;
          for (uint64_t r = gpm.first; r <= gpm.last; ++r) {
            const uint64_t* rr = rows.data() + (size_t)r * words + offsetof(scfq_partial, reserved) / 8;
            raw = gz_mulmod(gz_xpow8n(rr[3]), raw) ^ (uint32_t)rr[2];
            len += rr[3];
          }
          const uint32_t crc = raw ^ gz_mulmod(gz_xpow8n(len), 0xFFFFFFFFu) ^ 0xFFFFFFFFu;
          uint8_t t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
          const bool have_trailer = tfd >= 0 && gpm.end_byte >= 8 && pread(tfd, t, 8, (off_t)(gpm.end_byte - 8)) == 8;
          const uint32_t t_crc = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
          const uint32_t t_len = (uint32_t)t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
          if (!have_trailer || t_crc != crc || t_len != (uint32_t)(len & 0xFFFFFFFFull)) all_ok = false;      // damaged: gzread's verdict, from rank 0's readers
        }
        if (tfd >= 0) close(tfd);
        if (blk_groups.empty()) all_ok = false;
      }
      if (all_ok) {
        scfq_partial_identity(&all, want_hist ? hist_all.data() : nullptr);
        int before = -1;      // the last byte in front of the stretch being added (-1: nothing yet)
        for (int r = 0; r < world; ++r) {
          scfq_partial pr;
          std::memcpy(&pr, rows.data() + (size_t)r * words, sizeof pr);
          uint64_t* hr = want_hist ? rows.data() + (size_t)r * words + SCFQ_PARTIAL_WORDS : nullptr;
          gz_shard_fix(&pr, hr, before, (int)pr.reserved[1] - 1, o.flags);
          pr.reserved[1] = pr.reserved[2] = pr.reserved[3] = 0;
          if ((rc = scfq_partial_combine(&all, &pr, want_hist ? hist_all.data() : nullptr, hr))) return rc;
          if (pr.bytes) before = (int)(pr.last_byte & 0xFF);
        }
        return scfq_partial_finalize(&all, want_hist ? hist_all.data() : nullptr, out);
      }
      // a cut that was no member start after all (or a damaged file): every rank knows, rank 0 reads the whole file the ordinary way —
      // its readers are gzread byte for byte, error text included — and the others contribute the identity to the exchange below
      local = SCFQ_OK;
      scfq_partial_identity(&mine, want_hist ? hist.data() : nullptr);
      if (rank == 0) {
        scfq_opts o1 = o;
        o1.n_devices = std::min(o.n_devices, 1);
        local = count_file_partial(path, &o1, &mine, want_hist ? hist.data() : nullptr);
      }
    } else if (sharded) {
      if (o.n_devices >= 1 && hipSetDevice(o.device_ids[0]) != hipSuccess) local = SCFQ_EHIP;
      if (!local && b_hi > b_lo) {
        Ctx* c = nullptr;
        SessionLock sl;
        local = get_ctx(&c, sl);
        if (!local) local = begin_session(c, rank == 0);
        if (!local) {
          local = ingest_bgzf_device(c, img + b_lo, b_hi - b_lo, o.flags, opt_chunk(&o), timing, prev, fd, b_lo);
          if (local == kFallbackToHost || local == kNotPureBgzf) {
            // no room for the device path's buffers (kNotPureBgzf cannot happen: the range was walked): the host's block-parallel
            // inflate over the same members
            local = begin_session(c, rank == 0);
            if (!local) { BgzfSource src(fd, b_hi); src.pos = b_lo; local = ingest(c, src, prev, o.flags, opt_chunk(&o), timing); }
          }
        }
        if (!local) local = end_session(c, want_hist, &mine, want_hist ? hist.data() : nullptr);
      }
    } else if (rank == 0) {
      scfq_opts o1 = o;
      o1.n_devices = std::min(o.n_devices, 1);
      local = count_file_partial(path, &o1, &mine, want_hist ? hist.data() : nullptr);
    }
    if (!sharded_gz) {
      if (img) munmap(const_cast<uint8_t*>(img), (size_t)size);
      if (fd >= 0) close(fd);
    }
  } else {
    const int fd = open(path, O_RDONLY);
    struct stat sb;
    if (fd < 0 || fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) {
      if (fd >= 0) close(fd);
      local = SCFQ_EOPEN;
    } else {
      const uint64_t size = (uint64_t)sb.st_size;
      const uint64_t lo = size / (uint64_t)world * (uint64_t)rank + std::min<uint64_t>(size % (uint64_t)world, (uint64_t)rank);
      const uint64_t hi = size / (uint64_t)world * (uint64_t)(rank + 1) + std::min<uint64_t>(size % (uint64_t)world, (uint64_t)rank + 1);
      int prev = -1;
      if (lo) { uint8_t pb; if (pread(fd, &pb, 1, (off_t)(lo - 1)) != 1) local = SCFQ_EIO; else prev = pb; }
      if (!local && o.n_devices >= 1 && hipSetDevice(o.device_ids[0]) != hipSuccess) local = SCFQ_EHIP;
      if (!local) {
        Ctx* c = nullptr;
        SessionLock sl;
        local = get_ctx(&c, sl);
        if (!local) local = begin_session(c, lo == 0);
        if (!local && hi > lo) {
          FdSource src(fd, lo, hi);
          local = ingest(c, src, prev, o.flags, std::min<uint64_t>(opt_chunk(&o), std::max<uint64_t>((hi - lo + 4095) & ~4095ull, 4096)), timing);
        }
        if (!local) local = end_session(c, want_hist, &mine, want_hist ? hist.data() : nullptr);
      }
      close(fd);
    }
  }
  if (local) scfq_partial_identity(&mine, want_hist ? hist.data() : nullptr);
  mine.reserved[0] = (uint64_t)(int64_t)local;       // every rank learns whether any rank failed
  rc = scfq_comm_exchange(comm, &mine, want_hist ? hist.data() : nullptr, &all, want_hist ? hist_all.data() : nullptr, 0);
  if (rc) { std::snprintf(g_err, sizeof g_err, "%s", scfq_comm_error_detail()); return local ? local : rc; }
  if (local) return local;
  if (all.reserved[0]) { std::snprintf(g_err, sizeof g_err, "another rank failed to count its shard"); return SCFQ_EIO; }
  return scfq_partial_finalize(&all, want_hist ? hist_all.data() : nullptr, out);
}

// host only (include/sc_fqcount_debug.h): the two rules of the gzip-member shards, for the CPU tests
int64_t scfq_debug_gz_member_boundary(const char* path, uint64_t from, int* first_byte) {
  if (!path || !first_byte) return SCFQ_EARG;
  const int fd = open(path, O_RDONLY);
  struct stat sb;
  if (fd < 0 || fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode) || sb.st_size == 0) { if (fd >= 0) close(fd); return SCFQ_EOPEN; }
  void* m = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
  close(fd);
  if (m == MAP_FAILED) return SCFQ_EIO;
  const uint64_t n = (uint64_t)sb.st_size;
  const uint64_t at = gz_member_boundary(static_cast<const uint8_t*>(m), n, std::min(from, n), n, first_byte);
  munmap(m, (size_t)n);
  return (int64_t)at;
}
int scfq_debug_gz_shard_fix(scfq_partial* p, uint64_t* hist, int true_prev, int first_byte, uint32_t flags) {
  if (!p) return SCFQ_EARG;
  gz_shard_fix(p, hist, true_prev, first_byte, flags);
  return SCFQ_OK;
}

int scfq_prepare(const scfq_opts* opts) {
  int rc = check_opts(opts);
  if (rc) return rc;
  const scfq_opts o = opts_copy(opts);
  const int nd = std::max(1, o.n_devices);
  int prev = 0;
  HIPCHK(hipGetDevice(&prev));
  for (int d = 0; d < nd && !rc; ++d) {
    if (o.n_devices >= 1 && hipSetDevice(o.device_ids[d]) != hipSuccess) { rc = SCFQ_EHIP; break; }
    Ctx* c = nullptr;
    SessionLock sl;
    rc = get_ctx(&c, sl);
    if (!rc) rc = ensure_staging(c, kDefaultChunk, true);
    if (!rc) rc = want_copy_stream(c);
  }
  (void)hipSetDevice(prev);
  if (!rc && o.n_devices > 1 && !exchange_on_host(o.device_ids, o.n_devices)) {
    const std::vector<scfq_partial> none;
    const std::vector<std::vector<uint64_t>> none_h;
    rc = fold_device_partials(o, o.n_devices, none, none_h, false, nullptr, nullptr);
  }
  return rc;
}

int scfq_shutdown(void) {
  trace("shutdown");
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto& kv : g_ctx)
   for (auto& up : kv.second) {
    Ctx* c = up.get();
    (void)hipSetDevice(c->dev);
    if (c->compute) (void)hipStreamSynchronize(c->compute);
    if (c->copy) (void)hipStreamSynchronize(c->copy);
    if (c->h_pin[0]) (void)hipHostFree(c->h_pin[0]);
    for (int b = 0; b < 2; ++b) {
      if (c->d_stage[b]) (void)hipFree(c->d_stage[b]);
      if (c->ev_copied[b]) (void)hipEventDestroy(c->ev_copied[b]);
      if (c->ev_scanned[b]) (void)hipEventDestroy(c->ev_scanned[b]);
    }
    if (c->d_partials) (void)hipFree(c->d_partials);
    if (c->d_block_partials) (void)hipFree(c->d_block_partials);
    if (c->d_hist_partials) (void)hipFree(c->d_hist_partials);
    if (c->d_range_phase) (void)hipFree(c->d_range_phase);
    if (c->d_guess) (void)hipFree(c->d_guess);
    if (c->d_hist_wg) (void)hipFree(c->d_hist_wg);
    if (c->d_first_ord) (void)hipFree(c->d_first_ord);
    for (int b = 0; b < 2; ++b) {
      if (c->d_comp[b]) (void)hipFree(c->d_comp[b]);
      if (c->d_inf[b]) (void)hipFree(c->d_inf[b]);
      if (c->d_blk[b]) (void)hipFree(c->d_blk[b]);
      if (c->h_blk[b]) (void)hipHostFree(c->h_blk[b]);
      if (c->ev_piece[b]) (void)hipEventDestroy(c->ev_piece[b]);
      if (c->ev_fg[b]) (void)hipEventDestroy(c->ev_fg[b]);
      if (c->fg_stage[b]) (void)hipFree(c->fg_stage[b]);
    }
    note_dev_bytes(-(int64_t)(2 * c->fg_cap));
    if (c->d_dstatus) (void)hipFree(c->d_dstatus);
    if (c->d_comp_first) (void)hipFree(c->d_comp_first);
    if (c->d_inf_first) (void)hipFree(c->d_inf_first);
    // (what scfq_device_bytes_now() counted for this context goes with it)
    note_dev_bytes(-(int64_t)(2 * c->stage_cap + 2 * (c->comp_cap + c->inf_cap) + c->comp_first_cap + c->inf_first_cap));
    { GzShared& gs = gz_shared(c->dev); std::lock_guard<std::mutex> lk(gs.mu); for (int e = 0; e < GzShared::kMax; ++e) gz_free(&gs.buf[e]); }
    if (c->d_state) (void)hipFree(c->d_state);
    if (c->h_state) (void)hipHostFree(c->h_state);
    for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->cp_pool) (void)hipEventDestroy(e);
    if (c->ev_caller) (void)hipEventDestroy(c->ev_caller);
    if (c->compute) (void)hipStreamDestroy(c->compute);
    if (c->copy && !c->copy_is_alias) (void)hipStreamDestroy(c->copy);
  }
  scfq_dedup_release_pools();     // fq-dedup keeps its scratch in library-owned stream-ordered pools: give them back
  release_comms();                // communicators of the single-process multi-device path
  g_ctx.clear();
  trace("shutdown done");
  return SCFQ_OK;
}

// ---- whole (inflated) input into HBM: the staging step of the commands that need random access to records ------------
// Same source selection as scfq_count_file (plain pread / BGZF block-parallel inflate / serial gzread), same pinned
// double buffer and copy stream; the device buffer grows geometrically when the inflated size is not known up front.
int scfq_stage_file(const char* path, const scfq_opts* opts, void** dptr_out, uint64_t* n_out) {
  if (!path || !dptr_out || !n_out) return SCFQ_EARG;
  int rc = check_opts(opts);
  if (rc) return rc;
  *dptr_out = nullptr;
  *n_out = 0;
  const size_t plen = std::strlen(path);
  const bool is_gz = plen >= 3 && std::memcmp(path + plen - 3, ".gz", 3) == 0;     // src/fq_dedup.nim:32, src/fq_count.nim:31
  std::unique_ptr<Source> src;
  int fd = -1;
  gzFile gz = nullptr;
  uint64_t hint = 64ull << 20;
  struct Closer { int* fd; gzFile* gz; std::unique_ptr<Source>* s; ~Closer() { s->reset(); if (*gz) gzclose(*gz); if (*fd >= 0) close(*fd); } } closer{&fd, &gz, &src};
  struct stat sb;
  if (is_gz && bgzf_device_enabled()) {
    // pure BGZF: the inflated size is the sum of the ISIZE fields, and the members are inflated on the device straight
    // into the result buffer
    const int bfd = open(path, O_RDONLY);
    struct stat bsb;
    if (bfd >= 0 && fstat(bfd, &bsb) == 0 && S_ISREG(bsb.st_mode) && bsb.st_size > 0 && !std::getenv("SCFQ_NO_BGZF") && scfq_bgzf::probe(bfd)) {
      void* m = mmap(nullptr, (size_t)bsb.st_size, PROT_READ, MAP_PRIVATE, bfd, 0);
      if (m != MAP_FAILED) {
        struct Unmap { void* m; size_t n; int fd; ~Unmap() { munmap(m, n); close(fd); } } um{m, (size_t)bsb.st_size, bfd};
        const uint8_t* img = static_cast<const uint8_t*>(m);
        const uint64_t fsize = (uint64_t)bsb.st_size;
        if (bgzf_is_pure(img, fsize)) {
          uint64_t total = 0;
          for (uint64_t q = 0; q < fsize;) { uint32_t hl; const uint32_t bs = scfq_bgzf::block_size(img + q, fsize - q, &hl); total += scfq_bgzf::rd32(img + q + bs - 4); q += bs; }
          if (opts && opts->n_devices >= 1) HIPCHK(hipSetDevice(opts->device_ids[0]));
          Ctx* c = nullptr;
          SessionLock sl;
          rc = get_ctx(&c, sl);
          if (rc) return rc;
          if ((rc = want_copy_stream(c))) return rc;      // (staging a whole file: chunk k + 1 crosses PCIe under chunk k's work)
          rc = ensure_bgzf_device_buffers(c, fsize);
          if (rc == SCFQ_OK) {
            uint8_t* d_buf = nullptr;
            HIPCHK(hipMalloc(&d_buf, std::max<uint64_t>(total, 16)));
            struct BufGuard { uint8_t** p; ~BufGuard() { if (*p) (void)hipFree(*p); } } bg{&d_buf};
            HIPCHK(hipMemsetAsync(c->d_dstatus, 0, sizeof(uint32_t), c->compute));
            uint64_t pos = 0, off = 0;
            for (unsigned it = 0; pos < fsize; ++it) {
              const int b = it & 1;
              if (it >= 2) HIPCHK(hipEventSynchronize(c->ev_copied[b]));
              uint32_t nb = 0;
              uint64_t ob = 0;
              const int64_t used = bgzf_plan(img, fsize, pos, c->inf_cap, c->comp_cap, bgzf_members_per_launch(c->inf_cap), c->h_blk[b], &nb, &ob);
              if (used < 0) return SCFQ_EGZ;
              if (used == 0) break;
              if (it >= 2) HIPCHK(hipStreamWaitEvent(c->copy, c->ev_scanned[b], 0));
              { FileBytes fb; fb.img = img; fb.fd = bfd; if ((rc = copy_through_ring(c, c->d_comp[b], fb, pos, (uint64_t)used))) return rc; }
              HIPCHK(hipMemcpyAsync(c->d_blk[b], c->h_blk[b], nb * sizeof(scfq_dinflate::Block), hipMemcpyHostToDevice, c->copy));
              HIPCHK(hipEventRecord(c->ev_copied[b], c->copy));
              HIPCHK(hipStreamWaitEvent(c->compute, c->ev_copied[b], 0));
              if (nb) {
                hipLaunchKernelGGL(scfq_dinflate::bgzf_inflate, dim3((nb + scfq_dinflate::kWavesPerWg - 1) / scfq_dinflate::kWavesPerWg),
                                   dim3(64 * scfq_dinflate::kWavesPerWg), scfq_dinflate::kWavesPerWg * scfq_dinflate::kWaveLdsBytes,
                                   c->compute, c->d_comp[b], c->d_blk[b], nb, d_buf + off, c->d_dstatus, inflate_serial_loop());
                HIPCHK(hipGetLastError());
                if ((rc = bgzf_crc_ready(c))) return rc;
                hipLaunchKernelGGL(scfq_dinflate::bgzf_crc32_members, dim3(nb), dim3(256), 0, c->compute, d_buf + off, c->d_blk[b], nb, c->d_dstatus);
                HIPCHK(hipGetLastError());
              }
              HIPCHK(hipEventRecord(c->ev_scanned[b], c->compute));
              pos += (uint64_t)used;
              off += ob;
            }
            uint32_t st = 0;
            HIPCHK(hipMemcpyAsync(c->h_state + kStateWords - 1, c->d_dstatus, sizeof(uint32_t), hipMemcpyDeviceToHost, c->compute));
            HIPCHK(hipStreamSynchronize(c->compute));
            std::memcpy(&st, c->h_state + kStateWords - 1, sizeof st);
            if (st || off != total) { std::snprintf(g_err, sizeof g_err, "device inflate: error mask 0x%x", st); return SCFQ_EGZ; }
            *dptr_out = d_buf;
            *n_out = total;
            d_buf = nullptr;
            return SCFQ_OK;
          }
          if (rc != kFallbackToHost) return rc;
          rc = SCFQ_OK;
        }
      } else {
        close(bfd);
      }
    } else if (bfd >= 0) {
      close(bfd);
    }
  }
  if (is_gz) {
    fd = open(path, O_RDONLY);
    if (fd >= 0 && fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode)) hint = std::max<uint64_t>(hint, 4 * (uint64_t)sb.st_size);
    if (fd >= 0 && fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && !std::getenv("SCFQ_NO_BGZF") && scfq_bgzf::probe(fd)) {
      src.reset(new BgzfSource(fd, (uint64_t)sb.st_size));
    } else {
      if (fd >= 0) close(fd);
      fd = -1;
      src = open_gz_source(path, opt_chunk(opts), &gz);
      if (!src) return SCFQ_EOPEN;
    }
  } else {
    fd = open(path, O_RDONLY);
    if (fd < 0 || fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) return SCFQ_EOPEN;
    hint = std::max<uint64_t>((uint64_t)sb.st_size, 4096);
    src.reset(new FdSource(fd, 0, (uint64_t)sb.st_size));
  }
  if (opts && opts->n_devices >= 1) HIPCHK(hipSetDevice(opts->device_ids[0]));
  Ctx* c = nullptr;
  SessionLock sl;
  rc = get_ctx(&c, sl);
  if (rc) return rc;
  if ((rc = want_copy_stream(c))) return rc;      // (staging a whole file: chunk k + 1 crosses PCIe under chunk k's work)
  const uint64_t chunk = opt_chunk(opts);
  rc = ensure_staging(c, chunk, true);
  if (rc) return rc;
  uint8_t* d_buf = nullptr;
  uint64_t cap = hint, off = 0;
  HIPCHK(hipMalloc(&d_buf, cap));
  struct BufGuard { uint8_t** p; ~BufGuard() { if (*p) (void)hipFree(*p); } } bg{&d_buf};
  for (unsigned it = 0;; ++it) {
    const int b = it & 1;
    if (it >= 2) HIPCHK(hipEventSynchronize(c->ev_copied[b]));
    const int64_t got = src->fill(c->h_pin[b], chunk);
    if (got < 0) return (int)got;
    if (got == 0) break;
    if (off + (uint64_t)got > cap) {
      HIPCHK(hipStreamSynchronize(c->copy));
      const uint64_t ncap = std::max<uint64_t>(2 * cap, off + (uint64_t)got);
      uint8_t* nb = nullptr;
      HIPCHK(hipMalloc(&nb, ncap));
      hipError_t e = hipMemcpy(nb, d_buf, off, hipMemcpyDeviceToDevice);
      if (e != hipSuccess) { (void)hipFree(nb); HIPCHK(e); }
      (void)hipFree(d_buf);
      d_buf = nb;
      cap = ncap;
    }
    HIPCHK(hipMemcpyAsync(d_buf + off, c->h_pin[b], (size_t)got, hipMemcpyHostToDevice, c->copy));
    HIPCHK(hipEventRecord(c->ev_copied[b], c->copy));
    off += (uint64_t)got;
  }
  HIPCHK(hipStreamSynchronize(c->copy));
  *dptr_out = d_buf;
  *n_out = off;
  d_buf = nullptr;      // ownership passes to the caller (scfq_device_free)
  return SCFQ_OK;
}

int scfq_device_free(void* dptr) {
  if (dptr) HIPCHK(hipFree(dptr));
  return SCFQ_OK;
}

// ---- device-side BGZF inflate ---------------------------------------------------------------------------------------
// Walks the members of a BGZF image on the host (header fields only): fills the block table for the members whose
// inflated bytes fit into out_cap, starting at byte `pos`.  Returns the number of bytes consumed, 0 at the end of the
// image (or at a clean end-of-file marker run), -1 when the bytes at pos are not a BGZF member (caller falls back to the
// host path) and -2 for a truncated member.
}  // extern "C"
namespace {
int64_t bgzf_plan(const uint8_t* img, uint64_t n, uint64_t pos, uint64_t out_cap, uint64_t comp_cap, uint32_t max_blocks,
                  scfq_dinflate::Block* blocks, uint32_t* n_blocks, uint64_t* out_bytes) {
  uint64_t p = pos, out = 0;
  uint32_t nb = 0;
  while (p < n && nb < max_blocks) {
    uint32_t hl = 0;
    const uint32_t bs = scfq_bgzf::block_size(img + p, n - p, &hl);
    if (!bs) { if (p == pos) return -1; break; }        // something else follows: this chunk ends here, the next call reports it
    if (p + bs > n) return -2;
    const uint32_t isize = scfq_bgzf::rd32(img + p + bs - 4);
    if (isize > (1u << 16)) { if (p == pos) return -1; break; }
    if (out + isize > out_cap || (p + bs - pos) > comp_cap) break;
    scfq_dinflate::Block b;
    b.in_off = (uint32_t)(p - pos + hl);
    b.in_len = bs - hl - 8;
    b.out_off = (uint32_t)out;
    b.isize = isize;
    b.crc = scfq_bgzf::rd32(img + p + bs - 8);
    blocks[nb++] = b;
    out += isize;
    p += bs;
  }
  *n_blocks = nb;
  *out_bytes = out;
  return (int64_t)(p - pos);
}
}  // namespace
extern "C" {

// Diagnostic / test entry: inflate a whole BGZF image (host memory) on the device, result to host memory.
// Returns the inflated size, SCFQ_EARG when the image is not pure BGZF or does not fit, SCFQ_EGZ for a corrupt member.
int64_t scfq_debug_bgzf_inflate(const void* image, uint64_t n, void* out, uint64_t cap) {
  if ((!image && n) || (!out && cap)) return SCFQ_EARG;
  Ctx* c = nullptr;
  SessionLock sl;
  int rc = get_ctx(&c, sl);
  if (rc) return rc;
  const uint8_t* img = static_cast<const uint8_t*>(image);
  std::vector<scfq_dinflate::Block> blocks(1u << 20);
  uint64_t total = 0, pos = 0;
  uint8_t *d_comp = nullptr, *d_out = nullptr;
  scfq_dinflate::Block* d_blocks = nullptr;
  uint32_t* d_status = nullptr;
  const uint64_t kChunk = 64ull << 20;
  HIPCHK(hipMalloc(&d_comp, kChunk + 64));
  HIPCHK(hipMalloc(&d_out, kChunk));
  HIPCHK(hipMalloc(&d_blocks, (kChunk / 64 + 16) * sizeof(scfq_dinflate::Block)));
  HIPCHK(hipMalloc(&d_status, 64));
  struct Free { uint8_t* a; uint8_t* b; void* c; void* d; ~Free() { (void)hipFree(a); (void)hipFree(b); (void)hipFree(c); (void)hipFree(d); } } fr{d_comp, d_out, d_blocks, d_status};
  HIPCHK(hipMemsetAsync(d_status, 0, 64, c->compute));
  while (pos < n) {
    uint64_t ob = 0;
    uint32_t nb = 0;
    const int64_t used = bgzf_plan(img, n, pos, kChunk, kChunk, (uint32_t)(kChunk / 64), blocks.data(), &nb, &ob);
    if (used == -2) return SCFQ_EGZ;
    if (used < 0) return SCFQ_EARG;
    if (used == 0) break;
    if (total + ob > cap) return SCFQ_EARG;
    HIPCHK(hipMemcpyAsync(d_comp, img + pos, (size_t)used, hipMemcpyHostToDevice, c->compute));
    HIPCHK(hipMemcpyAsync(d_blocks, blocks.data(), nb * sizeof(scfq_dinflate::Block), hipMemcpyHostToDevice, c->compute));
    hipLaunchKernelGGL(scfq_dinflate::bgzf_inflate, dim3((nb + scfq_dinflate::kWavesPerWg - 1) / scfq_dinflate::kWavesPerWg),
                       dim3(64 * scfq_dinflate::kWavesPerWg), scfq_dinflate::kWavesPerWg * scfq_dinflate::kWaveLdsBytes, c->compute,
                       d_comp, d_blocks, nb, d_out, d_status, inflate_serial_loop());
    HIPCHK(hipGetLastError());
    { const int r_ = bgzf_crc_ready(c); if (r_) return r_; }
    hipLaunchKernelGGL(scfq_dinflate::bgzf_crc32_members, dim3(nb), dim3(256), 0, c->compute, d_out, d_blocks, nb, d_status);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(static_cast<uint8_t*>(out) + total, d_out, (size_t)ob, hipMemcpyDeviceToHost, c->compute));
    HIPCHK(hipStreamSynchronize(c->compute));
    total += ob;
    pos += (uint64_t)used;
  }
  uint32_t st = 0;
  HIPCHK(hipMemcpy(&st, d_status, 4, hipMemcpyDeviceToHost));
#ifdef SCFQ_LPROF
  { unsigned long long w[24]; HIPCHK(hipMemcpyFromSymbol(w, HIP_SYMBOL(scfq_dinflate::g_lprof), sizeof w));
    if (w[14]) {
      const double g = (double)w[14], ra = (double)std::max<unsigned long long>(1, w[11]);
      std::fprintf(stderr, "dprof: groups %llu symbols/group %.1f rounds/group %.2f chunks/group %.2f sub-groups+alone/group %.2f | cycles per round: window+lengths %.0f walk %.0f collect %.0f | per group: part A %.0f decode %.0f emit %.0f\n",
                   w[14], w[17] / g, w[11] / g, w[15] / g, w[16] / g, w[8] / ra, w[9] / ra, w[10] / ra, (w[8] + w[9] + w[10]) / g, w[12] / g, w[13] / g);
    }
    const double r = (double)std::max<unsigned long long>(1, w[5]);
    std::fprintf(stderr, "lprof: rounds %llu symbols/round %.2f slow rounds %.3f | cycles per round: window wait %.0f decode %.0f walk %.0f pending store (load wait) %.0f emit %.0f\n", w[5],
                 w[6] / r, w[7] / r, w[0] / r, w[1] / r, w[2] / r, w[3] / r, w[4] / r);
    unsigned long long z[24] = {0}; HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(scfq_dinflate::g_lprof), z, sizeof z)); }
#endif
#ifdef SCFQ_DSTATS
  { uint32_t w[8]; HIPCHK(hipMemcpy(w, d_status, 32, hipMemcpyDeviceToHost));
    std::fprintf(stderr, "dstats: deflate blocks %u literals %u matches %u match bytes %llu overlapping %u longer than 64: %u\n", w[1], w[2], w[3], (unsigned long long)w[4] << 4, w[5], w[6]); }
#endif
  if (st) { std::snprintf(g_err, sizeof g_err, "device inflate: error mask 0x%x (2 = corrupt deflate data, 4 = length, 8 = CRC-32)", st); return SCFQ_EGZ; }
  return (int64_t)total;
}

// ---- K5: line index of a device-resident input ------------------------------------------------------------------
// One pass over the input (fq_index_pos keeps the newline positions of every tile, fq_index_expand_pos turns them into offsets; the
// mask form — fq_index_masks keeps one bit per byte, fq_index_expand walks the bits — for inputs with 128+ newlines in a 4 KiB tile);
// SCFQ_INDEX_TWO_PASS=1 keeps the first form (K1 + K2 count, prefix kernel, second pass over the input) for comparison.
static int index_lines_two_pass(Ctx* c, const uint8_t* base, uint64_t n, uint64_t* d_line_off, uint64_t cap, uint64_t* lines_out);

// internal (fq-dedup): csrc/scfq_index_aux.hpp

int scfq_index_lines(const void* dptr, uint64_t n, uint64_t* d_line_off, uint64_t cap, uint64_t* lines_out) {
  return scfq_index_lines_ex(dptr, n, d_line_off, cap, lines_out, nullptr);
}

int scfq_index_lines_ex(const void* dptr, uint64_t n, uint64_t* d_line_off, uint64_t cap, uint64_t* lines_out, uint32_t* flags_out) {
  return scfq_index_lines_ex2(dptr, n, d_line_off, cap, lines_out, flags_out, nullptr);
}

int scfq_index_lines_ex2(const void* dptr, uint64_t n, uint64_t* d_line_off, uint64_t cap, uint64_t* lines_out, uint32_t* flags_out, scfq_index_aux* aux) {
  if ((!dptr && n) || !lines_out) return SCFQ_EARG;
  if (aux) { aux->filled = 0; aux->unk_complete = 0; aux->n_tiles = 0; }
  if (aux && (!aux->keys || !aux->idx || !aux->hdr || (aux->key_bytes != 4 && aux->key_bytes != 8) || aux->hash_bits > 56 || !d_line_off)) return SCFQ_EARG;
  if (flags_out) *flags_out = 1u;            // unknown until the one-pass kernel says otherwise
  Ctx* c = nullptr;
  SessionLock sl;
  int rc = get_ctx(&c, sl);
  if (rc) return rc;
  rc = begin_session(c, true);
  if (rc) return rc;
  if ((rc = wait_for_caller(c, nullptr))) return rc;      // input and line_off are the caller's device buffers
  const uint8_t* base = static_cast<const uint8_t*>(dptr);
  static const bool two_pass = env_int("SCFQ_INDEX_TWO_PASS", 0) != 0;
  if (two_pass) return index_lines_two_pass(c, base, n, d_line_off, cap, lines_out);
  *lines_out = 0;
  if (n == 0) {
    if (flags_out) *flags_out = 0;
    if (d_line_off && cap >= 1) { HIPCHK(hipMemsetAsync(d_line_off, 0, sizeof(uint64_t), c->compute)); HIPCHK(hipStreamSynchronize(c->compute)); }
    return SCFQ_OK;
  }
  const uint64_t B = (uint64_t)(uintptr_t)base, A0 = B & ~(uint64_t)(scfq::kTile - 1);
  const uint64_t n_tiles = (B + n - A0 + scfq::kTile - 1) / scfq::kTile;
  if (n_tiles >= (1ull << 32)) { std::snprintf(g_err, sizeof g_err, "a single index launch covers at most 16 TiB"); return SCFQ_EARG; }
  const uint32_t tpr = pick_tiles_per_range(c, n_tiles);
  const uint64_t n_ranges = (n_tiles + tpr - 1) / tpr;
  const bool write = d_line_off && cap >= 1;
  // The compact form first (fq_index_pos: 16-bit newline positions per tile, 256 B of scratch per tile), the mask form (a bit per
  // byte, 512 B per tile) when a tile holds more newlines than its slot — lines shorter than 33 bytes on average — or
  // SCFQ_INDEX_COMPACT=0 says so.  scratch: [n_tiles * (32 | 64)] positions or masks | [n_ranges] counts | [n_ranges + 1] first ordinals | flags
  static const bool compact_on = env_int("SCFQ_INDEX_COMPACT", 1) != 0;
  uint64_t* d_ord = nullptr;
  uint32_t* d_flags = nullptr;
  bool have_state = false;      // the compact form has brought the line count, the last byte and the flags back already
  for (int form = compact_on ? 0 : 1; form < 2; ++form) {
    const bool with_hash = form == 0 && aux && write;
    const uint64_t per_tile = form == 0 ? scfq::kPosCap / 4 + (with_hash ? scfq::kPosHashCap : 0) : 64;
    const uint64_t words = n_tiles * per_tile + 2 * n_ranges + 8;
    if (words > c->cap_first_ord) {
      HIPCHK(hipStreamSynchronize(c->compute));
      if (c->d_first_ord) HIPCHK(hipFree(c->d_first_ord));
      c->d_first_ord = nullptr;
      c->cap_first_ord = 0;
      const uint64_t want = words + words / 8;
      HIPCHK(hipMalloc(&c->d_first_ord, want * sizeof(uint64_t)));
      c->cap_first_ord = want;
    }
    uint64_t* d_tiles = c->d_first_ord;
    uint64_t* d_counts = d_tiles + n_tiles * per_tile;
    d_ord = d_counts + n_ranges;
    d_flags = reinterpret_cast<uint32_t*>(d_ord + n_ranges + 1);
    HIPCHK(hipMemsetAsync(d_flags, 0, 8, c->compute));
    if (write) HIPCHK(hipMemsetAsync(d_line_off, 0, sizeof(uint64_t), c->compute));          // line 0 starts at offset 0
    const unsigned grid = (unsigned)((n_ranges + scfq::kWavesPerBlock - 1) / scfq::kWavesPerBlock);
    if (form == 0) {
      scfq::IndexPosArgs pa{};
      pa.hash_at = with_hash ? d_tiles + n_tiles * (scfq::kPosCap / 4) : nullptr;      // (behind the position slots)
      pa.hash_seed = aux ? aux->seed : 0;
      pa.base = base;
      pa.n = n;
      pa.tiles_per_range = tpr;
      pa.n_ranges = n_ranges;
      pa.pos = reinterpret_cast<uint16_t*>(d_tiles);
      pa.counts = d_counts;
      pa.flags = d_flags;
      pa.want_cr = flags_out ? 1u : 0u;
      hipLaunchKernelGGL(scfq::fq_index_pos, dim3(grid), dim3(64 * scfq::kWavesPerBlock), scfq::kIndexPosLds, c->compute, pa);
    } else {
      scfq::IndexMaskArgs ma;
      ma.base = base;
      ma.n = n;
      ma.tiles_per_range = tpr;
      ma.n_ranges = n_ranges;
      ma.masks = d_tiles;
      ma.counts = d_counts;
      ma.flags_out = flags_out ? d_flags : nullptr;
      hipLaunchKernelGGL(scfq::fq_index_masks, dim3(grid), dim3(64 * scfq::kWavesPerBlock), scfq::kWavesPerBlock * 2 * scfq::kTile, c->compute, ma);
    }
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(scfq::fq_nl_prefix, dim3(1), dim3(1024), 0, c->compute, d_counts, n_ranges, (uint64_t)0, d_ord, 1u);
    HIPCHK(hipGetLastError());
    if (write && form == 0) {
      scfq::IndexExpandPosArgs ea{};
      if (with_hash) {
        ea.hash_at = d_tiles + n_tiles * (scfq::kPosCap / 4);
        ea.keys = aux->keys; ea.idx = aux->idx; ea.hdr = aux->hdr;
        ea.cap_records = aux->cap_records; ea.key_bytes = aux->key_bytes; ea.hash_bits = aux->hash_bits; ea.hash_seed = aux->seed;
        ea.unk = (aux->unk && aux->unk_tiles >= n_tiles) ? aux->unk : nullptr;
        ea.flags_rw = d_flags;
      }
      ea.pos = reinterpret_cast<const uint16_t*>(d_tiles);
      ea.flags = d_flags;
      ea.lead = B - A0;
      ea.n_tiles = (uint32_t)n_tiles;
      ea.tiles_per_range = tpr;
      ea.n_ranges = n_ranges;
      ea.first_ord = d_ord;
      ea.line_off = d_line_off;
      ea.cap = cap;
      ea.off_base = 0;
      hipLaunchKernelGGL(scfq::fq_index_expand_pos, dim3((unsigned)((n_ranges + 3) / 4)), dim3(256), 0, c->compute, ea);
      HIPCHK(hipGetLastError());
    } else if (write) {
      scfq::IndexExpandArgs ea;
      ea.masks = d_tiles;
      ea.lead = B - A0;
      ea.n_tiles = (uint32_t)n_tiles;
      ea.tiles_per_range = tpr;
      ea.n_ranges = n_ranges;
      ea.first_ord = d_ord;
      ea.line_off = d_line_off;
      ea.cap = cap;
      ea.off_base = 0;
      hipLaunchKernelGGL(scfq::fq_index_expand, dim3((unsigned)((n_ranges + 3) / 4)), dim3(256), 0, c->compute, ea);
      HIPCHK(hipGetLastError());
    }
    if (form == 1) break;
    // (the compact form's one question to the device: did every tile fit its slot?  Answered together with the line count below when
    // it did; a file of very short lines pays this wait and a second pass)
    HIPCHK(hipMemcpyAsync(c->h_state + 2, d_flags, 4, hipMemcpyDeviceToHost, c->compute));
    HIPCHK(hipMemcpyAsync(c->h_state, d_ord + n_ranges, sizeof(uint64_t), hipMemcpyDeviceToHost, c->compute));
    HIPCHK(hipMemcpyAsync(c->h_state + 1, base + n - 1, 1, hipMemcpyDeviceToHost, c->compute));
    HIPCHK(hipStreamSynchronize(c->compute));
    if (!(c->h_state[2] & 2u)) {
      have_state = true;
      if (with_hash) { aux->filled = 1; aux->n_tiles = n_tiles; aux->unk_complete = (aux->unk && aux->unk_tiles >= n_tiles && !(c->h_state[2] & 4u)) ? 1 : 0; }
      break;
    }
    trace("line index: a tile with more newlines than the compact form's slot holds, the mask form runs");
  }
  // first_ord[n_ranges] = 1 + number of '\n'
  if (!have_state) {
    HIPCHK(hipMemcpyAsync(c->h_state, d_ord + n_ranges, sizeof(uint64_t), hipMemcpyDeviceToHost, c->compute));
    HIPCHK(hipMemcpyAsync(c->h_state + 1, base + n - 1, 1, hipMemcpyDeviceToHost, c->compute));
    HIPCHK(hipMemcpyAsync(c->h_state + 2, d_flags, 4, hipMemcpyDeviceToHost, c->compute));
    HIPCHK(hipStreamSynchronize(c->compute));
  }
  if (flags_out) *flags_out = (uint32_t)(c->h_state[2] & 1u);
  c->h_state[0] -= 1;
  const uint64_t nl = c->h_state[0];
  const bool open_end = (uint8_t)(c->h_state[1] & 0xFF) != (uint8_t)'\n';
  const uint64_t lines = nl + (open_end ? 1u : 0u);
  *lines_out = lines;
  if (write && cap >= lines + 1 && open_end) {
    // the final line has no '\n': the sentinel pretends there is one right after the input
    c->h_state[0] = n + 1;
    HIPCHK(hipMemcpyAsync(d_line_off + lines, c->h_state, sizeof(uint64_t), hipMemcpyHostToDevice, c->compute));
    HIPCHK(hipStreamSynchronize(c->compute));
  }
  return SCFQ_OK;
}

static int index_lines_two_pass(Ctx* c, const uint8_t* base, uint64_t n, uint64_t* d_line_off, uint64_t cap, uint64_t* lines_out) {
  int rc = scan_async(c, base, n, -1, 0, false);
  if (rc) return rc;
  scfq_partial p;
  rc = end_session(c, false, &p, nullptr);      // synchronises: the newline count sizes the index
  if (rc) return rc;
  const uint64_t lines = p.nl + ((n > 0 && p.last_byte != (uint64_t)'\n') ? 1u : 0u);
  *lines_out = lines;
  if (!d_line_off || cap < lines + 1) return SCFQ_OK;   // count only / index does not fit: caller sizes and calls again
  HIPCHK(hipMemsetAsync(d_line_off, 0, sizeof(uint64_t), c->compute));          // line 0 starts at offset 0
  if (n) {
    const uint64_t n_ranges = c->last_ranges;
    if (n_ranges + 1 > c->cap_first_ord) {
      if (c->d_first_ord) HIPCHK(hipFree(c->d_first_ord));
      c->d_first_ord = nullptr;
      c->cap_first_ord = 0;
      const uint64_t want = std::max<uint64_t>(n_ranges + 1 + n_ranges / 4, 4096);
      HIPCHK(hipMalloc(&c->d_first_ord, want * sizeof(uint64_t)));
      c->cap_first_ord = want;
    }
    hipLaunchKernelGGL(scfq::fq_nl_prefix, dim3(1), dim3(1024), 0, c->compute, c->d_partials, n_ranges, (uint64_t)0, c->d_first_ord);
    HIPCHK(hipGetLastError());
    scfq::IndexArgs ia;
    ia.base = base;
    ia.n = n;
    ia.tiles_per_range = c->last_tpr;
    ia.n_ranges = n_ranges;
    ia.first_ord = c->d_first_ord;
    ia.line_off = d_line_off;
    ia.off_base = 0;
    const unsigned grid = (unsigned)((n_ranges + scfq::kWavesPerBlock - 1) / scfq::kWavesPerBlock);
    hipLaunchKernelGGL(scfq::fq_index_lines, dim3(grid), dim3(64 * scfq::kWavesPerBlock), scfq::kWavesPerBlock * 2 * scfq::kTile,
                       c->compute, ia);
    HIPCHK(hipGetLastError());
    if (p.last_byte != (uint64_t)'\n') {
      // the final line has no '\n': the sentinel pretends there is one right after the input
      c->h_state[0] = n + 1;
      HIPCHK(hipMemcpyAsync(d_line_off + lines, c->h_state, sizeof(uint64_t), hipMemcpyHostToDevice, c->compute));
    }
  }
  HIPCHK(hipStreamSynchronize(c->compute));
  return SCFQ_OK;
}

// ---- diagnostic: run the host-side source selection (plain pread / BGZF parallel inflate / serial gzread) of
// scfq_count_file without any device, writing the byte stream the scan would see into dst. Returns the byte
// count, or a negative SCFQ_* code; SCFQ_EARG when the stream is longer than cap.
// Test hook (host only, no device): the inflated bytes of a gzip file whose FIRST member is read in two halves — the serial decoder up to the
// first block boundary at or after `after_bytes` of output, then scfq_gzfast::Resume from that exact bit with the window, CRC-32 and
// length of the first half: the hand-over the device gzip path makes when a batch has no room (scfq_gzdev.hpp), without a device.
// Returns the number of bytes written to dst (the whole stream as gzread yields it), SCFQ_EGZ on a corrupt stream.
int64_t scfq_debug_gz_resume(const char* path, uint64_t after_bytes, void* dst, uint64_t cap, uint64_t chunk) {
  if (!path || (!dst && cap)) return SCFQ_EARG;
  if (chunk < (1u << 16)) chunk = 1u << 20;
  const int fd = open(path, O_RDONLY);
  struct stat sb;
  if (fd < 0 || fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode) || sb.st_size < 18) { if (fd >= 0) close(fd); return SCFQ_EOPEN; }
  const size_t n = (size_t)sb.st_size;
  void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
  close(fd);
  if (m == MAP_FAILED) return SCFQ_EIO;
  struct Unmap { void* m; size_t n; ~Unmap() { munmap(m, n); } } um{m, n};
  const uint8_t* img = static_cast<const uint8_t*>(m);
  const long h0 = scfq_gzfast::member_header(img, n);
  if (h0 <= 0) return SCFQ_EGZ;
  // first half: the serial decoder over the first member, logging every block header (bit, bytes before it)
  uint8_t* out = static_cast<uint8_t*>(dst);
  std::vector<uint64_t> log(2 * 65536);
  auto dec = std::unique_ptr<scfq_inflate::Decoder>(new scfq_inflate::Decoder());
  dec->begin(img, img + n);
  dec->in_next = img + h0;
  dec->boundary_log = log.data();
  dec->boundary_cap = log.size() / 2;
  std::vector<uint8_t> tmp(cap + scfq_inflate::kOutSlack + 64);
  uint8_t* o = tmp.data();
  const int r = dec->run(o, tmp.data() + cap);
  if (r < 0 && r != scfq_inflate::kErrTruncated) { /* a damaged first member may still have boundaries in front of the damage */ }
  uint64_t bit = 0, before = 0;
  bool have = false;
  for (size_t k = 0; k < dec->boundary_n; ++k)
    if (log[2 * k + 1] >= after_bytes || k + 1 == dec->boundary_n) { bit = log[2 * k]; before = log[2 * k + 1]; have = true; break; }
  if (!have || before > cap) return SCFQ_EGZ;
  std::memcpy(out, tmp.data(), (size_t)before);
  std::vector<uint8_t> window(scfq_gzfast::kWindow, 0);
  const uint32_t valid = (uint32_t)std::min<uint64_t>(before, scfq_gzfast::kWindow);
  std::memcpy(window.data() + scfq_gzfast::kWindow - valid, out + before - valid, valid);
  scfq_gzfast::Resume rs;
  rs.open(img, n, bit, window.data(), valid, scfq_crc::crc32(0u, out, (size_t)before), before);
  uint64_t total = before;
  std::vector<uint8_t> buf(chunk);
  for (;;) {
    const int64_t got = rs.next_chunk(buf.data(), chunk);
    if (got < 0) return SCFQ_EGZ;
    if (got == 0) break;
    if (total + (uint64_t)got > cap) return SCFQ_EARG;
    std::memcpy(out + total, buf.data(), (size_t)got);
    total += (uint64_t)got;
  }
  return (int64_t)total;
}

int64_t scfq_debug_read_file(const char* path, void* dst, uint64_t cap, uint64_t chunk) {
  if (!path || (!dst && cap)) return SCFQ_EARG;
  if (chunk == 0) chunk = kDefaultChunk;
  const size_t plen = std::strlen(path);
  const bool is_gz = plen >= 3 && std::memcmp(path + plen - 3, ".gz", 3) == 0;
  std::unique_ptr<Source> src;
  int fd = -1;
  gzFile gz = nullptr;
  if (is_gz) {
    fd = open(path, O_RDONLY);
    struct stat sb;
    if (fd >= 0 && fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && !std::getenv("SCFQ_NO_BGZF") && scfq_bgzf::probe(fd)) {
      src.reset(new BgzfSource(fd, (uint64_t)sb.st_size));
    } else {
      if (fd >= 0) close(fd);
      fd = -1;
      src = open_gz_source(path, chunk, &gz);
      if (!src) return SCFQ_EOPEN;
    }
  } else {
    fd = open(path, O_RDONLY);
    struct stat sb;
    if (fd < 0 || fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) { if (fd >= 0) close(fd); return SCFQ_EOPEN; }
    src.reset(new FdSource(fd, 0, (uint64_t)sb.st_size));
  }
  std::vector<uint8_t> buf(chunk);
  uint64_t total = 0;
  int64_t rc = 0;
  for (;;) {
    const int64_t got = src->fill(buf.data(), chunk);
    if (got < 0) { rc = got; break; }
    if (got == 0) break;
    if (total + (uint64_t)got > cap) { rc = SCFQ_EARG; break; }
    std::memcpy(static_cast<uint8_t*>(dst) + total, buf.data(), (size_t)got);
    total += (uint64_t)got;
  }
  src.reset();
  if (gz) gzclose(gz);
  if (fd >= 0) close(fd);
  return rc < 0 ? rc : (int64_t)total;
}

// ---- diagnostic: time of the scan kernel's load structure alone over a device-resident buffer (milliseconds, best of
// `reps`; < 0 on error). The buffer must be 4 KiB aligned; only whole tiles are streamed.
double scfq_debug_stream_ms(const void* dptr, uint64_t n, int reps) {
  Ctx* c = nullptr;
  SessionLock sl;
  if (get_ctx(&c, sl) || !dptr || ((uintptr_t)dptr & 4095) || n < (uint64_t)scfq::kTile) return -1.0;
  const uint32_t n_tiles = (uint32_t)std::min<uint64_t>(n / scfq::kTile, 0xFFFFFFFFull);
  const uint32_t tpr = pick_tiles_per_range(c, n_tiles);
  const unsigned ranges = (n_tiles + tpr - 1) / tpr, blocks = (ranges + scfq::kWavesPerBlock - 1) / scfq::kWavesPerBlock;
  if (ensure_partials(c, 16, false)) return -1.0;
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1.0;
  float best = -1.0f;
  for (int r = 0; r < std::max(1, reps) + 1; ++r) {
    (void)hipEventRecord(e0, c->compute);
    hipLaunchKernelGGL(scfq::fq_stream_null, dim3(blocks), dim3(256), scfq::kWavesPerBlock * 2 * scfq::kTile, c->compute,
                       static_cast<const uint8_t*>(dptr), n_tiles, tpr, reinterpret_cast<uint32_t*>(c->d_partials));
    (void)hipEventRecord(e1, c->compute);
    if (hipEventSynchronize(e1) != hipSuccess) { best = -1.0f; break; }
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (r > 0 && (best < 0 || ms < best)) best = ms;   // first launch is a warm-up
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return (double)best;
}

// ---- diagnostic: independent byte-serial device kernel (tests only; not used by any counting path) ----
int scfq_debug_partial_simple(const void* dptr, uint64_t n, int prev_byte, scfq_partial* out) {
  if (!out || (!dptr && n)) return SCFQ_EARG;
  Ctx* c = nullptr;
  SessionLock sl;
  int rc = get_ctx(&c, sl);
  if (rc) return rc;
  rc = begin_session(c, false);
  if (rc) return rc;
  if (n) {
    const uint64_t chunks = (n + 255) / 256;
    rc = ensure_partials(c, chunks, false);
    if (rc) return rc;
    hipLaunchKernelGGL(scfq::fq_scan_simple, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, c->compute,
                       static_cast<const uint8_t*>(dptr), n, prev_byte, chunks, c->d_partials);
    HIPCHK(hipGetLastError());
    const uint64_t n_blocks = (chunks + scfq::kFold1 - 1) / scfq::kFold1;
    hipLaunchKernelGGL(scfq::fq_fold_fused, dim3((unsigned)n_blocks), dim3(scfq::kFold1), 0, c->compute, c->d_partials,
                       chunks, c->d_block_partials, c->d_ticket, c->d_state, 1, (uint8_t*)nullptr, (uint8_t*)nullptr,
                       static_cast<const uint8_t*>(dptr), n);
    HIPCHK(hipGetLastError());
    c->fresh = false;
  }
  return end_session(c, false, out, nullptr);
}

}  // extern "C"
