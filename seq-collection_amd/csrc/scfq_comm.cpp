// scfq_comm.cpp — C1, the cross-rank exchange of shard partials, inside the product library (include/sc_fqcount.h:
// scfq_comm_*).  The reference is single-process and has no collective (its seam is one call per file,
// sc.nim:114-116); the exchange exists because the input is byte-range sharded across the GPUs of a node
// (SURVEY.md §8e).  The shard combine is ORDERED and non-commutative, so a sum-allreduce of final counters would be
// wrong: every rank all-gathers the 32-word partials (RCCL ncclAllGather of ncclUint64 over xGMI) and performs the
// same rank-ordered (+) fold on the host.  Payload: world x 256 B (+ world x 8 KiB with the quality histogram).
//
// librccl is loaded on first use (dlopen: 570 MB that a single-GPU host never needs; when the host process has
// already mapped an RCCL under the same soname — PyTorch ships one — that copy is the one that answers).
//
// Every RCCL / HIP call of a communicator runs on ITS OWN worker thread: the caller only ever waits on a condition
// variable with a deadline, so a missing rank or a stuck collective comes back as SCFQ_ERCCL instead of a hang, and a
// host can keep scanning the next shard while an exchange is in flight (scfq_comm_exchange_start / _finish).  The
// worker polls an event with short sleeps instead of blocking in the runtime: a blocking call on a second thread was
// measured (round 1) to delay the scanning thread's launches by ~0.1 ms through the runtime's locks.
//
// Transports: SCFQ_COMM_RCCL (default) and SCFQ_COMM_TCP, host sockets through rank 0 — an explicit opt-in for hosts
// without a shared RCCL fabric and for the CPU-only tests of the multi-process path; never chosen silently.
#include "../../include/sc_fqcount.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types and prototypes only: the library is not linked, see load_rccl()

#include <arpa/inet.h>
#include <dlfcn.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <sys/socket.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <cstdarg>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

using clk = std::chrono::steady_clock;
thread_local char g_cerr[512] = "";

void set_err(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
void set_err(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  std::vsnprintf(g_cerr, sizeof g_cerr, fmt, ap);
  va_end(ap);
  if (std::getenv("SCFQ_VERBOSE")) std::fprintf(stderr, "scfq comm: %s\n", g_cerr);
}

// ---- librccl, resolved at run time ---------------------------------------------------------------------------------
struct Rccl {
  void* h = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommAbort) CommAbort = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;
  std::string path;
};

Rccl* load_rccl() {
  static std::mutex mu;
  static Rccl r;
  static bool tried = false;
  std::lock_guard<std::mutex> lk(mu);
  if (tried) return r.h ? &r : nullptr;
  tried = true;
  std::vector<std::string> names;
  if (const char* e = std::getenv("SCFQ_RCCL_LIB")) names.push_back(e);
  names.insert(names.end(), {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"});
  for (const auto& n : names) {
    r.h = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (r.h) { r.path = n; break; }
  }
  if (!r.h) { set_err("dlopen(librccl.so.1) failed: %s", dlerror()); return nullptr; }
#define SCFQ_SYM(field, name)                                                        \
  r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.h, name));                   \
  if (!r.field) { set_err("librccl: missing symbol %s", name); dlclose(r.h); r.h = nullptr; return nullptr; }
  SCFQ_SYM(GetUniqueId, "ncclGetUniqueId")
  SCFQ_SYM(CommInitRank, "ncclCommInitRank")
  SCFQ_SYM(CommInitAll, "ncclCommInitAll")
  SCFQ_SYM(AllGather, "ncclAllGather")
  SCFQ_SYM(CommDestroy, "ncclCommDestroy")
  SCFQ_SYM(CommAbort, "ncclCommAbort")
  SCFQ_SYM(GetErrorString, "ncclGetErrorString")
  SCFQ_SYM(GetVersion, "ncclGetVersion")
#undef SCFQ_SYM
  return &r;
}

// ---- sockets: the rendezvous (unique id) and the TCP transport -------------------------------------------------------
int ms_left(clk::time_point deadline) {
  const auto d = std::chrono::duration_cast<std::chrono::milliseconds>(deadline - clk::now()).count();
  return d < 0 ? 0 : (int)std::min<long long>(d, 1 << 30);
}

bool io_all(int fd, void* buf, size_t n, bool wr, clk::time_point deadline) {
  uint8_t* p = static_cast<uint8_t*>(buf);
  while (n) {
    pollfd pf{fd, (short)(wr ? POLLOUT : POLLIN), 0};
    const int left = ms_left(deadline);
    const int pr = poll(&pf, 1, left);
    if (pr == 0) { set_err("socket %s timed out", wr ? "send" : "receive"); return false; }
    if (pr < 0) { if (errno == EINTR) continue; set_err("poll: %s", std::strerror(errno)); return false; }
    const ssize_t k = wr ? send(fd, p, n, MSG_NOSIGNAL) : recv(fd, p, n, 0);
    if (k == 0 && !wr) { set_err("peer closed the connection"); return false; }
    if (k < 0) { if (errno == EINTR || errno == EAGAIN) continue; set_err("socket %s: %s", wr ? "send" : "recv", std::strerror(errno)); return false; }
    p += k;
    n -= (size_t)k;
  }
  return true;
}

bool resolve(const char* host, int port, sockaddr_in* sa) {
  std::memset(sa, 0, sizeof *sa);
  sa->sin_family = AF_INET;
  sa->sin_port = htons((uint16_t)port);
  if (!host || !*host) { sa->sin_addr.s_addr = htonl(INADDR_LOOPBACK); return true; }
  if (inet_pton(AF_INET, host, &sa->sin_addr) == 1) return true;
  addrinfo hints{}, *res = nullptr;
  hints.ai_family = AF_INET;
  hints.ai_socktype = SOCK_STREAM;
  if (getaddrinfo(host, nullptr, &hints, &res) != 0 || !res) { set_err("cannot resolve %s", host); return false; }
  sa->sin_addr = reinterpret_cast<sockaddr_in*>(res->ai_addr)->sin_addr;
  freeaddrinfo(res);
  return true;
}

// Star around rank 0: fds[r] on rank 0 is the socket of rank r (fds[0] = -1); on rank r > 0 fds[0] is the socket to rank 0.
struct Star {
  int world = 1, rank = 0;
  std::vector<int> fds;
  // Label of one launch (FNV-1a of SCFQ_RENDEZVOUS_TOKEN; 0 when unset).  It guards against COLLISIONS — two launches that picked
  // the same port, a stale rank of an earlier run — and is NOT authentication: the hash crosses the wire in clear text and can be
  // replayed.  The rendezvous is meant for one node: rank 0 binds to 127.0.0.1 unless the host names another address explicitly,
  // and a host that does so is expected to do it on a network it trusts.
  static uint64_t token() {
    static const uint64_t t = [] {
      const char* e = std::getenv("SCFQ_RENDEZVOUS_TOKEN");
      if (!e || !*e) return (uint64_t)0;
      uint64_t h = 1469598103934665603ull;
      for (; *e; ++e) { h ^= (uint8_t)*e; h *= 1099511628211ull; }
      return h;
    }();
    return t;
  }
  ~Star() { close_all(); }
  void close_all() { for (int& f : fds) if (f >= 0) { close(f); f = -1; } }

  bool open(const char* host, int port, int w, int r, clk::time_point deadline) {
    world = w;
    rank = r;
    fds.assign(rank == 0 ? world : 1, -1);
    if (world == 1) return true;
    sockaddr_in sa;
    if (!resolve(host, port, &sa)) return false;
    const int one = 1;
    if (rank == 0) {
      const int ls = socket(AF_INET, SOCK_STREAM, 0);
      if (ls < 0) { set_err("socket: %s", std::strerror(errno)); return false; }
      setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
      if (bind(ls, reinterpret_cast<sockaddr*>(&sa), sizeof sa) != 0 || listen(ls, world) != 0) {
        set_err("rendezvous: cannot listen on %s:%d: %s", host ? host : "127.0.0.1", port, std::strerror(errno));
        close(ls);
        return false;
      }
      int have = 0;
      while (have < world - 1) {
        pollfd pf{ls, POLLIN, 0};
        const int pr = poll(&pf, 1, ms_left(deadline));
        if (pr == 0) { set_err("rendezvous: %d of %d ranks arrived before the deadline", have + 1, world); close(ls); return false; }
        if (pr < 0) { if (errno == EINTR) continue; set_err("poll: %s", std::strerror(errno)); close(ls); return false; }
        const int fd = accept(ls, nullptr, nullptr);
        if (fd < 0) continue;
        setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
        // (a connection that is not one of ours — a port scan, a health check — gets one second to say hello, not the whole
        // deadline; with SCFQ_RENDEZVOUS_TOKEN set in the launcher's environment the hello must carry that token too)
        int32_t hello[4] = {0, 0, 0, 0};
        const auto hello_deadline = std::min(deadline, clk::now() + std::chrono::seconds(1));
        if (!io_all(fd, hello, sizeof hello, false, hello_deadline) || hello[0] != 0x53434651) {
          close(fd);     // not one of ours (or too slow to say so: a genuine rank sees the close and connects again)
          continue;
        }
        // A peer that speaks the protocol gets a VERDICT word back, so that a rank that was turned away learns it here and now —
        // not later, as an unrelated failure of the id exchange: 1 = slot taken, 2 = wrong token, 3 = that rank is already here,
        // 4 = no such rank in this world.
        int32_t verdict = 1;
        if (hello[2] != (int32_t)(uint32_t)token() || hello[3] != (int32_t)(uint32_t)(token() >> 32)) verdict = 2;
        else if (hello[1] <= 0 || hello[1] >= world) verdict = 4;
        else if (fds[hello[1]] >= 0) verdict = 3;
        const bool told = io_all(fd, &verdict, sizeof verdict, true, std::min(deadline, clk::now() + std::chrono::seconds(1)));
        if (verdict != 1 || !told) { close(fd); continue; }
        fds[hello[1]] = fd;
        ++have;
      }
      close(ls);
      return true;
    }
    for (;;) {
      const int fd = socket(AF_INET, SOCK_STREAM, 0);
      if (fd < 0) { set_err("socket: %s", std::strerror(errno)); return false; }
      if (connect(fd, reinterpret_cast<sockaddr*>(&sa), sizeof sa) == 0) {
        setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
        int32_t hello[4] = {0x53434651, rank, (int32_t)(uint32_t)token(), (int32_t)(uint32_t)(token() >> 32)};
        int32_t verdict = 0;
        if (io_all(fd, hello, sizeof hello, true, deadline) && io_all(fd, &verdict, sizeof verdict, false, deadline)) {
          if (verdict == 1) { fds[0] = fd; return true; }
          close(fd);
          set_err("rendezvous: rank 0 at %s:%d turned rank %d away: %s", host ? host : "127.0.0.1", port, rank,
                  verdict == 2 ? "SCFQ_RENDEZVOUS_TOKEN differs (another launch on the same port?)"
                  : verdict == 3 ? "a process with this rank is already connected"
                  : verdict == 4 ? "no such rank in its world (different --shard-world?)" : "unknown verdict");
          return false;
        }
        // (no verdict: whoever listens there closed the connection — rank 0 does that to a hello that took more than a second — or
        // is not rank 0 at all: try again until the deadline)
      }
      close(fd);
      if (ms_left(deadline) == 0) { set_err("rendezvous: rank 0 at %s:%d did not answer before the deadline", host ? host : "127.0.0.1", port); return false; }
      std::this_thread::sleep_for(std::chrono::milliseconds(20));
    }
  }

  bool bcast(void* buf, size_t n, clk::time_point deadline) {
    if (world == 1) return true;
    if (rank == 0) {
      for (int r = 1; r < world; ++r) if (!io_all(fds[r], buf, n, true, deadline)) return false;
      return true;
    }
    return io_all(fds[0], buf, n, false, deadline);
  }

  // all[r * words ...] = rank r's row, on every rank
  bool allgather(const uint64_t* mine, uint32_t words, uint64_t* all, clk::time_point deadline) {
    std::memcpy(all + (size_t)rank * words, mine, words * sizeof(uint64_t));
    if (world == 1) return true;
    if (rank == 0) {
      for (int r = 1; r < world; ++r) if (!io_all(fds[r], all + (size_t)r * words, words * sizeof(uint64_t), false, deadline)) return false;
      for (int r = 1; r < world; ++r) if (!io_all(fds[r], all, (size_t)world * words * sizeof(uint64_t), true, deadline)) return false;
      return true;
    }
    if (!io_all(fds[0], const_cast<uint64_t*>(mine), words * sizeof(uint64_t), true, deadline)) return false;
    return io_all(fds[0], all, (size_t)world * words * sizeof(uint64_t), false, deadline);
  }
};

constexpr uint32_t kMaxWords = 8448;   // a partial with its histogram (1056 words), or a stretch's window map with its header (8197: scfq_count_file_sharded)

// A caller waits a little longer than the worker's own deadline, so that the worker's message (which says WHY) normally wins.
// SCFQ_COMM_TAKE_SLACK_MS / SCFQ_COMM_TEST_DELAY_MS (the worker sleeps before every gather) exist for the test of the case
// where it does not: a worker stuck inside the runtime whose answer arrives after its caller gave up.
int take_slack_ms() { static const int v = [] { const char* e = std::getenv("SCFQ_COMM_TAKE_SLACK_MS"); return e ? std::atoi(e) : 5000; }(); return v; }
int test_delay_ms() { static const int v = [] { const char* e = std::getenv("SCFQ_COMM_TEST_DELAY_MS"); return e ? std::atoi(e) : 0; }(); return v; }
int caller_wait_ms(int timeout_ms) { return timeout_ms > 0 ? timeout_ms + take_slack_ms() : 300000 + take_slack_ms(); }

// RCCL 2.27 prints a version banner on STDOUT when the first communicator of a process comes up; this library never
// writes to stdout (a host's stdout is its TSV, src/fq_count.nim:53).  While a communicator is being created, and until its
// first collective has run, descriptor 1 points at descriptor 2.  The redirect is REFERENCE COUNTED: the first creator in
// flight installs it, the last one restores descriptor 1, and the mutex only guards that bookkeeping — it is never held across
// the blocking ncclCommInitRank, so two threads of one process may bring up two ranks of the same world at the same time
// (thread per GPU).  What another thread of the host writes to stdout inside that window lands on stderr: hosts that print
// rows from several threads create their communicators first (scfq_prepare(), which `sc fq-count --devices=a,b` calls before
// its first row) — include/sc_fqcount.h says so next to scfq_comm_init_*.
struct StdoutToStderr {
  static std::mutex& mu() { static std::mutex m; return m; }
  static int& users() { static int n = 0; return n; }
  static int& saved() { static int fd = -1; return fd; }
  StdoutToStderr() {
    std::lock_guard<std::mutex> lk(mu());
    if (users()++ == 0) {
      std::fflush(stdout);
      saved() = dup(1);
      if (saved() >= 0) dup2(2, 1);
    }
  }
  ~StdoutToStderr() {
    std::lock_guard<std::mutex> lk(mu());
    if (--users() == 0 && saved() >= 0) { std::fflush(stdout); dup2(saved(), 1); close(saved()); saved() = -1; }
  }
  StdoutToStderr(const StdoutToStderr&) = delete;
  StdoutToStderr& operator=(const StdoutToStderr&) = delete;
};

struct Job {
  enum Kind { kInitRank, kAdopt, kGather, kStop } kind = kGather;
  uint64_t seq = 0;            // every job carries a ticket and its answer carries the same one: an answer that arrives after
                               // its caller gave up (deadline) is dropped instead of being handed to the NEXT caller
  std::vector<uint64_t> row;
  bool with_hist = false;
  int timeout_ms = 0;
  ncclUniqueId id{};
};
struct Done {
  uint64_t seq = 0;
  int rc = SCFQ_OK;
  std::vector<uint64_t> all;   // world x words
  uint32_t words = 0;
  bool with_hist = false;
  std::string err;
};

}  // namespace

struct scfq_comm {
  int world = 1, rank = 0, dev = 0, transport = SCFQ_COMM_RCCL;
  Rccl* rccl = nullptr;
  ncclComm_t nc = nullptr;
  hipStream_t st = nullptr;
  hipEvent_t ev = nullptr;
  uint64_t *h_in = nullptr, *h_out = nullptr, *d_in = nullptr, *d_out = nullptr;
  Star star;
  std::thread worker;
  std::mutex mu;
  std::condition_variable cv_in, cv_out;
  std::deque<Job> in;
  std::deque<Done> out;
  std::atomic<bool> broken{false};
  uint64_t exchanges = 0;
  uint64_t next_seq = 1;       // (under mu) ticket of the next job posted
  std::deque<uint64_t> waiting;   // (under mu) tickets of posted jobs nobody has collected yet, oldest first

  int hip_fail(const char* what, hipError_t e, Done* d) {
    char b[256];
    std::snprintf(b, sizeof b, "%s -> %s", what, hipGetErrorString(e));
    d->err = b;
    return d->rc = SCFQ_EHIP;
  }
  int nccl_fail(const char* what, ncclResult_t e, Done* d) {
    char b[256];
    std::snprintf(b, sizeof b, "%s -> %s (rank %d of %d, device %d, %s)", what, rccl->GetErrorString(e), rank, world, dev, rccl->path.c_str());
    d->err = b;
    return d->rc = SCFQ_ERCCL;
  }

  int device_setup(Done* d) {
    hipError_t e;
    if ((e = hipSetDevice(dev)) != hipSuccess) return hip_fail("hipSetDevice", e, d);
    if ((e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking)) != hipSuccess) return hip_fail("hipStreamCreateWithFlags", e, d);
    if ((e = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess) return hip_fail("hipEventCreateWithFlags", e, d);
    if ((e = hipHostMalloc(&h_in, kMaxWords * sizeof(uint64_t), hipHostMallocDefault)) != hipSuccess) return hip_fail("hipHostMalloc", e, d);
    if ((e = hipHostMalloc(&h_out, (size_t)world * kMaxWords * sizeof(uint64_t), hipHostMallocDefault)) != hipSuccess) return hip_fail("hipHostMalloc", e, d);
    if ((e = hipMalloc(&d_in, kMaxWords * sizeof(uint64_t))) != hipSuccess) return hip_fail("hipMalloc", e, d);
    if ((e = hipMalloc(&d_out, (size_t)world * kMaxWords * sizeof(uint64_t))) != hipSuccess) return hip_fail("hipMalloc", e, d);
    return SCFQ_OK;
  }

  void gather_rccl(const Job& j, Done* d) {
    const uint32_t words = (uint32_t)j.row.size();
    const auto deadline = clk::now() + std::chrono::milliseconds(j.timeout_ms > 0 ? j.timeout_ms : 300000);
    std::memcpy(h_in, j.row.data(), words * sizeof(uint64_t));
    hipError_t e;
    if ((e = hipMemcpyAsync(d_in, h_in, words * sizeof(uint64_t), hipMemcpyHostToDevice, st)) != hipSuccess) { hip_fail("hipMemcpyAsync(H2D)", e, d); return; }
    ncclResult_t r = rccl->AllGather(d_in, d_out, words, ncclUint64, nc, st);
    if (r != ncclSuccess) { nccl_fail("ncclAllGather", r, d); return; }
    if ((e = hipMemcpyAsync(h_out, d_out, (size_t)world * words * sizeof(uint64_t), hipMemcpyDeviceToHost, st)) != hipSuccess) { hip_fail("hipMemcpyAsync(D2H)", e, d); return; }
    if ((e = hipEventRecord(ev, st)) != hipSuccess) { hip_fail("hipEventRecord", e, d); return; }
    for (;;) {
      e = hipEventQuery(ev);
      if (e == hipSuccess) break;
      if (e != hipErrorNotReady) { hip_fail("hipEventQuery", e, d); return; }
      if (ms_left(deadline) == 0) {
        broken = true;
        char b[200];
        std::snprintf(b, sizeof b, "ncclAllGather did not complete within the deadline (rank %d of %d): a rank is missing or stuck", rank, world);
        d->err = b;
        d->rc = SCFQ_ERCCL;
        (void)rccl->CommAbort(nc);
        nc = nullptr;
        return;
      }
      std::this_thread::sleep_for(std::chrono::microseconds(30));
    }
    d->all.assign(h_out, h_out + (size_t)world * words);
  }

  void run() {
    for (;;) {
      Job j;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_in.wait(lk, [&] { return !in.empty(); });
        j = std::move(in.front());
        in.pop_front();
      }
      if (j.kind == Job::kStop) return;
      if (j.kind == Job::kGather && test_delay_ms() > 0) std::this_thread::sleep_for(std::chrono::milliseconds(test_delay_ms()));
      Done d;
      d.seq = j.seq;
      d.words = (uint32_t)j.row.size();
      d.with_hist = j.with_hist;
      if (j.kind == Job::kInitRank || (j.kind == Job::kAdopt && transport == SCFQ_COMM_RCCL)) {
        if (device_setup(&d) == SCFQ_OK && j.kind == Job::kInitRank) {
          ncclResult_t r = rccl->CommInitRank(&nc, world, j.id, rank);
          if (r != ncclSuccess) nccl_fail("ncclCommInitRank", r, &d);
        }
        if (d.rc == SCFQ_OK) {
          // one collective before the communicator is handed out: connections are set up (and anything RCCL wants to
          // print is printed) here, under the creator's deadline, not inside a host's first timed exchange
          Job w;
          w.row.assign(1, (uint64_t)rank);
          w.timeout_ms = j.timeout_ms;
          Done dw;
          gather_rccl(w, &dw);
          if (dw.rc) { d.rc = dw.rc; d.err = dw.err; }
          else for (int r = 0; r < world; ++r) if (dw.all[(size_t)r] != (uint64_t)r) { d.rc = SCFQ_ERCCL; d.err = "warm-up all-gather returned the wrong ranks"; }
        }
      } else if (j.kind == Job::kAdopt) {
        // TCP transport: nothing to set up on a device
      } else if (broken) {
        d.rc = SCFQ_ERCCL;
        d.err = "communicator is broken (an earlier exchange failed or timed out)";
      } else if (transport == SCFQ_COMM_TCP) {
        d.all.resize((size_t)world * d.words);
        const auto deadline = clk::now() + std::chrono::milliseconds(j.timeout_ms > 0 ? j.timeout_ms : 300000);
        if (!star.allgather(j.row.data(), d.words, d.all.data(), deadline)) { d.rc = SCFQ_ERCCL; d.err = g_cerr; broken = true; }
      } else {
        gather_rccl(j, &d);
      }
      {
        std::lock_guard<std::mutex> lk(mu);
        out.push_back(std::move(d));
      }
      cv_out.notify_all();
    }
  }

  void post(Job&& j) {
    {
      std::lock_guard<std::mutex> lk(mu);
      j.seq = next_seq++;
      if (j.kind != Job::kStop) waiting.push_back(j.seq);
      in.push_back(std::move(j));
    }
    cv_in.notify_one();
  }

  // the answer to the OLDEST job nobody has collected yet; SCFQ_ERCCL when the deadline passes first.  The communicator is then
  // marked broken and that job's ticket is retired: when its answer arrives after all it is dropped, so the next start / finish
  // pair can never be handed the previous exchange's rows.
  int take(Done* d, int timeout_ms) {
    std::unique_lock<std::mutex> lk(mu);
    if (waiting.empty()) { set_err("no exchange in flight on this communicator (rank %d of %d)", rank, world); return SCFQ_EARG; }
    const uint64_t want = waiting.front();
    auto mine = [&] {
      while (!out.empty() && out.front().seq < want) out.pop_front();      // answers whose callers gave up
      return !out.empty() && out.front().seq == want;
    };
    const bool ok = cv_out.wait_for(lk, std::chrono::milliseconds(timeout_ms > 0 ? timeout_ms : 300000), mine);
    waiting.pop_front();
    if (!ok) {
      broken = true;
      set_err("no answer from the exchange worker within %d ms (rank %d of %d): a rank is missing or the collective is stuck", timeout_ms, rank, world);
      return SCFQ_ERCCL;
    }
    *d = std::move(out.front());
    out.pop_front();
    if (d->rc) set_err("%s", d->err.c_str());
    return d->rc;
  }
};

namespace {

int fold_rows(const Done& d, int world, scfq_partial* folded, uint64_t* hist_folded) {
  const uint32_t words = d.words;
  scfq_partial acc;
  std::vector<uint64_t> acc_h(d.with_hist ? SCFQ_HIST_WORDS : 0);
  scfq_partial_identity(&acc, d.with_hist ? acc_h.data() : nullptr);
  for (int r = 0; r < world; ++r) {
    scfq_partial p;
    std::memcpy(&p, d.all.data() + (size_t)r * words, sizeof p);
    const uint64_t* h = d.with_hist ? d.all.data() + (size_t)r * words + SCFQ_PARTIAL_WORDS : nullptr;
    scfq_partial_combine(&acc, &p, d.with_hist ? acc_h.data() : nullptr, h);
  }
  *folded = acc;
  if (d.with_hist && hist_folded) std::memcpy(hist_folded, acc_h.data(), SCFQ_HIST_WORDS * sizeof(uint64_t));
  return SCFQ_OK;
}

void destroy_impl(scfq_comm* c) {
  if (!c) return;
  if (c->worker.joinable()) {
    Job j;
    j.kind = Job::kStop;
    c->post(std::move(j));
    if (c->broken) {
      // the worker may sit inside a call that never returns: give it a moment, then let it go
      for (int k = 0; k < 100; ++k) {
        { std::lock_guard<std::mutex> lk(c->mu); if (c->in.empty()) break; }
        std::this_thread::sleep_for(std::chrono::milliseconds(10));
      }
      bool drained;
      { std::lock_guard<std::mutex> lk(c->mu); drained = c->in.empty(); }
      if (drained) c->worker.join();
      else { c->worker.detach(); return; }   // leak on purpose: the stuck thread still references *c
    } else {
      c->worker.join();
    }
  }
  if (c->transport == SCFQ_COMM_RCCL) {
    (void)hipSetDevice(c->dev);
    if (c->nc && c->rccl) { if (c->broken) (void)c->rccl->CommAbort(c->nc); else (void)c->rccl->CommDestroy(c->nc); }
    if (c->st) (void)hipStreamDestroy(c->st);
    if (c->ev) (void)hipEventDestroy(c->ev);
    if (c->h_in) (void)hipHostFree(c->h_in);
    if (c->h_out) (void)hipHostFree(c->h_out);
    if (c->d_in) (void)hipFree(c->d_in);
    if (c->d_out) (void)hipFree(c->d_out);
  }
  delete c;
}

int start_and_wait(scfq_comm* c, Job&& j, int timeout_ms) {
  c->worker = std::thread([c] { c->run(); });
  c->post(std::move(j));
  Done d;
  return c->take(&d, caller_wait_ms(timeout_ms));     // the worker's own deadline fires first and says why
}

}  // namespace

extern "C" {

const char* scfq_comm_error_detail(void) { return g_cerr; }

int scfq_comm_unique_id(void* id, uint64_t cap) {
  if (!id || cap < SCFQ_COMM_ID_BYTES) return SCFQ_EARG;
  Rccl* r = load_rccl();
  if (!r) return SCFQ_ERCCL;
  ncclUniqueId u;
  const ncclResult_t e = r->GetUniqueId(&u);
  if (e != ncclSuccess) { set_err("ncclGetUniqueId -> %s", r->GetErrorString(e)); return SCFQ_ERCCL; }
  static_assert(sizeof u == SCFQ_COMM_ID_BYTES, "RCCL unique id size");
  std::memcpy(id, &u, sizeof u);
  return SCFQ_OK;
}

int scfq_comm_init_rank(const void* id, int world, int rank, int device, int timeout_ms, scfq_comm** out) {
  if (!id || !out || world < 1 || rank < 0 || rank >= world || device < 0) return SCFQ_EARG;
  *out = nullptr;
  Rccl* r = load_rccl();
  if (!r) return SCFQ_ERCCL;
  auto* c = new scfq_comm;
  c->world = world;
  c->rank = rank;
  c->dev = device;
  c->rccl = r;
  Job j;
  j.kind = Job::kInitRank;
  j.timeout_ms = timeout_ms;
  std::memcpy(&j.id, id, sizeof j.id);
  StdoutToStderr quiet;
  const int rc = start_and_wait(c, std::move(j), timeout_ms);
  if (rc) { c->broken = true; destroy_impl(c); return rc; }
  *out = c;
  return SCFQ_OK;
}

int scfq_comm_init_rendezvous(const char* host, int port, int world, int rank, int device, int transport, int timeout_ms,
                              scfq_comm** out) {
  if (!out || world < 1 || rank < 0 || rank >= world || port <= 0 || port > 65535) return SCFQ_EARG;
  if (transport != SCFQ_COMM_RCCL && transport != SCFQ_COMM_TCP) return SCFQ_EARG;
  *out = nullptr;
  const auto deadline = clk::now() + std::chrono::milliseconds(timeout_ms > 0 ? timeout_ms : 300000);
  if (transport == SCFQ_COMM_TCP) {
    auto* c = new scfq_comm;
    c->world = world;
    c->rank = rank;
    c->dev = device;
    c->transport = SCFQ_COMM_TCP;
    if (!c->star.open(host, port, world, rank, deadline)) { delete c; return SCFQ_ERCCL; }
    Job j;
    j.kind = Job::kAdopt;
    const int rc = start_and_wait(c, std::move(j), timeout_ms);
    if (rc) { c->broken = true; destroy_impl(c); return rc; }
    *out = c;
    return SCFQ_OK;
  }
  if (device < 0) return SCFQ_EARG;
  uint8_t id[SCFQ_COMM_ID_BYTES] = {0};
  if (rank == 0) {
    const int rc = scfq_comm_unique_id(id, sizeof id);
    if (rc) return rc;
  }
  {
    Star s;      // the sockets only carry the unique id; RCCL opens its own connections from it
    if (!s.open(host, port, world, rank, deadline) || !s.bcast(id, sizeof id, deadline)) return SCFQ_ERCCL;
  }
  return scfq_comm_init_rank(id, world, rank, device, ms_left(deadline) > 0 ? ms_left(deadline) : 1, out);
}

int scfq_comm_init_all(int n, const int32_t* device_ids, int timeout_ms, scfq_comm** out) {
  if (n < 1 || !device_ids || !out) return SCFQ_EARG;
  for (int k = 0; k < n; ++k) out[k] = nullptr;
  Rccl* r = load_rccl();
  if (!r) return SCFQ_ERCCL;
  std::vector<ncclComm_t> ncs(n, nullptr);
  std::vector<int> devs(device_ids, device_ids + n);
  int prev = 0;
  (void)hipGetDevice(&prev);
  StdoutToStderr quiet;
  const ncclResult_t e = r->CommInitAll(ncs.data(), n, devs.data());
  (void)hipSetDevice(prev);
  if (e != ncclSuccess) { set_err("ncclCommInitAll(%d devices) -> %s", n, r->GetErrorString(e)); return SCFQ_ERCCL; }
  // every worker first runs one collective (see run()): all of them are started before any is waited for
  for (int k = 0; k < n; ++k) {
    auto* c = new scfq_comm;
    c->world = n;
    c->rank = k;
    c->dev = devs[k];
    c->rccl = r;
    c->nc = ncs[k];
    out[k] = c;
    c->worker = std::thread([c] { c->run(); });
    Job j;
    j.kind = Job::kAdopt;
    j.timeout_ms = timeout_ms;
    c->post(std::move(j));
  }
  int rc = SCFQ_OK;
  for (int k = 0; k < n; ++k) {
    Done d;
    const int rk = out[k]->take(&d, caller_wait_ms(timeout_ms));
    if (rk) { out[k]->broken = true; if (!rc) rc = rk; }
  }
  if (rc) {
    for (int k = 0; k < n; ++k) { destroy_impl(out[k]); out[k] = nullptr; }
  }
  return rc;
}

int scfq_comm_is_broken(const scfq_comm* c) { return (c && c->broken) ? 1 : 0; }
int scfq_comm_world(const scfq_comm* c) { return c ? c->world : SCFQ_EARG; }
int scfq_comm_rank(const scfq_comm* c) { return c ? c->rank : SCFQ_EARG; }
const char* scfq_comm_transport(const scfq_comm* c) {
  if (!c) return "";
  if (c->transport == SCFQ_COMM_TCP) return "tcp";
  static thread_local char b[160];
  int v = 0;
  if (c->rccl) (void)c->rccl->GetVersion(&v);
  std::snprintf(b, sizeof b, "RCCL %d.%d.%d", v / 10000, (v / 100) % 100, v % 100);
  return b;
}

int scfq_comm_allgather_u64(scfq_comm* c, const uint64_t* mine, uint32_t words, uint64_t* all, int timeout_ms) {
  if (!c || !mine || !all || words == 0 || words > kMaxWords) return SCFQ_EARG;
  if (c->broken) { set_err("communicator is broken (an earlier exchange failed or timed out): destroy it and create a new one"); return SCFQ_ERCCL; }
  Job j;
  j.row.assign(mine, mine + words);
  j.timeout_ms = timeout_ms;
  c->post(std::move(j));
  Done d;
  const int rc = c->take(&d, caller_wait_ms(timeout_ms));
  if (rc) return rc;
  std::memcpy(all, d.all.data(), d.all.size() * sizeof(uint64_t));
  return SCFQ_OK;
}

int scfq_comm_exchange_start(scfq_comm* c, const scfq_partial* mine, const uint64_t* hist, int timeout_ms) {
  if (!c || !mine) return SCFQ_EARG;
  if (c->broken) { set_err("communicator is broken (an earlier exchange failed or timed out): destroy it and create a new one"); return SCFQ_ERCCL; }
  Job j;
  j.with_hist = hist != nullptr;
  j.row.resize(SCFQ_PARTIAL_WORDS + (hist ? SCFQ_HIST_WORDS : 0));
  static_assert(sizeof(scfq_partial) == SCFQ_PARTIAL_WORDS * sizeof(uint64_t), "partial is 32 words");
  std::memcpy(j.row.data(), mine, sizeof *mine);
  if (hist) std::memcpy(j.row.data() + SCFQ_PARTIAL_WORDS, hist, SCFQ_HIST_WORDS * sizeof(uint64_t));
  j.timeout_ms = timeout_ms;
  c->post(std::move(j));
  c->exchanges += 1;
  return SCFQ_OK;
}

int scfq_comm_exchange_finish(scfq_comm* c, scfq_partial* folded, uint64_t* hist_folded, int timeout_ms) {
  if (!c || !folded) return SCFQ_EARG;
  {
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->broken && c->waiting.empty()) { set_err("communicator is broken (an earlier exchange failed or timed out)"); return SCFQ_ERCCL; }
  }
  Done d;
  const int rc = c->take(&d, caller_wait_ms(timeout_ms));
  if (rc) return rc;
  return fold_rows(d, c->world, folded, hist_folded);
}

int scfq_comm_exchange(scfq_comm* c, const scfq_partial* mine, const uint64_t* hist, scfq_partial* folded,
                       uint64_t* hist_folded, int timeout_ms) {
  const int rc = scfq_comm_exchange_start(c, mine, hist, timeout_ms);
  if (rc) return rc;
  return scfq_comm_exchange_finish(c, folded, hist_folded, timeout_ms);
}

int scfq_comm_destroy(scfq_comm* c) {
  destroy_impl(c);
  return SCFQ_OK;
}

}  // extern "C"
