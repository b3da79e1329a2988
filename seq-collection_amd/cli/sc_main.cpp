// sc_main.cpp — C++ host for the `sc fq-count` command over the C ABI (include/sc_fqcount.h).
//
// The reference host is Nim (sc.nim + src/fq_count.nim); no Nim toolchain exists in this image, so the
// host side above the C ABI is written in C++ and mirrors the reference's surface for this path:
//   sc.nim:103-116                 command "fq-count": -t/--header, -b/--basename, -a/--absolute, [fastq ...]
//   sc.nim:274-293                 stdin '-' -> literal "STDIN"; bare `sc` -> help
//   src/fq_count.nim:14-53         one TSV row per file, argv order; "Unable to open file" exit 2
//   src/utils/helpers.nim:29-34    "Error <code>: <msg>" in red on stderr, exit <code>
//   src/utils/helpers.nim:200-224  header / basename / absolute columns
// Everything between "open the file" and "format the row" is one call into libsc_fqcount_hip.so.
// GPU-only additions are new long options and never change the reference's 5-column row.
#include "../../include/sc_fqcount.h"
#include "../../include/sc_fqcount_debug.h"      // --stats: the library's stage marks

#include <limits.h>
#include <signal.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <mutex>
#include <thread>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static const char* kVersion = "0.0.2-mi355x";
static const char* kHeader = "reads\tgc_content\tgc_bases\tn_bases\tbases";   // src/fq_count.nim:7-11

static void error_msg(const std::string& msg, int code) {           // helpers.nim:29-30 (colorize fgRed)
  std::fprintf(stderr, "\x1b[31mError %d: %s\x1b[0m\n", code, msg.c_str());
}
[[noreturn]] static void quit_error(const std::string& msg, int code = 1) {   // helpers.nim:32-34
  error_msg(msg, code);
  std::exit(code);
}

static std::string join_nonempty(const std::vector<std::string>& parts) {     // filterIt(it.len > 0).join("\t")
  std::string out;
  for (const auto& p : parts) {
    if (p.empty()) continue;
    if (!out.empty()) out += '\t';
    out += p;
  }
  return out;
}

static std::string output_header(const std::string& header, bool basename, bool absolute) {   // helpers.nim:200-208
  return join_nonempty({header, basename ? "basename" : "", absolute ? "absolute" : ""});
}

static std::string last_path_part(std::string p) {   // Nim os.lastPathPart
  while (p.size() > 1 && p.back() == '/') p.pop_back();
  if (p == "/") return "";
  const size_t k = p.find_last_of('/');
  return k == std::string::npos ? p : p.substr(k + 1);
}

static std::string normalize_join(const std::string& root, const std::string& rel) {   // Nim os.absolutePath/joinPath
  std::vector<std::string> comp;
  auto feed = [&](const std::string& s) {
    size_t i = 0;
    while (i <= s.size()) {
      size_t j = s.find('/', i);
      if (j == std::string::npos) j = s.size();
      std::string c = s.substr(i, j - i);
      if (c == "..") { if (!comp.empty()) comp.pop_back(); }
      else if (!c.empty() && c != ".") comp.push_back(c);
      i = j + 1;
    }
  };
  feed(root);
  feed(rel);
  std::string out;
  for (const auto& c : comp) { out += '/'; out += c; }
  return out.empty() ? "/" : out;
}

static std::string absolute_path(const std::string& p) {
  if (!p.empty() && p[0] == '/') return p;          // absolute inputs are returned verbatim
  char cwd[PATH_MAX];
  if (!getcwd(cwd, sizeof cwd)) return p;
  return normalize_join(cwd, p);
}

static std::string get_absolute(const std::string& path) {   // helpers.nim:210-213
  struct stat sb;
  if (lstat(path.c_str(), &sb) == 0 && S_ISLNK(sb.st_mode)) {
    char buf[PATH_MAX];
    ssize_t n = readlink(path.c_str(), buf, sizeof buf - 1);
    if (n >= 0) { buf[n] = 0; return absolute_path(buf); }   // relative targets resolve against the CWD (reference quirk)
  }
  return absolute_path(path);
}

static std::string output_w_fnames(const std::string& line, const std::string& path, bool basename, bool absolute) {
  return join_nonempty({line, basename ? last_path_part(path) : "", absolute ? get_absolute(path) : ""});   // helpers.nim:215-224
}

static void help_fq_count(FILE* f) {   // docs/fq-count.md:5-19
  std::fputs(
      "Counts lines in a FASTQ\n\n"
      "Usage:\n"
      "  fq-count [options] [fastq ...]\n\n"
      "Arguments:\n"
      "  [fastq ...]      Input FASTQ\n\n"
      "Options:\n"
      "  -t, --header               Output the header\n"
      "  -b, --basename             Add basename column\n"
      "  -a, --absolute             Add column for absolute path\n"
      "  -h, --help                 Show this help\n"
      "\nMI355X options (additions; the TSV row is unchanged):\n"
      "      --devices=LIST         Comma-separated HIP device ids to shard each file across (default: current device)\n"
      "      --struct-check         Report header lines not starting '@' / separator lines not starting '+' on stderr\n"
      "      --qual-hist            Print the quality-byte histogram on stderr\n"
      "      --stats                Print bytes / device milliseconds / GB/s as JSON on stderr\n"
      "      --jobs=N               Keep up to N files in flight (default: 2; rows still come out in argument order)\n"
      "      --shard-rank=R --shard-world=W --rendezvous=HOST:PORT [--transport=rccl|tcp]\n"
      "                             One process per GPU: this process scans byte range R of W of every file on its device\n"
      "                             (--devices=ID, default R), the partials are exchanged (RCCL all-gather), rank 0 prints\n"
      "                             the rows. Defaults come from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT.\n",
      f);
}

static void help_top(FILE* f) {
  std::fprintf(f,
               "sc (%s) — MI355X-native host for the fq-count path of seq-collection\n\n"
               "Usage:\n  sc COMMAND\n\nCommands:\n\nFASTQ\n  fq-count         Counts lines in a FASTQ\n"
               "  fq-dedup         Removes exact duplicates from FASTQ Files\n  fq-meta          Output metadata for FASTQ\n\n"
               "Options:\n  -h, --help                 Show this help\n  -v, --version              Show version\n"
               "      --debug                Debug mode\n",
               kVersion);
}

static bool stdin_is_fifo() {   // sc.nim:50-53
  struct stat st;
  return fstat(0, &st) == 0 && S_ISFIFO(st.st_mode);
}

// One file's outcome: what proc fq_count would have echoed / which error it would have quit with.
struct FileResult {
  int exit_code = 0;          // 0, or the quit_error code
  std::string error;          // message for quit_error
  std::string row;            // stdout line (without newline)
  std::string extra;          // stderr lines of the MI355X additions
};

// proc fq_count*(fastq: string, basename: bool, absolute: bool)      src/fq_count.nim:14  (compute part)
static FileResult fq_count_compute(const std::string& fastq, bool basename, bool absolute, const scfq_opts& opts, bool stats,
                                   scfq_comm* comm = nullptr) {
  FileResult r;
  if (fastq.size() < 3) { r.exit_code = 1; r.error = "index out of bounds"; return r; }   // fastq[^3 .. ^1] raises; sc.nim:299-305 -> exit 1
  scfq_counts c;
  std::memset(&c, 0, sizeof c);
  c.struct_size = sizeof c;
  // one process per GPU: this rank's byte range, exchange, every rank gets the whole file's counters
  const int rc = comm ? scfq_count_file_sharded(fastq.c_str(), &opts, comm, &c) : scfq_count_file(fastq.c_str(), &opts, &c);
  if (rc == SCFQ_EOPEN) {
    const bool gz = fastq.compare(fastq.size() - 3, 3, ".gz") == 0;
    // plain: stream == nil -> quit_error(..., 2) (fq_count.nim:35-36); .gz: the stream constructor raises -> exit 1 (sc.nim:299-305)
    r.exit_code = gz ? 1 : 2;
    r.error = "Unable to open file: " + fastq;
    return r;
  }
  if (rc != SCFQ_OK) {
    r.exit_code = 1;
    r.error = std::string(scfq_strerror(rc));
    const char* d = scfq_last_error_detail();
    if (d && *d) { r.error += ": "; r.error += d; }
    return r;
  }
  char row[256];
  scfq_format_tsv(&c, row, sizeof row);
  r.row = output_w_fnames(row, fastq, basename, absolute);
  char buf[512];
  if (opts.flags & SCFQ_STRUCT_CHECK) {
    std::snprintf(buf, sizeof buf, "%s\tbad_at=%llu\tbad_plus=%llu\n", fastq.c_str(), (unsigned long long)c.bad_at, (unsigned long long)c.bad_plus);
    r.extra += buf;
  }
  if (opts.flags & SCFQ_QUAL_HIST) {
    r.extra += fastq + "\tqual_hist";
    for (int v = 0; v < 256; ++v)
      if (c.qual_hist[v]) { std::snprintf(buf, sizeof buf, "\t%d:%llu", v, (unsigned long long)c.qual_hist[v]); r.extra += buf; }
    r.extra += "\n";
  }
  if (stats) {
    scfq_timing t;
    std::memset(&t, 0, sizeof t);
    t.struct_size = sizeof t;
    scfq_last_timing(&t);
    const double gbs = t.scan_kernel_ms > 0 ? (double)t.scan_bytes / (t.scan_kernel_ms * 1e-3) / 1e9 : 0.0;
    char js[1024];
    std::snprintf(js, sizeof js,
                  "{\"file\": \"%s\", \"input_bytes\": %llu, \"scan_kernel_ms\": %.4f, \"fold_kernel_ms\": %.4f, "
                  "\"scan_launches\": %llu, \"scan_GBps\": %.1f, \"hbm_peak_GBps\": 8000, \"roofline_frac\": %.4f, "
                  "\"host_fill_ms\": %.2f, \"ingest_wall_ms\": %.2f, \"h2d_bytes\": %llu, \"device_bytes_high_water\": %llu, \"stages_ms\": ",
                  fastq.c_str(), (unsigned long long)c.input_bytes, t.scan_kernel_ms, t.fold_kernel_ms,
                  (unsigned long long)t.scan_launches, gbs, gbs / 8000.0, t.host_fill_ms, t.ingest_wall_ms,
                  (unsigned long long)t.h2d_bytes, (unsigned long long)scfq_device_bytes_high_water());
    r.extra += js;
    // where the process's time went up to this row: [name, ms since the library was loaded] (include/sc_fqcount_debug.h)
    scfq_debug_stage_mark("sc: row computed");
    // (with --jobs another session's thread may add a mark between the sizing call and the filling one: ask again until the
    // text fits, and keep the row valid JSON whatever happens)
    std::string stages;
    for (int attempt = 0; attempt < 8; ++attempt) {
      stages.assign((size_t)scfq_debug_stages(nullptr, 0) + 256, '\0');
      const uint64_t need = scfq_debug_stages(&stages[0], stages.size());
      if (need < stages.size()) { stages.resize(std::strlen(stages.c_str())); break; }
      stages.clear();
    }
    if (stages.empty()) stages = "[]";
    r.extra += stages + "}\n";
  }
  return r;
}

// ... and its output part: echo the row (fq_count.nim:53) or quit_error
static void fq_count_emit(const FileResult& r) {
  if (r.exit_code) quit_error(r.error, r.exit_code);
  std::printf("%s\n", r.row.c_str());
  if (std::fflush(stdout) != 0) { /* EPIPE is swallowed like sc.nim:304 */ }
  if (!r.extra.empty()) std::fputs(r.extra.c_str(), stderr);
}

// Nim 1.0.6 `$float` as used by stderr.writeLine (src/fq_dedup.nim:80): "%.16g", ".0" appended when the text has neither
// '.' nor a letter, NaN -> "nan"
static std::string nim_float(double v) {
  if (v != v) return "nan";
  char f[64];
  std::snprintf(f, sizeof f, "%.16g", v);
  std::string t(f);
  if (t.find_first_of(".abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ") == std::string::npos) t += ".0";
  return t;
}

// command "fq-dedup" (sc.nim:118-122): one positional FASTQ, no options; proc fq_dedup (src/fq_dedup.nim:14-84)
static int cmd_fq_dedup(const std::vector<std::string>& params) {
  auto help = [](FILE* f) {
    std::fputs("Removes exact duplicates from FASTQ Files\n\nUsage:\n  fq-dedup [options] fastq\n\nArguments:\n"
               "  fastq            Input FASTQ\n\nOptions:\n  -h, --help                 Show this help\n", f);
  };
  std::vector<std::string> files;
  for (size_t i = 1; i < params.size(); ++i) {
    if (params[i] == "-h" || params[i] == "--help") { help(stdout); return 0; }
    files.push_back(params[i]);
  }
  if (files.size() != 1) { help(stdout); quit_error(files.empty() ? "Missing argument: fastq" : "Unknown argument(s): " + files[1], 1); }
  const std::string& fastq = files[0];
  if (fastq == "STDIN") quit_error("This command does not support stdin");                    // sc.nim:58-60
  struct stat sb;
  if (stat(fastq.c_str(), &sb) != 0 || S_ISDIR(sb.st_mode))                                    // helpers.nim:39-43
    quit_error(fastq + " does not exist or is not readable");
  scfq_dedup_stats st;
  std::memset(&st, 0, sizeof st);
  st.struct_size = sizeof st;
  // the reference announces "No Duplicates Found" before pass 2 starts echoing (fq_dedup.nim:51-53); here the FASTQ goes
  // to fd 1 first and all diagnostics follow: stdout is byte-identical and the lines on stderr keep their order
  std::fflush(stdout);
  const int rc = scfq_dedup_file(fastq.c_str(), nullptr, 1, &st);
  if (rc == SCFQ_EOPEN) quit_error("Unable to open file: " + fastq, 2);                         // fq_dedup.nim:37-38
  if (rc == SCFQ_EPIPE) return 0;              // `errno: 32 Broken pipe` is swallowed (sc.nim:304); any other IOError exits 1 (sc.nim:299-305)
  if (rc != SCFQ_OK) quit_error(std::string(scfq_strerror(rc)) + ": " + scfq_dedup_error_detail() + scfq_last_error_detail(), 1);
  if (st.duplicates == 0) std::fputs("No Duplicates Found\nCopying fq to stdout\n", stderr);   // :51-53 (check.len == 0)
  std::fprintf(stderr, "total_reads: %llu\n", (unsigned long long)st.total_reads);            // :74
  std::fprintf(stderr, "duplicates %llu\n", (unsigned long long)st.duplicates);                // :75
  std::fprintf(stderr, "false-positive: %llu\n", (unsigned long long)st.false_positive);      // :79
  std::fprintf(stderr, "false-positive-rate: %s\n", nim_float((double)st.false_positive / (double)st.duplicates).c_str());   // :80
  return 0;
}

// command "fq-meta" (sc.nim:67-79): -n/--lines (default 100), -t/--header, -b/--basename, -a/--absolute, [fastq ...]
static int cmd_fq_meta(const std::vector<std::string>& params) {
  auto help = [](FILE* f) {
    std::fputs("Output metadata for FASTQ\n\nUsage:\n  fq-meta [options] [fastq ...]\n\nArguments:\n  [fastq ...]      List of FASTQ files\n\n"
               "Options:\n  -n, --lines=LINES          Number of sequences to sample (n_lines) for qual and index/barcode determination (default: 100)\n"
               "  -t, --header               Output the header\n  -b, --basename             Add basename column\n"
               "  -a, --absolute             Add column for absolute path\n  -h, --help                 Show this help\n"
               "\nMI355X option (addition):\n      --whole-file           Quality range from the histogram of every quality line (device scan)\n", f);
  };
  bool header = false, basename = false, absolute = false;
  uint32_t flags = 0;
  std::string lines = "100";
  std::vector<std::string> files;
  for (size_t i = 1; i < params.size(); ++i) {
    const std::string& a = params[i];
    if (a == "-h" || a == "--help") { help(stdout); return 0; }
    else if (a == "-t" || a == "--header") header = true;
    else if (a == "-b" || a == "--basename") basename = true;
    else if (a == "-a" || a == "--absolute") absolute = true;
    else if (a == "--whole-file") flags |= SCFQ_META_WHOLE_FILE;
    else if ((a == "-n" || a == "--lines") && i + 1 < params.size()) lines = params[++i];
    else if (a.rfind("--lines=", 0) == 0) lines = a.substr(8);
    else if (a.rfind("-n", 0) == 0 && a.size() > 2 && a[1] == 'n') lines = a.substr(a[2] == '=' ? 3 : 2);
    else if (a.size() > 1 && a[0] == '-' && a != "-") { help(stdout); quit_error("Error: Unknown option: " + a, 1); }
    else files.push_back(a);
  }
  char* endp = nullptr;
  const long n = std::strtol(lines.c_str(), &endp, 10);
  if (lines.empty() || *endp || n < 0) quit_error("invalid integer: " + lines, 1);      // parseInt raises ValueError (sc.nim:79)
  if (header) std::printf("%s\n", output_header(scfq_meta_header(), basename, absolute).c_str());   // sc.nim:75-76
  for (const auto& fastq : files) {                                                          // sc.nim:77-79
    char row[4096];
    const int rc = scfq_meta_file_tsv(fastq.c_str(), (uint32_t)n, flags, row, sizeof row);
    if (rc == SCFQ_EOPEN) quit_error("Unable to open file: " + fastq, 2);                   // fq_meta.nim:223-224
    if (rc == SCFQ_EARG) quit_error("index out of bounds", 1);                              // extract_read_info IndexError -> sc.nim:299-305
    if (rc < 0) quit_error(std::string(scfq_strerror(rc)) + ": " + scfq_last_error_detail(), 1);
    std::printf("%s\n", output_w_fnames(row, fastq, basename, absolute).c_str());
  }
  std::fflush(stdout);
  return 0;
}

int main(int argc, char** argv) {
  scfq_debug_stage_mark("sc: main entered");
  signal(SIGPIPE, SIG_IGN);   // sc.nim:45-46
  std::vector<std::string> params(argv + 1, argv + argc);
  if (stdin_is_fifo())        // sc.nim:274-284
    for (auto& p : params)
      if (p == "-") { p = "STDIN"; break; }
  if (params.empty() || params[0] == "-h" || params[0] == "--help") { help_top(stdout); return 0; }
  if (params[0] == "-v" || params[0] == "--version") { std::printf("%s\n", kVersion); return 0; }
  if (params[0] == "fq-dedup") return cmd_fq_dedup(params);
  if (params[0] == "fq-meta") return cmd_fq_meta(params);
  if (params[0] != "fq-count") {
    help_top(stdout);
    quit_error("Unknown command: " + params[0] + " (this build provides the FASTQ commands only)", 1);
  }
  if (params.size() == 1) { help_fq_count(stdout); return 0; }   // sc.nim:288-290: len <= 1 -> "-h"

  bool header = false, basename = false, absolute = false, stats = false;
  int jobs = 0;      // 0: not given
  int shard_rank = -1, shard_world = 0, transport = SCFQ_COMM_RCCL;
  std::string rendezvous;
  std::vector<std::string> files;
  std::vector<int32_t> devices;
  uint32_t flags = 0;
  bool only_positional = false;
  for (size_t i = 1; i < params.size(); ++i) {
    const std::string& a = params[i];
    if (only_positional || a.empty() || a[0] != '-' || a == "-") { files.push_back(a); continue; }
    if (a == "--") { only_positional = true; continue; }
    if (a == "-h" || a == "--help") { help_fq_count(stdout); return 0; }
    if (a == "--header") header = true;
    else if (a == "--basename") basename = true;
    else if (a == "--absolute") absolute = true;
    else if (a == "--debug") {}
    else if (a == "--struct-check") flags |= SCFQ_STRUCT_CHECK;
    else if (a == "--qual-hist") flags |= SCFQ_QUAL_HIST;
    else if (a == "--stats") { stats = true; flags |= SCFQ_TIMING; }
    else if (a.rfind("--jobs=", 0) == 0) jobs = std::max(1, std::atoi(a.substr(7).c_str()));
    else if (a.rfind("--shard-rank=", 0) == 0) shard_rank = std::atoi(a.substr(13).c_str());
    else if (a.rfind("--shard-world=", 0) == 0) shard_world = std::atoi(a.substr(14).c_str());
    else if (a.rfind("--rendezvous=", 0) == 0) rendezvous = a.substr(13);
    else if (a == "--transport=tcp") transport = SCFQ_COMM_TCP;
    else if (a == "--transport=rccl") transport = SCFQ_COMM_RCCL;
    else if (a.rfind("--devices=", 0) == 0) {
      const std::string list = a.substr(10);
      size_t p = 0;
      while (p < list.size()) {
        size_t q = list.find(',', p);
        if (q == std::string::npos) q = list.size();
        devices.push_back(std::atoi(list.substr(p, q - p).c_str()));
        p = q + 1;
      }
    } else if (a.size() >= 2 && a[1] != '-') {
      for (size_t k = 1; k < a.size(); ++k) {   // combined short flags: -tb
        if (a[k] == 't') header = true;
        else if (a[k] == 'b') basename = true;
        else if (a[k] == 'a') absolute = true;
        else if (a[k] == 'h') { help_fq_count(stdout); return 0; }
        else { help_fq_count(stdout); quit_error(std::string("Error: Unknown option: -") + a[k], 1); }
      }
    } else {
      help_fq_count(stdout);
      quit_error("Error: Unknown option: " + a, 1);   // UsageError funnel, sc.nim:294-298
    }
  }

  scfq_opts opts;
  std::memset(&opts, 0, sizeof opts);
  opts.struct_size = sizeof opts;
  opts.flags = flags;
  opts.n_devices = (int32_t)devices.size();
  opts.device_ids = devices.empty() ? nullptr : devices.data();
  if (const char* e = std::getenv("SC_GPU_CHUNK")) opts.chunk_bytes = std::strtoull(e, nullptr, 10);

  // ---- one process per GPU (addition): byte-range shard per rank, exchange inside the library, rank 0 prints ----------
  if (shard_world > 0 || shard_rank >= 0) {
    if (shard_world <= 0 && std::getenv("WORLD_SIZE")) shard_world = std::atoi(std::getenv("WORLD_SIZE"));
    if (shard_rank < 0 && std::getenv("RANK")) shard_rank = std::atoi(std::getenv("RANK"));
    if (rendezvous.empty() && std::getenv("MASTER_PORT"))
      rendezvous = std::string(std::getenv("MASTER_ADDR") ? std::getenv("MASTER_ADDR") : "127.0.0.1") + ":" + std::getenv("MASTER_PORT");
    const size_t colon = rendezvous.rfind(':');
    if (shard_world < 1 || shard_rank < 0 || shard_rank >= shard_world || colon == std::string::npos)
      quit_error("--shard-rank / --shard-world / --rendezvous=HOST:PORT do not describe a rank", 1);
    const std::string host = rendezvous.substr(0, colon);
    const int port = std::atoi(rendezvous.substr(colon + 1).c_str());
    if (devices.empty()) devices.push_back(shard_rank);
    opts.n_devices = 1;
    opts.device_ids = devices.data();
    scfq_comm* comm = nullptr;
    const int rc = scfq_comm_init_rendezvous(host.c_str(), port, shard_world, shard_rank, devices[0], transport, 0, &comm);
    if (rc) quit_error(std::string(scfq_strerror(rc)) + ": " + scfq_comm_error_detail(), 1);
    if (header && shard_rank == 0) std::printf("%s\n", output_header(kHeader, basename, absolute).c_str());
    else if (!header && files.empty()) quit_error("No FASTQ specified", 3);
    for (const auto& f : files) {
      const FileResult r = fq_count_compute(f, basename, absolute, opts, stats, comm);
      if (r.exit_code || shard_rank == 0) fq_count_emit(r);       // every rank quits with the reference's message on an error
      else if (!r.extra.empty()) std::fputs(r.extra.c_str(), stderr);      // (the additions' stderr lines, e.g. --stats of this rank's share)
    }
    scfq_comm_destroy(comm);
    scfq_shutdown();
    return 0;
  }

  if (header) std::printf("%s\n", output_header(kHeader, basename, absolute).c_str());   // sc.nim:110-111
  else if (files.empty()) quit_error("No FASTQ specified", 3);                           // sc.nim:112-113
  // several devices in one process: their RCCL communicators come up BEFORE the first row (while one is created the library
  // points descriptor 1 at descriptor 2 — RCCL prints a banner on stdout — and a row printed in that window would be lost to
  // stderr); a failure here is reported by the first counting call
  if (devices.size() > 1 && !files.empty()) (void)scfq_prepare(&opts);
  // sc.nim:114-116 takes the files one after the other.  Here two are in flight unless --jobs says otherwise (a file's
  // pipeline — read or copy, inflate, scan — leaves the device idle most of the time; file k + 1's ingest runs under file k's
  // kernels): rows still come out in argv order and the first failing file ends the run where the sequential loop would have.
  if (jobs == 0) jobs = devices.size() > 1 ? 1 : (int)std::min<size_t>(2, std::max<size_t>(1, files.size()));
  if (jobs <= 1 || files.size() <= 1) {
    for (const auto& f : files) fq_count_emit(fq_count_compute(f, basename, absolute, opts, stats));   // sc.nim:114-116
  } else {
    // --jobs=N: up to N files in flight (each a session on its own device context: inflate / read of one file overlaps
    // the scans of the others); rows are still emitted strictly in argv order, and the first failing file ends the
    // run with the reference's message and exit code exactly where the sequential loop would have.
    std::vector<FileResult> results(files.size());
    std::vector<char> done(files.size(), 0);
    std::mutex mu;
    std::condition_variable cv;
    std::atomic<size_t> next{0};
    std::vector<std::thread> pool;
    const size_t nthreads = std::min<size_t>((size_t)jobs, files.size());
    for (size_t t = 0; t < nthreads; ++t)
      pool.emplace_back([&] {
        for (;;) {
          const size_t i = next.fetch_add(1);
          if (i >= files.size()) return;
          FileResult r = fq_count_compute(files[i], basename, absolute, opts, stats);
          { std::lock_guard<std::mutex> lk(mu); results[i] = std::move(r); done[i] = 1; }
          cv.notify_all();
        }
      });
    for (size_t i = 0; i < files.size(); ++i) {
      { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return done[i] != 0; }); }
      if (results[i].exit_code) {   // stop handing out files, let the in-flight ones finish, then quit like the sequential loop
        next.store(files.size());
        for (auto& th : pool) th.join();
        scfq_shutdown();
      }
      fq_count_emit(results[i]);
    }
    for (auto& th : pool) if (th.joinable()) th.join();
  }
  // A process that is about to end does not give its device memory back piece by piece (scfq_shutdown: 40 - 50 ms of frees and
  // stream destruction after a 10 GB .gz): the rows are out, the driver reclaims everything with the process.  SC_CLEAN_EXIT=1 keeps
  // the orderly shutdown (leak checkers).
  std::fflush(stdout);
  std::fflush(stderr);
  if (std::getenv("SC_CLEAN_EXIT")) { scfq_shutdown(); return 0; }
  _exit(0);
}
