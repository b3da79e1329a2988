// scfq_host.cpp — device-independent part of the C ABI (include/sc_fqcount.h): the partial monoid
// on the host, finalisation into the reference's counters, and the reference's output formatting.
//
// Reference anchors:
//   src/fq_count.nim:39-41   reads = #{i : i mod 4 == 1}  == ceil(lines/4)
//   src/fq_count.nim:42-45   sequence line is i mod 4 == 2  == relative class r = 1 from phase 0
//   src/fq_count.nim:47-51   output fields and `$` formatting (Nim 1.0.6 `$float` = "%.16g" + ".0" rule)
#include "../../include/sc_fqcount.h"

#include <cstdio>
#include <cstring>

extern "C" {

void scfq_partial_identity(scfq_partial* p, uint64_t* hist) {
  if (p) std::memset(p, 0, sizeof(*p));
  if (hist) std::memset(hist, 0, SCFQ_HIST_WORDS * sizeof(uint64_t));
}

static inline void rot_add(uint64_t* acc, const uint64_t* b, unsigned k) {
  uint64_t t[4];
  for (unsigned r = 0; r < 4; ++r) t[r] = acc[r] + b[(r - k) & 3u];
  for (unsigned r = 0; r < 4; ++r) acc[r] = t[r];
}

int scfq_partial_combine(scfq_partial* acc, const scfq_partial* b, uint64_t* hist_acc, const uint64_t* hist_b) {
  if (!acc || !b) return SCFQ_EARG;
  const unsigned k = (unsigned)(acc->nl & 3u);
  rot_add(acc->gc, b->gc, k);
  rot_add(acc->n, b->n, k);
  rot_add(acc->len, b->len, k);
  rot_add(acc->starts, b->starts, k);
  rot_add(acc->first_at, b->first_at, k);
  rot_add(acc->first_plus, b->first_plus, k);
  {
    // which class histograms stay complete: b's class c lands on class (c + k) & 3 of the result
    const uint64_t cb = (b->hist_class >= 1 && b->hist_class <= 4) ? (((b->hist_class - 1) + k) & 3u) + 1 : b->hist_class;
    if (acc->hist_class == 0) acc->hist_class = cb;
    else if (cb != 0 && cb != acc->hist_class) acc->hist_class = 5;
  }
  if (hist_acc && hist_b) {
    for (unsigned c = 0; c < 4; ++c) {
      const uint64_t* src = hist_b + ((c - k) & 3u) * 256;
      uint64_t* dst = hist_acc + c * 256;
      for (unsigned v = 0; v < 256; ++v) dst[v] += src[v];
    }
  }
  acc->reserved[0] |= b->reserved[0];   // status word of a sharded count: non-zero when a contributing shard failed
  acc->nl += b->nl;
  if (b->bytes) acc->last_byte = b->last_byte;
  acc->bytes += b->bytes;
  return SCFQ_OK;
}

int scfq_partial_finalize(const scfq_partial* p, const uint64_t* hist, scfq_counts* out) {
  if (!p || !out) return SCFQ_EARG;
  if (out->struct_size != sizeof(scfq_counts)) return SCFQ_EARG;
  std::memset(out, 0, sizeof(*out));
  out->struct_size = sizeof(scfq_counts);
  out->abi_version = SCFQ_ABI_VERSION;
  // relative class 1 from phase 0 is the reference's `i mod 4 == 2`
  out->gc_bases = p->gc[1];
  out->n_bases = p->n[1];
  out->bases = p->len[1];
  out->newlines = p->nl;
  out->input_bytes = p->bytes;
  // a final line without '\n' still counts (Nim readLine returns it)
  out->lines = p->nl + ((p->bytes > 0 && p->last_byte != (uint64_t)'\n') ? 1u : 0u);
  out->reads = (out->lines + 3) / 4;
  out->bad_at = p->starts[0] - p->first_at[0];
  out->bad_plus = p->starts[2] - p->first_plus[2];
  if (hist) {
    if (p->hist_class != 0 && p->hist_class != 4) return SCFQ_ESPEC;   // the speculative form completed another class
    std::memcpy(out->qual_hist, hist + 3 * 256, 256 * sizeof(uint64_t));   // class 3 == `i mod 4 == 0`
  }
  return SCFQ_OK;
}

// Nim 1.0.6 system/formatfloat.nim writeFloatToBuffer: sprintf("%.16g"), ',' -> '.', append ".0"
// when no '.' and no letter, collapse any NaN spelling to "nan" and infinities to "inf"/"-inf".
static int nim_float_to_str(double v, char* f, size_t cap) {
  int m = std::snprintf(f, cap, "%.16g", v);
  bool has_dot = false;
  for (int k = 0; k < m; ++k) {
    if (f[k] == ',') { f[k] = '.'; has_dot = true; }
    else if ((f[k] >= 'a' && f[k] <= 'z') || (f[k] >= 'A' && f[k] <= 'Z') || f[k] == '.') has_dot = true;
  }
  if (!has_dot) { f[m] = '.'; f[m + 1] = '0'; f[m + 2] = 0; m += 2; }
  if (m > 0 && (f[m - 1] == 'n' || f[m - 1] == 'N')) { std::strcpy(f, "nan"); m = 3; }
  else if (m > 0 && (f[m - 1] == 'f' || f[m - 1] == 'F')) { std::strcpy(f, f[0] == '-' ? "-inf" : "inf"); m = (int)std::strlen(f); }
  return m;
}

int scfq_format_tsv(const scfq_counts* c, char* buf, uint64_t cap) {
  if (!c) return SCFQ_EARG;
  // $(gc_cnt.float / (total_len - n_cnt).float)      src/fq_count.nim:48 (int64 -> float64)
  const double v = (double)(int64_t)c->gc_bases / (double)(int64_t)(c->bases - c->n_bases);
  char f[96];
  nim_float_to_str(v, f, sizeof f);
  char tmp[256];
  const int m = std::snprintf(tmp, sizeof tmp, "%llu\t%s\t%llu\t%llu\t%llu", (unsigned long long)c->reads, f,
                              (unsigned long long)c->gc_bases, (unsigned long long)c->n_bases,
                              (unsigned long long)c->bases);
  if (buf && cap) {
    const uint64_t ncopy = ((uint64_t)m < cap - 1) ? (uint64_t)m : cap - 1;
    std::memcpy(buf, tmp, ncopy);
    buf[ncopy] = 0;
  }
  return m;
}

const char* scfq_strerror(int rc) {
  switch (rc) {
    case SCFQ_OK: return "ok";
    case SCFQ_EOPEN: return "unable to open file";
    case SCFQ_EGZ: return "gzip stream error";
    case SCFQ_EHIP: return "HIP runtime error (is a gfx950 GPU visible?)";
    case SCFQ_ERCCL: return "collective exchange error";
    case SCFQ_EARG: return "invalid argument";
    case SCFQ_EIO: return "read / write error";
    case SCFQ_EPIPE: return "broken pipe on the output descriptor";
    case SCFQ_ENOMEM: return "out of host memory";
    case SCFQ_ESPEC: return "speculative quality histogram did not verify across shards (retry with SCFQ_HIST_EXACT)";
    default: return "unknown error";
  }
}

}  // extern "C"
