// scfq_meta.cpp — `sc fq-meta` (next row of SURVEY.md §8f-2): reference src/fq_meta.nim:10-278, CLI sc.nim:67-79.
//
// fq-meta looks at the first `sample_n` records only (fq_meta.nim:226: i < sample_n * 4): at most a few hundred lines, pure
// host-side string work, so this is host code by nature — what the device adds is optional: with
// SCFQ_META_WHOLE_FILE the quality range (min_qual / max_qual and the format guess built on it) comes from K3's
// quality-line histogram of the WHOLE file instead of the sampled records (an addition; the default is the reference's
// sampled semantics).
//
// Restated from the reference, with the third-party pieces it leans on:
//   * line iterator: Nim 1.0.6 readLine ('\n' ends a line, a '\r' directly before it is dropped)
//   * strutils.split(set of separators): every separator char splits, empty fields are kept
//   * strutils.strip(chars = {'@'}): leading and trailing '@' removed
//   * nim-regex `find` = unanchored search, `match` = whole string.  Every pattern of fq_meta.nim:47-92 is a fixed-length
//     sequence of literals and character classes with optional ^ / $ anchors, so it is matched here as a list of byte
//     classes; the barcode pattern re"[ATCGN\+\-]{3,12}+" (:208) is read as "3 or more bytes of that set" (the trailing
//     `+` repeats the bounded group) — parity unpinned beyond the fixtures, which all carry 6-8 byte indices.
//   * CountTable.largest (:258): ties between equally frequent barcodes resolve in hash-slot order in the reference;
//     here the first one seen wins (unpinned, no fixture has a tie).
#include "../../include/sc_fqcount.h"

#include <zlib.h>

#include <algorithm>
#include <bitset>
#include <cctype>
#include <cstdio>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

namespace {

using Strs = std::vector<std::string>;

// ---- the tiny pattern language of fq_meta.nim:47-92 ------------------------------------------------------------------
struct Pattern {
  bool anchor_start = false, anchor_end = false;
  std::vector<std::bitset<256>> pos;
  explicit Pattern(const char* re) {
    const char* p = re;
    if (*p == '^') { anchor_start = true; ++p; }
    while (*p) {
      if (*p == '$' && !p[1]) { anchor_end = true; break; }
      std::bitset<256> cls;
      if (*p == '[') {
        ++p;
        while (*p && *p != ']') {
          if (p[1] == '-' && p[2] && p[2] != ']') { for (int c = (unsigned char)p[0]; c <= (unsigned char)p[2]; ++c) cls.set((size_t)c); p += 3; }
          else { cls.set((unsigned char)*p); ++p; }
        }
        if (*p == ']') ++p;
      } else {
        cls.set((unsigned char)*p);
        ++p;
      }
      int rep = 1;
      if (*p == '{') { rep = std::atoi(p + 1); while (*p && *p != '}') ++p; if (*p == '}') ++p; }
      for (int k = 0; k < rep; ++k) pos.push_back(cls);
    }
  }
  bool found_in(const std::string& s) const {     // regex.find: unanchored search honouring ^ and $
    const size_t m = pos.size();
    if (s.size() < m) return false;
    const size_t lo = anchor_end ? s.size() - m : 0, hi = anchor_start ? 0 : s.size() - m;
    for (size_t st = lo; st <= hi; ++st) {
      bool ok = true;
      for (size_t k = 0; k < m && ok; ++k) ok = pos[k].test((unsigned char)s[st + k]);
      if (ok) return true;
      if (st == hi) break;
    }
    return false;
  }
};

struct Instrument { Pattern pattern; Strs sequencer; };
struct Flowcell { Pattern pattern; Strs sequencer; std::string description; };

const std::vector<Instrument>& instrument_ids() {      // fq_meta.nim:47-60
  static const std::vector<Instrument> v = {
      {Pattern("HWI-M[0-9]{4}$"), {"MiSeq"}},        {Pattern("HWUSI"), {"GenomeAnalyzerIIx"}},
      {Pattern("M[0-9]{5}$"), {"MiSeq"}},            {Pattern("A[0-9]{5}$"), {"NovaSeq"}},
      {Pattern("HWI-C[0-9]{5}$"), {"HiSeq1500"}},    {Pattern("C[0-9]{5}$"), {"HiSeq1500"}},
      {Pattern("HWI-D[0-9]{5}$"), {"HiSeq2500"}},    {Pattern("D[0-9]{5}$"), {"HiSeq2500"}},
      {Pattern("J[0-9]{5}$"), {"HiSeq3000"}},        {Pattern("K[0-9]{5}$"), {"HiSeq3000", "HiSeq4000"}},
      {Pattern("E[0-9]{5}$"), {"HiSeqX"}},           {Pattern("NB[0-9]{6}$"), {"NextSeq"}},
      {Pattern("NS[0-9]{6}$"), {"NextSeq"}},         {Pattern("MN[0-9]{5}$"), {"MiniSeq"}}};
  return v;
}

const std::vector<Flowcell>& flowcell_ids() {           // fq_meta.nim:70-92
  static const std::vector<Flowcell> v = {
      {Pattern("AAXX$"), {"GenomeAnalyzer"}, ""},
      {Pattern("C[A-Z,0-9]{4}ANXX$"), {"HiSeq1500", "HiSeq2000", "HiSeq2500"}, "High Output (8-lane) v4 flow cell"},
      {Pattern("C[A-Z,0-9]{4}ACXX$"), {"HiSeq1000", "HiSeq1500", "HiSeq2000", "HiSeq2500"}, "High Output (8-lane) v3 flow cell"},
      {Pattern("H[A-Z,0-9]{4}ADXX$"), {"HiSeq1500", "HiSeq2500"}, "Rapid Run (2-lane) v1 flow cell"},
      {Pattern("H[A-Z,0-9]{4}BCXX$"), {"HiSeq1500", "HiSeq2500"}, "Rapid Run (2-lane) v2 flow cell"},
      {Pattern("H[A-Z,0-9]{4}BCXY$"), {"HiSeq1500", "HiSeq2500"}, "Rapid Run (2-lane) v2 flow cell"},
      {Pattern("H[A-Z,0-9]{4}BBXX$"), {"HiSeq4000"}, "(8-lane) v1 flow cell"},
      {Pattern("H[A-Z,0-9]{4}BBXY$"), {"HiSeq4000"}, "(8-lane) v1 flow cell"},
      {Pattern("H[A-Z,0-9]{4}CCXX$"), {"HiSeqX"}, "(8-lane) flow cell"},
      {Pattern("H[A-Z,0-9]{4}CCXY$"), {"HiSeqX"}, "(8-lane) flow cell"},
      {Pattern("H[A-Z,0-9]{4}ALXX$"), {"HiSeqX"}, "(8-lane) flow cell"},
      {Pattern("H[A-Z,0-9]{4}AGXX$"), {"NextSeq"}, "High output flow cell"},
      {Pattern("H[A-Z,0-9]{4}BGXX$"), {"NextSeq"}, "High output flow cell"},
      {Pattern("H[A-Z,0-9]{4}BGXY$"), {"NextSeq"}, "High output flow cell"},
      {Pattern("H[A-Z,0-9]{4}BGX2$"), {"NextSeq"}, "High output flow cell"},
      {Pattern("H[A-Z,0-9]{4}AFXX$"), {"NextSeq"}, "Mid output flow cell"},
      {Pattern("H[A-Z,0-9]{4}DMXX$"), {"NovaSeq"}, "S2 flow cell"},
      {Pattern("H[A-Z,0-9]{4}DSXX$"), {"NovaSeq"}, "S2 flow cell"},
      {Pattern("^A[A-Z,0-9]{4}$"), {"MiSeq"}, "MiSeq flow cell"},
      {Pattern("^B[A-Z,0-9]{4}$"), {"MiSeq"}, "MiSeq flow cell"},
      {Pattern("^D[A-Z,0-9]{4}$"), {"MiSeq"}, "MiSeq nano flow cell"},
      {Pattern("^G[A-Z,0-9]{4}$"), {"MiSeq"}, "MiSeq micro flow cell"}};
  return v;
}

struct FastqType { const char* name; const char* phred; int minimum, maximum; };
const FastqType kTypes[] = {   // fq_meta.nim:35-39
    {"Sanger", "Phred+33", 0, 40},         {"Solexa", "Solexa+64", 59, 104},     {"Illumina 1.3+", "Phred+64", 64, 104},
    {"Illumina 1.5+", "Phred+64", 64, 104}, {"Illumina 1.8+", "Phred+33", 0, 42}};

// index into the reference's printable table (fq_meta.nim:10): '!' = 0 ... '~' = 93, anything else -1 (strutils.find)
int qual_to_int(unsigned char c) { return (c >= 33 && c <= 126) ? (int)c - 33 : -1; }

Strs split_set(const std::string& s, const char* seps) {     // strutils.split(s, set[char]): empty fields kept
  Strs out;
  std::string cur;
  for (char c : s) {
    if (std::strchr(seps, c) && c) { out.push_back(cur); cur.clear(); }
    else cur += c;
  }
  out.push_back(cur);
  return out;
}

std::string strip_at(const std::string& s) {                  // strip(chars = {'@'})
  size_t a = 0, b = s.size();
  while (a < b && s[a] == '@') ++a;
  while (b > a && s[b - 1] == '@') --b;
  return s.substr(a, b - a);
}

void dedup_keep_first(Strs& v) {                              // sequtils.deduplicate
  Strs out;
  for (const auto& x : v)
    if (std::find(out.begin(), out.end(), x) == out.end()) out.push_back(x);
  v.swap(out);
}

bool contains(const Strs& v, const char* x) { return std::find(v.begin(), v.end(), x) != v.end(); }

// fq_meta.nim:120-152
void detect_sequencer(const std::string& machine, const std::string& flowcell, Strs& sequencers, std::string& prob, std::string& fc_desc) {
  Strs by_iid, by_fcid;
  std::string desc;
  for (const auto& k : instrument_ids())
    if (k.pattern.found_in(machine)) by_iid.insert(by_iid.end(), k.sequencer.begin(), k.sequencer.end());
  for (const auto& k : flowcell_ids())
    if (k.pattern.found_in(flowcell)) { desc = k.description; by_fcid.insert(by_fcid.end(), k.sequencer.begin(), k.sequencer.end()); }
  sequencers.clear(); prob.clear(); fc_desc.clear();
  if (by_iid.empty() && by_fcid.empty()) return;
  if (by_iid.empty()) { sequencers = by_fcid; prob = "likely:flowcell"; fc_desc = desc; return; }
  if (by_fcid.empty()) { sequencers = by_iid; prob = "likely:machine"; fc_desc = desc; return; }
  Strs inter;
  for (const auto& i : by_iid)
    for (const auto& j : by_fcid)
      if (i == j) inter.push_back(i);
  dedup_keep_first(inter);
  if (!inter.empty()) { sequencers = inter; prob = "high:machine+flowcell"; fc_desc = desc; return; }
  sequencers = by_iid;
  sequencers.insert(sequencers.end(), by_fcid.begin(), by_fcid.end());
  dedup_keep_first(sequencers);
  prob = "uncertain";
}

std::string sequencer_name(const Strs& s) {                   // fq_meta.nim:180-194
  if (contains(s, "HiSeq2000") || contains(s, "HiSeq2500")) return "HiSeq2000/2500";
  if (contains(s, "HiSeq1500") || contains(s, "HiSeq2500")) return "HiSeq1500/2500";
  if (contains(s, "HiSeq3000") || contains(s, "HiSeq4000")) return "HiSeq3000/4000";
  return s.empty() ? "" : s.back();
}

bool is_barcode(const std::string& b) {                       // regex.match(barcode, re"[ATCGN\+\-]{3,12}+")
  if (b.size() < 3) return false;
  for (char c : b)
    if (!std::strchr("ATCGN+-", c) || !c) return false;
  return true;
}

// line reader over plain files and gzip (zlib gzread; fq_meta.nim:218-222 picks by a case-INSENSITIVE ".gz" suffix)
struct LineReader {
  gzFile gz = nullptr;
  FILE* f = nullptr;
  std::vector<unsigned char> buf;
  size_t pos = 0, len = 0;
  bool eof = false;
  bool open(const char* path) {
    const size_t n = std::strlen(path);
    const bool is_gz = n >= 3 && path[n - 3] == '.' && std::tolower((unsigned char)path[n - 2]) == 'g' && std::tolower((unsigned char)path[n - 1]) == 'z';
    buf.resize(1 << 16);
    if (is_gz) gz = gzopen(path, "rb"); else f = std::fopen(path, "rb");
    return gz || f;
  }
  ~LineReader() { if (gz) gzclose(gz); if (f) std::fclose(f); }
  bool fill() {
    if (eof) return false;
    int r = gz ? gzread(gz, buf.data(), (unsigned)buf.size()) : (int)std::fread(buf.data(), 1, buf.size(), f);
    if (r <= 0) { eof = true; return false; }
    pos = 0; len = (size_t)r;
    return true;
  }
  bool at_end() { return pos >= len && !fill(); }              // stream.atEnd()
  bool read_line(std::string& line) {                          // readLine: '\n' ends, "\r\n" -> '\r' dropped too
    line.clear();
    bool any = false;
    for (;;) {
      if (pos >= len && !fill()) return any;
      any = true;
      const unsigned char c = buf[pos++];
      if (c == '\n') { if (!line.empty() && line.back() == '\r') line.pop_back(); return true; }
      line += (char)c;
    }
  }
};

std::string join(const Strs& v, const char* sep) {
  std::string out;
  for (size_t k = 0; k < v.size(); ++k) { if (k) out += sep; out += v[k]; }
  return out;
}

}  // namespace

extern "C" {

const char* scfq_meta_header(void) {     // fq_meta.nim:11-26
  return "machine\tsequencer\tprob_sequencer\tflowcell\tflowcell_description\trun\tlane\tsequence_id\tindex1\tindex2\t"
         "qual_format\tqual_phred\tqual_multiple\tmin_qual\tmax_qual\tn_lines";
}

int scfq_meta_file_tsv(const char* path, uint32_t sample_n, uint32_t flags, char* out, uint64_t cap) {
  if (!path || (!out && cap)) return SCFQ_EARG;
  LineReader rd;
  if (!rd.open(path)) return SCFQ_EOPEN;                       // fq_meta.nim:223-224 -> quit_error(..., 2)
  std::string sequence_id, machine, run, lane, flowcell, line;
  int qual_min = -1, qual_max = -1;
  Strs barcodes;
  uint64_t i = 0;
  while (!rd.at_end() && i < (uint64_t)sample_n * 4) {          // :226
    if (!rd.read_line(line)) break;
    if (i % 4 == 0) {                                          // :229
      const Strs parts = split_set(line, ":/#");
      const bool slash = line.find('/') != std::string::npos;
      if (i == 0) {                                            // extract_read_info, :155-178
        if (parts.size() == 1) sequence_id = strip_at(parts[0]);
        else {
          machine = strip_at(parts[0]);
          if (slash) lane = parts[1];                          // @HWUSI-EAS100R:6:73:941:1973#ATGGGC/1
          else {
            if (parts.size() < 4) return SCFQ_EARG;            // qual_line[2] / [3]: IndexError in the reference (exit 1)
            run = parts[1];
            flowcell = parts[2];
            const size_t us = flowcell.rfind('_');
            if (us != std::string::npos) flowcell = flowcell.substr(us + 1);
            lane = parts[3];
          }
        }
      }
      if (parts.size() > 2) {                                  // :233-239
        const std::string& bc = slash ? parts[parts.size() - 2] : parts.back();
        if (is_barcode(bc)) barcodes.push_back(bc);
      }
    }
    if (i % 4 == 3 && !line.empty()) {                         // qual_min_max, :97-102
      int lo = qual_min >= 0 ? qual_min : 1 << 30, hi = qual_min >= 0 ? qual_max : -(1 << 30);
      for (unsigned char c : line) { const int q = qual_to_int(c); lo = std::min(lo, q); hi = std::max(hi, q); }
      qual_min = lo; qual_max = hi;
    }
    ++i;
  }
  if (flags & SCFQ_META_WHOLE_FILE) {
    // addition: quality range over every quality line of the file, from the device histogram (K3)
    scfq_counts c;
    std::memset(&c, 0, sizeof c);
    c.struct_size = sizeof c;
    scfq_opts o;
    std::memset(&o, 0, sizeof o);
    o.struct_size = sizeof o;
    o.flags = SCFQ_QUAL_HIST;
    const int rc = scfq_count_file(path, &o, &c);
    if (rc) return rc;
    int lo = 1 << 30, hi = -(1 << 30);
    for (int b = 0; b < 256; ++b)
      if (c.qual_hist[b]) { const int q = qual_to_int((unsigned char)b); lo = std::min(lo, q); hi = std::max(hi, q); }
    if (hi >= lo) { qual_min = lo; qual_max = hi; }
  }
  Strs seqs;
  std::string sequencer, prob, fc_desc;
  if (!machine.empty() || !flowcell.empty()) {                 // :251-253
    detect_sequencer(machine, flowcell, seqs, prob, fc_desc);
    sequencer = sequencer_name(seqs);
  }
  Strs names, phreds;
  for (const auto& t : kTypes)                                 // :255
    if (qual_min >= t.minimum && qual_max <= t.maximum) { names.push_back(t.name); phreds.push_back(t.phred); }
  dedup_keep_first(phreds);
  std::string top_barcode;                                     // :256-258
  {
    std::vector<std::pair<std::string, int>> counts;
    for (const auto& b : barcodes) {
      auto it = std::find_if(counts.begin(), counts.end(), [&](const std::pair<std::string, int>& p) { return p.first == b; });
      if (it == counts.end()) counts.push_back({b, 1}); else ++it->second;
    }
    int best = 0;
    for (const auto& p : counts) if (p.second > best) { best = p.second; top_barcode = p.first; }
  }
  const Strs fields = {machine, sequencer, prob, flowcell, fc_desc, run, lane, sequence_id, top_barcode, "",
                       join(names, ";"), join(phreds, ";"), names.size() > 1 ? "true" : "false",
                       qual_min >= 0 ? std::to_string(qual_min) : "", qual_max >= 0 ? std::to_string(qual_max) : "",
                       std::to_string(i / 4)};                 // :262-277 ($(i/4).int)
  const std::string row = join(fields, "\t");
  if (out && cap) {
    const size_t ncopy = std::min<size_t>(row.size(), (size_t)cap - 1);
    std::memcpy(out, row.data(), ncopy);
    out[ncopy] = 0;
  }
  return (int)row.size();
}

}  // extern "C"
