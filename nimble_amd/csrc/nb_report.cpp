// nb_report.cpp — native `nimble report` reducer: per-read TSV -> feature<TAB>count<TAB>cell_barcode (host, product code).
//
// Same arithmetic and text as nimble_amd/report.py, which mirrors /root/reference/nimble/__main__.py:213-302 (report,
// check_df_from_input, convert_df_to_proper_umi) and /root/reference/nimble/utils.py:119-224 (per_umi_thresholding,
// umi_intersection, intersect_lists): float64 shares `score / len(features)` summed per feature with pandas' Kahan
// group_sum in row-then-feature order, plain left-to-right total, `share / total < threshold`, each round recomputed
// from the ORIGINAL rows minus that round's to_drop only (utils.py:146-173). Both implementations are held to the same
// golden fixtures (tests/golden/report_cases.json: outputs of the reference's own report()).
// Why native: the reference's per-UMI Python loop (utils.py:125-131,158-165) is the slow part of `report`
// (SURVEY.md §8f item 1); this streams the TSV once and reduces 10^8-row tables in minutes instead of hours.
// Compiled with -ffp-contract=off: no fused multiply-add may alter the double arithmetic.
#include <fcntl.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "nb_inflate.h"
#include "nb_internal.h"

namespace {

const char* const kNA[] = {"", "#N/A", "#N/A N/A", "#NA", "-1.#IND", "-1.#QNAN", "-NaN", "-nan", "1.#IND", "1.#QNAN", "<NA>", "N/A", "NA",
                           "NULL", "NaN", "None", "n/a", "nan", "null"};

bool is_na(const char* b, size_t n) {
  if (n > 8) return false;
  if (n) switch (b[0]) { case '#': case '-': case '1': case '<': case 'N': case 'n': break; default: return false; }
  for (const char* s : kNA) if (std::strlen(s) == n && std::memcmp(s, b, n) == 0) return true;
  return false;
}

// String interner without per-string heap objects: bytes live in one arena, lookups go through an open-addressing table
// of 64-bit hashes (the per-row cost of `report` is three of these lookups, so this is the hot structure of the parser).
struct Interner {
  std::vector<char> arena;
  std::vector<uint64_t> off;        // off[id] .. off[id + 1]
  std::vector<uint64_t> hashes;     // per id
  std::vector<uint32_t> slots;      // id + 1, 0 = empty
  uint64_t mask = 0;
  Interner() { off.push_back(0); slots.assign(1 << 12, 0); mask = slots.size() - 1; }
  static uint64_t hash(const char* b, size_t n) {
    uint64_t h = 0x9E3779B97F4A7C15ull ^ (n * 0xD6E8FEB86659FD93ull);
    while (n >= 8) { uint64_t w; std::memcpy(&w, b, 8); h = (h ^ w) * 0xD6E8FEB86659FD93ull; h ^= h >> 32; b += 8; n -= 8; }
    uint64_t w = 0; std::memcpy(&w, b, n); h = (h ^ w) * 0xD6E8FEB86659FD93ull; h ^= h >> 29;
    return h * 0x9E3779B97F4A7C15ull;
  }
  size_t size() const { return hashes.size(); }
  const char* ptr(uint32_t id) const { return arena.data() + off[id]; }
  size_t len(uint32_t id) const { return (size_t)(off[id + 1] - off[id]); }
  std::string str(uint32_t id) const { return std::string(ptr(id), len(id)); }
  void grow() {
    std::vector<uint32_t> ns(slots.size() * 2, 0); const uint64_t nm = ns.size() - 1;
    for (uint32_t id = 0; id < hashes.size(); ++id) { uint64_t p = (hashes[id] >> 20) & nm; while (ns[p]) p = (p + 1) & nm; ns[p] = id + 1; }
    slots.swap(ns); mask = nm;
  }
  uint32_t get(const char* b, size_t n) {
    const uint64_t h = hash(b, n);
    uint64_t p = (h >> 20) & mask;
    for (;;) {
      const uint32_t s = slots[p];
      if (!s) break;
      const uint32_t id = s - 1;
      if (hashes[id] == h && len(id) == n && std::memcmp(ptr(id), b, n) == 0) return id;
      p = (p + 1) & mask;
    }
    const uint32_t id = (uint32_t)hashes.size();
    slots[p] = id + 1; hashes.push_back(h);
    arena.insert(arena.end(), b, b + n); off.push_back(arena.size());
    if (hashes.size() * 2 > slots.size()) grow();
    return id;
  }
  uint32_t get(const std::string& s) { return get(s.data(), s.size()); }
  // rank[id] = position of the string in lexicographic (byte) order — Python's str ordering for the ASCII data at hand
  std::vector<uint32_t> ranks() const {
    std::vector<uint32_t> order(size());
    for (uint32_t i = 0; i < order.size(); ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
      const size_t lx = len(x), ly = len(y); const int c = std::memcmp(ptr(x), ptr(y), std::min(lx, ly));
      return c ? c < 0 : lx < ly; });
    std::vector<uint32_t> r(size());
    for (uint32_t i = 0; i < order.size(); ++i) r[order[i]] = i;
    return r;
  }
};

struct Rec { uint32_t cb; uint64_t umi; uint32_t feat; uint32_t seq; int64_t iscore; double fscore; };   // cb/feat: RANK of the string; umi: any key

struct Kahan {   // pandas group_sum (compensated), arrival order
  double s = 0.0, c = 0.0;
  void add(double v) {
    const double y = v - c, t = s + y;
    double cc = (t - s) - y;
    if (cc != cc || std::isinf(cc)) cc = 0.0;
    c = cc; s = t;
  }
};

bool parse_int_strict(const char* b, size_t n, int64_t* out) {
  if (n == 0) return false;
  size_t i = 0; bool neg = false;
  if (b[0] == '+' || b[0] == '-') { neg = b[0] == '-'; i = 1; if (n == 1) return false; }
  int64_t v = 0;
  for (; i < n; ++i) { if (b[i] < '0' || b[i] > '9') return false; v = v * 10 + (b[i] - '0'); }
  *out = neg ? -v : v; return true;
}

std::string quote_minimal(const std::string& f) {   // csv.QUOTE_MINIMAL with delimiter '\t', quotechar '"'
  if (f.find_first_of("\t\"\n\r") == std::string::npos) return f;
  std::string o = "\"";
  for (char c : f) { if (c == '"') o += '"'; o += c; }
  return o + "\"";
}


int reducer_threads() {
  int threads = (int)std::thread::hardware_concurrency();
  if (const char* e = std::getenv("NB_REPORT_THREADS")) threads = std::atoi(e);
  return std::max(1, std::min(threads, 32));
}

// The reduction proper, shared by the TSV and the array entry points. recs carry RANKS: cb / feat index cb_by_rank /
// feat_by_rank (strings in lexicographic order), umi is any key that identifies the UMI (only grouping matters: the
// output is ordered by (cell, feature) and integer scores do not depend on summation order; for float scores the TSV
// path passes the rank of the UMI string, which reproduces pandas' group order).
//
// Cost model: most UMIs are one row; the outcome of a one-row UMI depends only on (feature string, score), so those are
// memoised and the general thresholding loop runs for multi-row UMIs only. Results (surviving feature lists) are interned;
// counting uses that the rows arrive sorted by cell, so each cell's results are tallied in a small local vector.
int32_t reduce_records(std::vector<Rec>& recs, const std::vector<std::string>& cb_by_rank, const std::vector<std::string>& feat_by_rank,
                       bool any_float, double threshold, int32_t disable_thresholding, const char* output_path, int64_t* status_out) {
  auto write_empty = [&]() -> int32_t {
    FILE* o = std::fopen(output_path, "wb"); if (!o) { nb::set_error("report: cannot write output"); return NB_ERR_INVALID; }
    std::fclose(o); status_out[0] = 1; return NB_OK;
  };
  if (recs.empty()) return write_empty();
  Interner singles;
  // ---- groupby(cb, umi, features).sum(): keys sorted, sums in file order ----
  auto rec_less = [](const Rec& a, const Rec& b) {
    if (a.cb != b.cb) return a.cb < b.cb;
    if (a.umi != b.umi) return a.umi < b.umi;
    if (a.feat != b.feat) return a.feat < b.feat;
    return a.seq < b.seq; };
  nb::parallel_sort(recs, reducer_threads(), rec_less);
  struct Row { uint32_t cb; uint64_t umi; uint32_t feat; int64_t iscore; double fscore; };
  std::vector<Row> rows;
  rows.reserve(recs.size());
  for (size_t i = 0; i < recs.size();) {
    size_t j = i; int64_t is = 0; Kahan k;
    while (j < recs.size() && recs[j].cb == recs[i].cb && recs[j].umi == recs[i].umi && recs[j].feat == recs[i].feat) {
      if (any_float) k.add(recs[j].fscore); else is += recs[j].iscore;
      ++j; }
    rows.push_back(Row{recs[i].cb, recs[i].umi, recs[i].feat, is, any_float ? k.s : (double)is});
    i = j;
  }
  std::vector<Rec>().swap(recs);
  // single features of every distinct feature string (duplicates inside a row are kept: "A,A,B" gives A two shares)
  std::vector<std::vector<uint32_t>> split(feat_by_rank.size());
  for (size_t f = 0; f < feat_by_rank.size(); ++f) {
    const std::string& s = feat_by_rank[f]; size_t p = 0;
    for (size_t i = 0; i <= s.size(); ++i) if (i == s.size() || s[i] == ',') { split[f].push_back(singles.get(s.data() + p, i - p)); p = i + 1; }
  }
  const std::vector<uint32_t> rsingle = singles.ranks();

  // results: surviving single-feature lists (ordered by rank), interned through their byte image
  Interner results;
  auto intern_result = [&](std::vector<uint32_t>& res) -> uint32_t {
    std::sort(res.begin(), res.end(), [&](uint32_t a, uint32_t b) { return rsingle[a] < rsingle[b]; });
    return results.get((const char*)res.data(), res.size() * sizeof(uint32_t));
  };
  constexpr int32_t NOTHING = -1, EMPTY_INTERSECTION = -2;   // outcome codes beside result ids >= 0

  // ---- per (cb, umi): thresholding + intersection -> outcome ----
  std::vector<uint32_t> fid, res; std::vector<Kahan> sums; std::vector<char> present, drop, nd, survivors, keepset, inter;
  auto outcome_of = [&](size_t i, size_t j) -> int32_t {
    // local dense ids of the single features in this group
    fid.clear();
    for (size_t r = i; r < j; ++r) for (uint32_t f : split[rows[r].feat]) if (std::find(fid.begin(), fid.end(), f) == fid.end()) fid.push_back(f);
    const size_t nf = fid.size();
    auto local = [&](uint32_t f) { return (size_t)(std::find(fid.begin(), fid.end(), f) - fid.begin()); };
    survivors.assign(nf, 1);
    if (!disable_thresholding) {
      drop.assign(nf, 0);
      auto shares = [&](bool& any_row, double& ftotal, int64_t& itotal) {
        sums.assign(nf, Kahan()); present.assign(nf, 0); any_row = false; ftotal = 0.0; itotal = 0;
        for (size_t r = i; r < j; ++r) {
          size_t cnt = 0;
          for (uint32_t f : split[rows[r].feat]) if (!drop[local(f)]) ++cnt;
          if (!cnt) continue;
          any_row = true;
          const double spf = rows[r].fscore / (double)cnt;
          if (any_float) ftotal = ftotal + rows[r].fscore; else itotal += rows[r].iscore;
          for (uint32_t f : split[rows[r].feat]) { const size_t l = local(f); if (!drop[l]) { sums[l].add(spf); present[l] = 1; } }
        }
      };
      bool any_row; double ftotal; int64_t itotal;
      shares(any_row, ftotal, itotal);
      for (;;) {
        bool have_any = false;
        for (size_t l = 0; l < nf; ++l) have_any |= present[l] != 0;
        if (!have_any) { std::fill(survivors.begin(), survivors.end(), 0); break; }
        const double tot = any_float ? ftotal : (double)itotal;
        bool any_drop = false; nd.assign(nf, 0);
        for (size_t l = 0; l < nf; ++l) if (present[l] && (sums[l].s / tot < threshold)) { nd[l] = 1; any_drop = true; }
        if (!any_drop) { for (size_t l = 0; l < nf; ++l) survivors[l] = present[l]; break; }
        drop = nd;                                  // only THIS round's to_drop is excluded (utils.py:159)
        shares(any_row, ftotal, itotal);
        if (!any_row) { std::fill(survivors.begin(), survivors.end(), 0); break; }
      }
    }
    // rows keep features ∩ survivors; empty rows vanish; intersect the rest
    inter.assign(nf, 1); bool any_kept = false;
    for (size_t r = i; r < j; ++r) {
      keepset.assign(nf, 0); bool nonempty = false;
      for (uint32_t f : split[rows[r].feat]) { const size_t l = local(f); if (survivors[l]) { keepset[l] = 1; nonempty = true; } }
      if (!nonempty) continue;
      any_kept = true;
      for (size_t l = 0; l < nf; ++l) inter[l] = inter[l] && keepset[l];
    }
    if (!any_kept) return NOTHING;
    res.clear();
    for (size_t l = 0; l < nf; ++l) if (inter[l]) res.push_back(fid[l]);
    if (res.empty()) return EMPTY_INTERSECTION;
    return (int32_t)intern_result(res);
  };

  std::unordered_map<uint64_t, int32_t> single_row_memo;   // (feature string, score) -> outcome of a one-row UMI
  struct Tally { uint32_t cb, result; int64_t count; };
  std::vector<Tally> tallies; std::vector<uint32_t> cell_results;
  int64_t dropped = 0; bool any_umi = false;
  auto flush_cell = [&](uint32_t cb) {
    std::sort(cell_results.begin(), cell_results.end());
    for (size_t a = 0; a < cell_results.size();) {
      size_t b = a; while (b < cell_results.size() && cell_results[b] == cell_results[a]) ++b;
      tallies.push_back(Tally{cb, cell_results[a], (int64_t)(b - a)}); a = b;
    }
    cell_results.clear();
  };
  for (size_t i = 0; i < rows.size();) {
    size_t j = i;
    while (j < rows.size() && rows[j].cb == rows[i].cb && rows[j].umi == rows[i].umi) ++j;
    int32_t oc;
    if (j == i + 1 && !any_float && rows[i].iscore >= 0 && rows[i].iscore < ((int64_t)1 << 31)) {
      const uint64_t key = (uint64_t)rows[i].feat << 32 | (uint64_t)rows[i].iscore;
      auto it = single_row_memo.find(key);
      if (it == single_row_memo.end()) it = single_row_memo.emplace(key, outcome_of(i, j)).first;
      oc = it->second;
    } else {
      oc = outcome_of(i, j);
    }
    if (oc != NOTHING) any_umi = true;
    if (oc == EMPTY_INTERSECTION) ++dropped;
    else if (oc >= 0) cell_results.push_back((uint32_t)oc);
    if (j == rows.size() || rows[j].cb != rows[i].cb) flush_cell(rows[i].cb);
    i = j;
  }
  if (!any_umi) return write_empty();
  status_out[1] = dropped;
  // result strings and their lexicographic ranks; output ordered by (cell_barcode, feature)
  std::vector<std::string> result_str(results.size());
  for (uint32_t r = 0; r < results.size(); ++r) {
    const uint32_t* ids = (const uint32_t*)results.ptr(r); const size_t n = results.len(r) / sizeof(uint32_t);
    std::string& o = result_str[r];
    for (size_t t = 0; t < n; ++t) { uint32_t id; std::memcpy(&id, (const char*)ids + t * sizeof(uint32_t), sizeof id); if (t) o += ','; o.append(singles.ptr(id), singles.len(id)); }
  }
  std::vector<uint32_t> order(results.size()), rrank(results.size());
  for (uint32_t i = 0; i < order.size(); ++i) order[i] = i;
  std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return result_str[a] < result_str[b]; });
  for (uint32_t i = 0; i < order.size(); ++i) rrank[order[i]] = i;
  std::sort(tallies.begin(), tallies.end(), [&](const Tally& a, const Tally& b) { return a.cb != b.cb ? a.cb < b.cb : rrank[a.result] < rrank[b.result]; });
  FILE* o = std::fopen(output_path, "wb");
  if (!o) { nb::set_error("report: cannot write output"); return NB_ERR_INVALID; }
  std::vector<char> obuf(1 << 22); std::setvbuf(o, obuf.data(), _IOFBF, obuf.size());
  std::vector<std::string> quoted_result(results.size());
  for (uint32_t r = 0; r < results.size(); ++r) quoted_result[r] = quote_minimal(result_str[r]);
  std::string quoted_cb; uint32_t quoted_cb_of = UINT32_MAX;
  for (const Tally& t : tallies) {
    if (t.cb != quoted_cb_of) { quoted_cb = quote_minimal(cb_by_rank[t.cb]); quoted_cb_of = t.cb; }
    std::fwrite(quoted_result[t.result].data(), 1, quoted_result[t.result].size(), o);
    std::fprintf(o, "\t%lld\t", (long long)t.count);
    std::fwrite(quoted_cb.data(), 1, quoted_cb.size(), o);
    std::fputc('\n', o);
  }
  std::fclose(o);
  return NB_OK;
}

// plain text, BGZF (block-parallel inflate: the tables this engine writes) or any other gzip file (serial inflate)
struct TextSource {
  int fd = -1; FILE* fp = nullptr; int kind = 0;   // 0 plain, 1 BGZF, 2 gzip
  nb::BgzfReader bgzf; nb::GzipStream* gz = nullptr; std::vector<uint8_t> stage; const char* error = nullptr;
  bool open(const char* path, int threads) {
    fd = ::open(path, O_RDONLY);
    if (fd < 0) return false;
    uint8_t first[18]; const long got = (long)::read(fd, first, 18);
    ::lseek(fd, 0, SEEK_SET);
    if (got >= 18 && nb::BgzfReader::block_size(first, (size_t)got) > 0) { kind = 1; fp = fdopen(fd, "rb"); if (!fp) return false; bgzf.fp = fp; bgzf.threads = threads; }
    else if (got >= 2 && first[0] == 0x1f && first[1] == 0x8b) { kind = 2; gz = new nb::GzipStream(); gz->fd = fd; stage.resize(nb::GzipStream::HISTORY + ((size_t)8 << 20) + nb::GzipStream::SLACK + 64); }
    return true;
  }
  // up to cap bytes (cap >= 65536) to dst; 0 at end of file, -1 on error
  long read(char* dst, size_t cap) {
    if (kind == 1) { const long n = bgzf.read((uint8_t*)dst, cap); if (n < 0) error = bgzf.error; return n; }
    if (kind == 2) {
      uint8_t* payload = stage.data() + nb::GzipStream::HISTORY;
      const long n = gz->read(payload, std::min<size_t>(cap - 512, (size_t)8 << 20));   // (may exceed its cap by one match: < 512 bytes)
      if (n < 0) { error = gz->error; return n; }
      std::memcpy(dst, payload, (size_t)n); return n;
    }
    const long n = (long)::read(fd, dst, std::min<size_t>(cap, (size_t)1 << 30));
    if (n < 0) error = "read error";
    return n;
  }
  ~TextSource() { delete gz; if (fp) std::fclose(fp); else if (fd >= 0) ::close(fd); }
};

// ---- the TSV front end: blocks of text are parsed by worker threads with thread-local interners, then merged ----
constexpr int32_t RETRY_WITH_INTERNED_UMIS = 1000;

// 3 bits per base, first base most significant, left-aligned in 63 bits: numeric order = lexicographic order of the
// strings (A < C < G < N < T as in ASCII; a proper prefix sorts first because the padding code is 0)
inline bool pack_umi(const char* b, size_t n, uint64_t* out) {
  if (n > 21) return false;
  uint64_t v = 0;
  for (size_t i = 0; i < n; ++i) {
    uint64_t c;
    switch (b[i]) { case 'A': c = 1; break; case 'C': c = 2; break; case 'G': c = 3; break; case 'N': c = 4; break; case 'T': c = 5; break; default: return false; }
    v |= c << (60 - 3 * i);
  }
  *out = v; return true;
}

struct ParseWorker {
  Interner cbs, feats, raw_feats, umis;
  std::vector<uint32_t> canon_of;
  std::vector<std::string> parts;
  std::vector<Rec> recs;                 // ids local to this worker; seq = index inside the slice
  std::vector<size_t> slice_begin;       // recs.size() at the start of every block's slice
  bool any_float = false, umi_unpackable = false, any_body = false;
  int perr = 0;
};

// parses the lines of [b, e) (no header); returns 0, or -2 for a non-numeric score
int parse_slice(ParseWorker& w, const char* b, const char* e, const int ix[4], bool pack_umis) {
  int max_col = std::max(std::max(ix[0], ix[1]), std::max(ix[2], ix[3]));
  while (b < e) {
    const char* nl = (const char*)std::memchr(b, '\n', (size_t)(e - b));
    const char* le = nl ? nl : e;
    size_t n = (size_t)(le - b);
    const char* line = b;
    b = nl ? nl + 1 : e;
    if (n && line[n - 1] == '\r') --n;
    if (n == 0) continue;                         // pandas skips blank lines
    w.any_body = true;
    if (ix[0] < 0 || ix[1] < 0 || ix[2] < 0 || ix[3] < 0) { w.perr = -1; return -1; }
    const char* fb[4] = {nullptr, nullptr, nullptr, nullptr}; size_t fn[4] = {0, 0, 0, 0};
    size_t s = 0; int col = 0;
    for (size_t i = 0; i <= n && col <= max_col; ++i) if (i == n || line[i] == '\t') {
        for (int k = 0; k < 4; ++k) if (ix[k] == col) { fb[k] = line + s; fn[k] = i - s; }
        ++col; s = i + 1; }
    bool skip = false;
    for (int k = 0; k < 4; ++k) if (!fb[k] || is_na(fb[k], fn[k])) skip = true;   // dropna + != '' (__main__.py:244-245)
    if (skip) continue;
    Rec r{};
    if (!parse_int_strict(fb[1], fn[1], &r.iscore)) {
      std::string tok(fb[1], fn[1]); char* end = nullptr;
      r.fscore = std::strtod(tok.c_str(), &end);
      if (end == tok.c_str() || *end != '\0') { w.perr = -2; return -2; }
      w.any_float = true; r.iscore = INT64_MIN;
    } else r.fscore = (double)r.iscore;
    // sort the features inside the row (__main__.py:248): once per distinct raw string
    const uint32_t raw = w.raw_feats.get(fb[0], fn[0]);
    if (raw == w.canon_of.size()) {
      w.parts.clear(); size_t p = 0;
      for (size_t i = 0; i <= fn[0]; ++i) if (i == fn[0] || fb[0][i] == ',') { w.parts.emplace_back(fb[0] + p, i - p); p = i + 1; }
      std::sort(w.parts.begin(), w.parts.end());
      std::string joined;
      for (size_t i = 0; i < w.parts.size(); ++i) { if (i) joined += ','; joined += w.parts[i]; }
      w.canon_of.push_back(w.feats.get(joined));
    }
    r.feat = w.canon_of[raw]; r.cb = w.cbs.get(fb[2], fn[2]);
    if (pack_umis) { if (!pack_umi(fb[3], fn[3], &r.umi)) { w.umi_unpackable = true; return 0; } }
    else r.umi = w.umis.get(fb[3], fn[3]);
    w.recs.push_back(r);
  }
  return 0;
}

template <class F>
void run_threads(int T, F&& f) {
  if (T <= 1) { f(0); return; }
  std::vector<std::thread> th;
  for (int t = 1; t < T; ++t) th.emplace_back([&f, t] { f(t); });
  f(0);
  for (auto& x : th) x.join();
}

int32_t report_tsv_pass(const char* input_path, const char* output_path, double threshold, int32_t disable_thresholding, int64_t* status_out,
                        int threads, bool pack_umis) {
  status_out[0] = 0; status_out[1] = 0;
  // plain and gzip files alike: the reference's pd.read_csv infers gzip from a ".gz" name
  // (/root/reference/nimble/__main__.py:219), which is what the aligner writes for BAM input
  TextSource in;
  if (!in.open(input_path, threads)) { nb::set_error(std::string("nb_report_tsv: cannot open ") + input_path); return NB_ERR_INVALID; }
  auto write_empty = [&]() -> int32_t {
    FILE* o = std::fopen(output_path, "wb"); if (!o) { nb::set_error("nb_report_tsv: cannot write output"); return NB_ERR_INVALID; }
    std::fclose(o); status_out[0] = 1; return NB_OK;
  };
  constexpr size_t BLOCK = (size_t)64 << 20;
  std::vector<char> buf(BLOCK + (1 << 16)); size_t have = 0; bool eof = false, any_bytes = false, header_done = false;
  int ix[4] = {-1, -1, -1, -1};
  std::vector<ParseWorker> workers((size_t)threads);
  size_t n_blocks = 0;
  size_t target = BLOCK;
  while (!eof) {
    while (have < target && !eof) {
      if (buf.size() - have < 65536) buf.resize(buf.size() + BLOCK);
      const long gr = in.read(buf.data() + have, buf.size() - have);
      if (gr < 0) { nb::set_error(std::string("nb_report_tsv: ") + (in.error ? in.error : "read error")); return NB_ERR_INVALID; }
      if (gr == 0) eof = true;
      any_bytes |= gr > 0; have += (size_t)gr;
    }
    // the part that ends on a line boundary (everything at end of file)
    size_t usable = have;
    if (!eof) { while (usable > 0 && buf[usable - 1] != '\n') --usable; if (usable == 0) { target = have + BLOCK; continue; } }   // (one line longer than the block)
    target = BLOCK;
    const char* b = buf.data(); const char* e = buf.data() + usable;
    if (!header_done && b < e) {
      const char* nl = (const char*)std::memchr(b, '\n', (size_t)(e - b));
      const char* le = nl ? nl : e; size_t n = (size_t)(le - b);
      if (n && b[n - 1] == '\r') --n;
      size_t s = 0; int col = 0;
      for (size_t i = 0; i <= n; ++i) if (i == n || b[i] == '\t') {
          const std::string name(b + s, i - s);
          static const char* const want[4] = {"nimble_features", "nimble_score", "r1_CB", "r1_UB"};
          for (int k = 0; k < 4; ++k) if (ix[k] < 0 && name == want[k]) ix[k] = col;   // first occurrence, as list.index
          ++col; s = i + 1; }
      header_done = true;
      b = nl ? nl + 1 : e;
    }
    // slices on line boundaries
    const int T = (int)std::min<size_t>((size_t)threads, std::max<size_t>(1, (size_t)(e - b) >> 20));
    std::vector<const char*> cut((size_t)T + 1);
    cut[0] = b; cut[(size_t)T] = e;
    for (int t = 1; t < T; ++t) {
      const char* p = b + (size_t)(e - b) * (size_t)t / (size_t)T;
      if (p < cut[(size_t)t - 1]) p = cut[(size_t)t - 1];
      const char* nl = (const char*)std::memchr(p, '\n', (size_t)(e - p));
      cut[(size_t)t] = nl ? nl + 1 : e;
    }
    for (auto& w : workers) w.slice_begin.push_back(w.recs.size());
    run_threads(T, [&](int t) { parse_slice(workers[(size_t)t], cut[(size_t)t], cut[(size_t)t + 1], ix, pack_umis); });
    ++n_blocks;
    bool stop = false;
    for (auto& w : workers) if (w.perr || w.umi_unpackable) stop = true;
    if (stop) break;
    std::memmove(buf.data(), buf.data() + usable, have - usable); have -= usable;
  }
  bool any_body = false, any_float = false; int perr = 0;
  for (auto& w : workers) { any_body |= w.any_body; any_float |= w.any_float; if (w.perr && !perr) perr = w.perr; }
  if (perr == -1 && any_body) { nb::set_error("nb_report_tsv: input TSV lacks nimble_features / nimble_score / r1_CB / r1_UB"); return NB_ERR_INVALID; }
  if (perr == -2) { nb::set_error("nb_report_tsv: non-numeric nimble_score"); return NB_ERR_INVALID; }
  for (auto& w : workers) if (w.umi_unpackable) return RETRY_WITH_INTERNED_UMIS;
  size_t total = 0;
  for (auto& w : workers) { w.slice_begin.push_back(w.recs.size()); total += w.recs.size(); }
  if (!any_bytes || total == 0) return write_empty();

  // ---- merge: worker-local ids -> global ranks (pandas sorts group keys lexicographically); seq = file order ----
  Interner cbs, feats, umis;
  std::vector<std::vector<uint32_t>> map_cb(workers.size()), map_feat(workers.size()), map_umi(workers.size());
  for (size_t t = 0; t < workers.size(); ++t) {
    ParseWorker& w = workers[t];
    map_cb[t].resize(w.cbs.size()); map_feat[t].resize(w.feats.size());
    for (uint32_t i = 0; i < w.cbs.size(); ++i) map_cb[t][i] = cbs.get(w.cbs.ptr(i), w.cbs.len(i));
    for (uint32_t i = 0; i < w.feats.size(); ++i) map_feat[t][i] = feats.get(w.feats.ptr(i), w.feats.len(i));
    if (!pack_umis) { map_umi[t].resize(w.umis.size()); for (uint32_t i = 0; i < w.umis.size(); ++i) map_umi[t][i] = umis.get(w.umis.ptr(i), w.umis.len(i)); }
  }
  const std::vector<uint32_t> rcb = cbs.ranks(), rfeat = feats.ranks();
  std::vector<uint32_t> rumi;
  if (!pack_umis && any_float) rumi = umis.ranks();      // (the UMI order only matters for float sums; packed codes are ordered already)
  std::vector<std::string> cb_by_rank(cbs.size()), feat_by_rank(feats.size());
  for (uint32_t i = 0; i < cbs.size(); ++i) cb_by_rank[rcb[i]] = cbs.str(i);
  for (uint32_t i = 0; i < feats.size(); ++i) feat_by_rank[rfeat[i]] = feats.str(i);
  // file-order sequence numbers: blocks in order, within a block the workers' slices in order
  std::vector<std::vector<uint64_t>> seq_base(workers.size(), std::vector<uint64_t>(n_blocks, 0));
  uint64_t running = 0;
  for (size_t blk = 0; blk < n_blocks; ++blk)
    for (size_t t = 0; t < workers.size(); ++t) { seq_base[t][blk] = running; running += workers[t].slice_begin[blk + 1] - workers[t].slice_begin[blk]; }
  if (running >= ((uint64_t)1 << 32)) { nb::set_error("nb_report_tsv: more than 2^32 rows"); return NB_ERR_UNSUPPORTED; }
  std::vector<Rec> recs(total);
  std::vector<size_t> out_off(workers.size() + 1, 0);
  for (size_t t = 0; t < workers.size(); ++t) out_off[t + 1] = out_off[t] + workers[t].recs.size();
  run_threads((int)workers.size(), [&](int ti) {
    const size_t t = (size_t)ti; ParseWorker& w = workers[t];
    for (size_t blk = 0; blk < n_blocks; ++blk)
      for (size_t i = w.slice_begin[blk]; i < w.slice_begin[blk + 1]; ++i) {
        Rec r = w.recs[i];
        r.cb = rcb[map_cb[t][r.cb]]; r.feat = rfeat[map_feat[t][r.feat]];
        if (!pack_umis) { const uint32_t g = map_umi[t][(size_t)r.umi]; r.umi = any_float ? rumi[g] : g; }
        r.seq = (uint32_t)(seq_base[t][blk] + (i - w.slice_begin[blk]));
        recs[out_off[t] + i] = r;
      }
    std::vector<Rec>().swap(w.recs);
  });
  return reduce_records(recs, cb_by_rank, feat_by_rank, any_float, threshold, disable_thresholding, output_path, status_out);
}

}  // namespace

extern "C" {

// status_out[0] = 0 normal, 1 wrote the empty output ("no data"); status_out[1] = UMIs dropped for empty intersection
int32_t nb_report_tsv(const char* input_path, const char* output_path, double threshold, int32_t disable_thresholding, int64_t* status_out) {
  if (!input_path || !output_path || !status_out) { nb::set_error("nb_report_tsv: null argument"); return NB_ERR_INVALID; }
  const int threads = reducer_threads();
  // UMIs are first handled as packed codes (no hash table for the largest key set of the table); a UMI that cannot be
  // packed (other letters, longer than 21) makes the whole parse start over with interned UMI strings
  int32_t rc = report_tsv_pass(input_path, output_path, threshold, disable_thresholding, status_out, threads, true);
  if (rc == RETRY_WITH_INTERNED_UMIS) rc = report_tsv_pass(input_path, output_path, threshold, disable_thresholding, status_out, threads, false);
  return rc;
}

// Same reduction from arrays (no TSV): row i = (cb_ids[i] -> cb_strings, umi_keys[i], feat_ids[i] -> feat_strings with the
// features of a string already sorted and comma-joined, scores[i]). Rows need not be merged or ordered. Used by the
// fused per-cell counting of `align` (nimble_amd/fused.py), which pre-merges identical rows on the GPU.
int32_t nb_report_rows(int64_t n, const uint32_t* cb_ids, const uint64_t* umi_keys, const uint32_t* feat_ids, const int64_t* scores,
                       int64_t n_cb, const char* const* cb_strings, int64_t n_feat, const char* const* feat_strings, double threshold,
                       int32_t disable_thresholding, const char* output_path, int64_t* status_out) {
  if (n < 0 || !output_path || !status_out || (n && (!cb_ids || !umi_keys || !feat_ids || !scores || !cb_strings || !feat_strings))) { nb::set_error("nb_report_rows: bad arguments"); return NB_ERR_INVALID; }
  status_out[0] = 0; status_out[1] = 0;
  Interner cbs, feats;
  std::vector<uint32_t> cb_of((size_t)n_cb), feat_of((size_t)n_feat);
  for (int64_t i = 0; i < n_cb; ++i) cb_of[(size_t)i] = cbs.get(cb_strings[i], std::strlen(cb_strings[i]));
  for (int64_t i = 0; i < n_feat; ++i) feat_of[(size_t)i] = feats.get(feat_strings[i], std::strlen(feat_strings[i]));
  const std::vector<uint32_t> rcb = cbs.ranks(), rfeat = feats.ranks();
  std::vector<std::string> cb_by_rank(cbs.size()), feat_by_rank(feats.size());
  for (uint32_t i = 0; i < cbs.size(); ++i) cb_by_rank[rcb[i]] = cbs.str(i);
  for (uint32_t i = 0; i < feats.size(); ++i) feat_by_rank[rfeat[i]] = feats.str(i);
  std::vector<Rec> recs; recs.reserve((size_t)n);
  for (int64_t i = 0; i < n; ++i) {
    if ((int64_t)cb_ids[i] >= n_cb || (int64_t)feat_ids[i] >= n_feat) { nb::set_error("nb_report_rows: id out of range"); return NB_ERR_INVALID; }
    if (cb_strings[cb_ids[i]][0] == 0 || feat_strings[feat_ids[i]][0] == 0) continue;        // null cb / features rows are dropped
    Rec r{}; r.cb = rcb[cb_of[cb_ids[i]]]; r.umi = umi_keys[i]; r.feat = rfeat[feat_of[feat_ids[i]]]; r.seq = (uint32_t)i;
    r.iscore = scores[i]; r.fscore = (double)scores[i];
    recs.push_back(r);
  }
  return reduce_records(recs, cb_by_rank, feat_by_rank, false, threshold, disable_thresholding, output_path, status_out);
}

}  // extern "C"
