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
#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "nb_internal.h"

namespace {

const char* const kNA[] = {"", "#N/A", "#N/A N/A", "#NA", "-1.#IND", "-1.#QNAN", "-NaN", "-nan", "1.#IND", "1.#QNAN", "<NA>", "N/A", "NA",
                           "NULL", "NaN", "None", "n/a", "nan", "null"};

bool is_na(const char* b, size_t n) {
  for (const char* s : kNA) if (std::strlen(s) == n && std::memcmp(s, b, n) == 0) return true;
  return false;
}

struct Interner {
  std::unordered_map<std::string, uint32_t> map;
  std::vector<std::string> strs;
  uint32_t get(const std::string& s) {
    auto it = map.find(s);
    if (it != map.end()) return it->second;
    const uint32_t id = (uint32_t)strs.size();
    map.emplace(s, id); strs.push_back(s); return id;
  }
  // rank[id] = position of strs[id] in lexicographic (byte) order — Python's str ordering for the ASCII data at hand
  std::vector<uint32_t> ranks() const {
    std::vector<uint32_t> order(strs.size());
    for (uint32_t i = 0; i < order.size(); ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return strs[a] < strs[b]; });
    std::vector<uint32_t> r(strs.size());
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


// The reduction proper, shared by the TSV and the array entry points. recs carry RANKS: cb / feat index cb_by_rank /
// feat_by_rank (strings in lexicographic order), umi is any key that identifies the UMI (only grouping matters: the
// output is ordered by (cell, feature) and integer scores do not depend on summation order; for float scores the TSV
// path passes the rank of the UMI string, which reproduces pandas' group order).
int32_t reduce_records(std::vector<Rec>& recs, const std::vector<std::string>& cb_by_rank, const std::vector<std::string>& feat_by_rank,
                       bool any_float, double threshold, int32_t disable_thresholding, const char* output_path, int64_t* status_out) {
  auto write_empty = [&]() -> int32_t {
    FILE* o = std::fopen(output_path, "wb"); if (!o) { nb::set_error("report: cannot write output"); return NB_ERR_INVALID; }
    std::fclose(o); status_out[0] = 1; return NB_OK;
  };
  if (recs.empty()) return write_empty();
  Interner singles;
  // ---- groupby(cb, umi, features).sum(): keys sorted, sums in file order ----
  std::sort(recs.begin(), recs.end(), [&](const Rec& a, const Rec& b) {
    if (a.cb != b.cb) return a.cb < b.cb;
    if (a.umi != b.umi) return a.umi < b.umi;
    if (a.feat != b.feat) return a.feat < b.feat;
    return a.seq < b.seq; });
  struct Row { uint32_t cb; uint64_t umi; uint32_t feat; int64_t iscore; double fscore; };
  std::vector<Row> rows;
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
    for (size_t i = 0; i <= s.size(); ++i) if (i == s.size() || s[i] == ',') { split[f].push_back(singles.get(s.substr(p, i - p))); p = i + 1; }
  }
  const std::vector<uint32_t> rsingle = singles.ranks();

  // ---- per (cb, umi): thresholding + intersection ----
  std::unordered_map<std::string, int64_t> counts;   // key = cb \x01 feature-string
  int64_t dropped = 0; bool any_umi = false;
  std::vector<uint32_t> fid; std::vector<Kahan> sums; std::vector<char> present, drop, keepset, inter;
  fid.reserve(64);
  for (size_t i = 0; i < rows.size();) {
    size_t j = i;
    while (j < rows.size() && rows[j].cb == rows[i].cb && rows[j].umi == rows[i].umi) ++j;
    // local dense ids of the single features in this group
    fid.clear();
    for (size_t r = i; r < j; ++r) for (uint32_t f : split[rows[r].feat]) if (std::find(fid.begin(), fid.end(), f) == fid.end()) fid.push_back(f);
    const size_t nf = fid.size();
    auto local = [&](uint32_t f) { return (size_t)(std::find(fid.begin(), fid.end(), f) - fid.begin()); };
    std::vector<char> survivors(nf, 1);
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
        bool any_drop = false; std::vector<char> nd(nf, 0);
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
    if (any_kept) {
      any_umi = true;
      std::vector<uint32_t> res;
      for (size_t l = 0; l < nf; ++l) if (inter[l]) res.push_back(fid[l]);
      if (res.empty()) ++dropped;
      else {
        std::sort(res.begin(), res.end(), [&](uint32_t a, uint32_t b) { return rsingle[a] < rsingle[b]; });
        std::string key = cb_by_rank[rows[i].cb]; key += '\x01';
        for (size_t t = 0; t < res.size(); ++t) { if (t) key += ','; key += singles.strs[res[t]]; }
        ++counts[key];
      }
    }
    i = j;
  }
  if (!any_umi) return write_empty();
  status_out[1] = dropped;
  std::vector<std::pair<std::string, int64_t>> out(counts.begin(), counts.end());
  std::sort(out.begin(), out.end(), [](const auto& a, const auto& b) {   // (cell_barcode, feature) lexicographic
    const size_t pa = a.first.find('\x01'), pb = b.first.find('\x01');
    const int c = a.first.compare(0, pa, b.first, 0, pb);
    if (c) return c < 0;
    return a.first.compare(pa + 1, std::string::npos, b.first, pb + 1, std::string::npos) < 0; });
  FILE* o = std::fopen(output_path, "wb");
  if (!o) { nb::set_error("report: cannot write output"); return NB_ERR_INVALID; }
  for (const auto& kv : out) {
    const size_t p = kv.first.find('\x01');
    const std::string cb = kv.first.substr(0, p), feat = kv.first.substr(p + 1);
    std::fprintf(o, "%s\t%lld\t%s\n", quote_minimal(feat).c_str(), (long long)kv.second, quote_minimal(cb).c_str());
  }
  std::fclose(o);
  return NB_OK;
}

}  // namespace

extern "C" {

// status_out[0] = 0 normal, 1 wrote the empty output ("no data"); status_out[1] = UMIs dropped for empty intersection
int32_t nb_report_tsv(const char* input_path, const char* output_path, double threshold, int32_t disable_thresholding, int64_t* status_out) {
  if (!input_path || !output_path || !status_out) { nb::set_error("nb_report_tsv: null argument"); return NB_ERR_INVALID; }
  status_out[0] = 0; status_out[1] = 0;
  // gzopen reads plain and gzip files alike: the reference's pd.read_csv infers gzip from a ".gz" name
  // (/root/reference/nimble/__main__.py:219), which is what the aligner writes for BAM input
  gzFile in = gzopen(input_path, "rb");
  if (!in) { nb::set_error(std::string("nb_report_tsv: cannot open ") + input_path); return NB_ERR_INVALID; }
  gzbuffer(in, 1 << 20);
  auto write_empty = [&]() -> int32_t {
    FILE* o = std::fopen(output_path, "wb"); if (!o) { nb::set_error("nb_report_tsv: cannot write output"); return NB_ERR_INVALID; }
    std::fclose(o); status_out[0] = 1; return NB_OK;
  };
  // ---- stream lines ----
  std::vector<char> buf(1 << 22); size_t have = 0; bool eof = false;
  std::string line; bool header_done = false; int ix[4] = {-1, -1, -1, -1};
  Interner cbs, umis, feats;
  std::vector<Rec> recs; bool any_float = false; uint32_t seq = 0;
  std::vector<std::string> parts;
  auto handle_line = [&](const char* b, size_t n) -> int {
    if (n && b[n - 1] == '\r') --n;
    if (!header_done) {
      header_done = true;
      size_t s = 0; int col = 0;
      for (size_t i = 0; i <= n; ++i) if (i == n || b[i] == '\t') {
          const std::string name(b + s, i - s);
          static const char* const want[4] = {"nimble_features", "nimble_score", "r1_CB", "r1_UB"};
          for (int k = 0; k < 4; ++k) if (ix[k] < 0 && name == want[k]) ix[k] = col;   // first occurrence, as list.index
          ++col; s = i + 1; }
      return 0;
    }
    if (n == 0) return 0;                       // pandas skips blank lines
    if (ix[0] < 0 || ix[1] < 0 || ix[2] < 0 || ix[3] < 0) return -1;
    const char* fb[4] = {nullptr, nullptr, nullptr, nullptr}; size_t fn[4] = {0, 0, 0, 0};
    size_t s = 0; int col = 0;
    for (size_t i = 0; i <= n; ++i) if (i == n || b[i] == '\t') {
        for (int k = 0; k < 4; ++k) if (ix[k] == col) { fb[k] = b + s; fn[k] = i - s; }
        ++col; s = i + 1; }
    for (int k = 0; k < 4; ++k) if (!fb[k] || is_na(fb[k], fn[k])) return 0;   // dropna + != '' (__main__.py:244-245)
    Rec r{}; r.seq = seq++;
    if (!parse_int_strict(fb[1], fn[1], &r.iscore)) {
      std::string tok(fb[1], fn[1]); char* end = nullptr;
      r.fscore = std::strtod(tok.c_str(), &end);
      if (end == tok.c_str() || *end != '\0') return -2;
      any_float = true; r.iscore = INT64_MIN;
    } else r.fscore = (double)r.iscore;
    // sort the features inside the row (__main__.py:248)
    parts.clear(); size_t p = 0;
    for (size_t i = 0; i <= fn[0]; ++i) if (i == fn[0] || fb[0][i] == ',') { parts.emplace_back(fb[0] + p, i - p); p = i + 1; }
    std::sort(parts.begin(), parts.end());
    std::string joined;
    for (size_t i = 0; i < parts.size(); ++i) { if (i) joined += ','; joined += parts[i]; }
    r.feat = feats.get(joined); r.cb = cbs.get(std::string(fb[2], fn[2])); r.umi = umis.get(std::string(fb[3], fn[3]));
    recs.push_back(r); return 0;
  };
  bool any_bytes = false, any_body = false; int perr = 0;
  while (!eof && perr == 0) {
    const int gr = gzread(in, buf.data() + have, (unsigned)std::min<size_t>(buf.size() - have, (size_t)1 << 30));
    const size_t got = gr > 0 ? (size_t)gr : 0;
    if (got == 0) eof = true;
    any_bytes |= got > 0; have += got;
    size_t start = 0;
    for (;;) {
      const char* nl = (const char*)std::memchr(buf.data() + start, '\n', have - start);
      if (!nl) break;
      const size_t n = (size_t)(nl - (buf.data() + start));
      if (header_done && n > 0) any_body = true;
      perr = handle_line(buf.data() + start, n); if (perr) break;
      start += n + 1;
    }
    if (eof && start < have && perr == 0) { if (header_done) any_body = true; perr = handle_line(buf.data() + start, have - start); start = have; }
    std::memmove(buf.data(), buf.data() + start, have - start); have -= start;
    if (have == buf.size()) buf.resize(buf.size() * 2);
  }
  gzclose(in);
  if (perr == -1 && any_body) { nb::set_error("nb_report_tsv: input TSV lacks nimble_features / nimble_score / r1_CB / r1_UB"); return NB_ERR_INVALID; }
  if (perr == -2) { nb::set_error("nb_report_tsv: non-numeric nimble_score"); return NB_ERR_INVALID; }
  if (!any_bytes || recs.empty()) return write_empty();

  // strings -> ranks (pandas sorts group keys lexicographically), then the shared reduction
  const std::vector<uint32_t> rcb = cbs.ranks(), rumi = umis.ranks(), rfeat = feats.ranks();
  std::vector<std::string> cb_by_rank(cbs.strs.size()), feat_by_rank(feats.strs.size());
  for (size_t i = 0; i < cbs.strs.size(); ++i) cb_by_rank[rcb[i]] = cbs.strs[i];
  for (size_t i = 0; i < feats.strs.size(); ++i) feat_by_rank[rfeat[i]] = feats.strs[i];
  for (auto& r : recs) { r.cb = rcb[r.cb]; r.umi = rumi[(size_t)r.umi]; r.feat = rfeat[r.feat]; }
  return reduce_records(recs, cb_by_rank, feat_by_rank, any_float, threshold, disable_thresholding, output_path, status_out);
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
  for (int64_t i = 0; i < n_cb; ++i) cb_of[(size_t)i] = cbs.get(cb_strings[i]);
  for (int64_t i = 0; i < n_feat; ++i) feat_of[(size_t)i] = feats.get(feat_strings[i]);
  const std::vector<uint32_t> rcb = cbs.ranks(), rfeat = feats.ranks();
  std::vector<std::string> cb_by_rank(cbs.strs.size()), feat_by_rank(feats.strs.size());
  for (size_t i = 0; i < cbs.strs.size(); ++i) cb_by_rank[rcb[i]] = cbs.strs[i];
  for (size_t i = 0; i < feats.strs.size(); ++i) feat_by_rank[rfeat[i]] = feats.strs[i];
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
