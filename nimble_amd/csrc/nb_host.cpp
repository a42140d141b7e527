// nb_host.cpp — host helpers of the 10x path (product code): cell-barcode whitelist correction and per-read TSV rows.
//
// Whitelist correction restates /root/reference/nimble/fastq_barcode_processor.py:17-128 (same rule as
// nimble_amd/barcodes.py, which the tests hold it to): exact match -> keep; else the whitelist entries at Hamming
// distance 1; none -> drop the pair; one -> it; several -> the one whose differing base has the lowest phred quality in
// the raw read (ties: lexicographically smallest — the reference's tie order is Python set iteration, not
// reproducible). Results are cached per raw barcode: the first sighting decides (:90-92). UMIs are never corrected.
// The TSV formatter emits the per-read rows of docs/SPEC.md §5 (columns consumed at nimble/__main__.py:237-245).
#include <zlib.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "nb_internal.h"

struct nb_whitelist {
  std::unordered_set<std::string> set;
  std::unordered_map<std::string, std::string> cache;   // raw -> corrected ("" = no correction)
  int64_t perfect = 0, corrected = 0, none = 0;
};

extern "C" {

int32_t nb_whitelist_load(const char* path, nb_whitelist** out) {
  if (!path || !out) { nb::set_error("nb_whitelist_load: null argument"); return NB_ERR_INVALID; }
  gzFile f = gzopen(path, "rb");
  if (!f) { nb::set_error(std::string("nb_whitelist_load: cannot open ") + path); return NB_ERR_INVALID; }
  nb_whitelist* w = new nb_whitelist();
  std::vector<char> line(1 << 16);
  while (gzgets(f, line.data(), (int)line.size())) {
    size_t n = std::strlen(line.data());
    while (n && (line[n - 1] == '\n' || line[n - 1] == '\r' || line[n - 1] == ' ' || line[n - 1] == '\t')) --n;
    size_t b = 0; while (b < n && (line[b] == ' ' || line[b] == '\t')) ++b;
    if (n > b) w->set.emplace(line.data() + b, n - b);
  }
  gzclose(f);
  *out = w; return NB_OK;
}

int32_t nb_whitelist_destroy(nb_whitelist* w) { delete w; return NB_OK; }

int64_t nb_whitelist_size(const nb_whitelist* w) { return w ? (int64_t)w->set.size() : 0; }

// n raw barcodes of fixed width cb_len (+ their quality characters) -> corrected barcodes (same layout) and ok[i] = 1,
// or ok[i] = 0 when the pair must be dropped. Sequential on purpose: the cache makes the first sighting decide.
int32_t nb_whitelist_correct(nb_whitelist* w, const char* raw, const char* qual, int64_t n, int32_t cb_len, char* out, uint8_t* ok,
                             int64_t* stats /* perfect, corrected, none: cumulative */) {
  if (!w || !raw || !qual || !out || !ok || cb_len < 1) { nb::set_error("nb_whitelist_correct: bad arguments"); return NB_ERR_INVALID; }
  std::string key, cand, best;
  for (int64_t i = 0; i < n; ++i) {
    key.assign(raw + i * cb_len, (size_t)cb_len);
    auto it = w->cache.find(key);
    const std::string* res;
    if (it != w->cache.end()) res = &it->second;
    else {
      std::string r;
      if (w->set.count(key)) r = key;
      else {
        int n_cand = 0; int best_q = 1 << 30; best.clear();
        const char* q = qual + i * cb_len;
        for (int p = 0; p < cb_len; ++p)
          for (const char b : {'A', 'C', 'G', 'T'}) {
            if (b == key[(size_t)p]) continue;
            cand = key; cand[(size_t)p] = b;
            if (!w->set.count(cand)) continue;
            ++n_cand;
            const int ql = (int)(unsigned char)q[p] - 33;
            if (ql < best_q || (ql == best_q && cand < best)) { best_q = ql; best = cand; }
          }
        if (n_cand >= 1) r = best;   // a single candidate is returned whatever its quality; several: lowest quality, then smallest
      }
      res = &w->cache.emplace(key, std::move(r)).first->second;
    }
    if (res->empty()) { ok[i] = 0; ++w->none; std::memset(out + i * cb_len, 'N', (size_t)cb_len); }
    else { ok[i] = 1; std::memcpy(out + i * cb_len, res->data(), (size_t)cb_len); if (*res == key) ++w->perfect; else ++w->corrected; }
  }
  if (stats) { stats[0] = w->perfect; stats[1] = w->corrected; stats[2] = w->none; }
  return NB_OK;
}

// Formats the per-read rows of one batch into `out` (caller-allocated, out_cap bytes):
//   <features>\t<mates>\t<CB>\t<UMI>\t<r1 fwd>\t<r1 rev>\t<r2 fwd>\t<r2 rev>\n
// for every read i with reason == CALLED, a class whose feature string is non-NULL (class_feature[cls] == NULL = rejected
// by max_hits_to_report / discard_multi_hits) and cb_ok[i] != 0 (cb_ok may be NULL = all ok).
// Returns NB_ERR_OVERFLOW if out_cap is too small (nothing is lost: call again with a larger buffer).
int32_t nb_format_tenx_rows(int64_t n, const uint32_t* cls, const uint32_t* info, const uint64_t* cov, const char* const* class_feature,
                            int64_t n_classes, const char* cb, int32_t cb_len, const uint8_t* cb_ok, const char* umi, int32_t umi_len,
                            char* out, int64_t out_cap, int64_t* out_len, int64_t* n_rows) {
  if (!cls || !info || !cov || !class_feature || !cb || !umi || !out || !out_len || !n_rows) { nb::set_error("nb_format_tenx_rows: null argument"); return NB_ERR_INVALID; }
  int64_t pos = 0, rows = 0;
  auto put_u = [&](unsigned v) { char tmp[12]; int k = 0; do { tmp[k++] = (char)('0' + v % 10); v /= 10; } while (v); while (k) out[pos++] = tmp[--k]; };
  for (int64_t i = 0; i < n; ++i) {
    const uint32_t reason = (info[i] >> 16) & 0xFF;
    if (reason != NB_CALLED || cls[i] == NB_NO_CLASS || (int64_t)cls[i] >= n_classes) continue;
    const char* f = class_feature[cls[i]];
    if (!f || (cb_ok && !cb_ok[i])) continue;
    const size_t fl = std::strlen(f);
    if (pos + (int64_t)fl + cb_len + umi_len + 64 > out_cap) { nb::set_error("nb_format_tenx_rows: output buffer too small"); return NB_ERR_OVERFLOW; }
    std::memcpy(out + pos, f, fl); pos += (int64_t)fl; out[pos++] = '\t';
    put_u((info[i] >> 25) & 3u); out[pos++] = '\t';
    const size_t cl = strnlen(cb + i * cb_len, (size_t)cb_len), ul = strnlen(umi + i * umi_len, (size_t)umi_len);   // fields may be NUL padded (BAM tags)
    std::memcpy(out + pos, cb + i * cb_len, cl); pos += (int64_t)cl; out[pos++] = '\t';
    std::memcpy(out + pos, umi + i * umi_len, ul); pos += (int64_t)ul;
    for (int j = 0; j < 4; ++j) { out[pos++] = '\t'; put_u((unsigned)((cov[i] >> (16 * j)) & 0xFFFF)); }
    out[pos++] = '\n'; ++rows;
  }
  *out_len = pos; *n_rows = rows; return NB_OK;
}

}  // extern "C"
