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
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "nb_internal.h"

struct nb_whitelist {
  std::unordered_set<std::string> set;
  std::unordered_map<std::string, std::string> cache;   // raw -> corrected ("" = no correction)
  int64_t perfect = 0, corrected = 0, none = 0;
  // exact-match fast path: when every entry is ACGT of one length <= 32, the 2-bit codes sit in an open-addressing table;
  // exact matches (the bulk of real data) are order-independent, so they are resolved by worker threads and only the rest
  // walks the sequential first-sighting cache
  int threads = 1;
  int fast_len = 0;                 // 0 = fast path unavailable
  std::vector<uint64_t> table;      // code + 1, 0 = empty
  uint64_t mask = 0;
};

namespace {

inline bool encode_cb(const char* s, int len, uint64_t* out) {
  uint64_t v = 0;
  for (int i = 0; i < len; ++i) {
    uint64_t c;
    switch (s[i]) { case 'A': c = 0; break; case 'C': c = 1; break; case 'G': c = 2; break; case 'T': c = 3; break; default: return false; }
    v = (v << 2) | c;
  }
  *out = v; return true;
}

inline uint64_t mix_cb(uint64_t x) { x ^= x >> 31; x *= 0x9E3779B97F4A7C15ull; x ^= x >> 29; return x; }

void build_fast_table(nb_whitelist* w) {
  w->fast_len = 0;
  if (w->set.empty()) return;
  const size_t len = w->set.begin()->size();
  if (len < 1 || len > 32) return;
  std::vector<uint64_t> codes; codes.reserve(w->set.size());
  for (const auto& s : w->set) { uint64_t c; if (s.size() != len || !encode_cb(s.data(), (int)len, &c)) return; codes.push_back(c); }
  size_t cap = 64; while (cap < codes.size() * 4) cap <<= 1;
  w->table.assign(cap, 0); w->mask = cap - 1;
  for (uint64_t c : codes) { uint64_t p = mix_cb(c) & w->mask; while (w->table[p]) p = (p + 1) & w->mask; w->table[p] = c + 1; }
  w->fast_len = (int)len;
}

inline bool fast_contains(const nb_whitelist* w, uint64_t c) {
  uint64_t p = mix_cb(c) & w->mask;
  for (;;) { const uint64_t v = w->table[p]; if (!v) return false; if (v == c + 1) return true; p = (p + 1) & w->mask; }
}

}  // namespace

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
  build_fast_table(w);
  *out = w; return NB_OK;
}

int32_t nb_whitelist_set_threads(nb_whitelist* w, int32_t threads) {
  if (!w) { nb::set_error("nb_whitelist_set_threads: null handle"); return NB_ERR_INVALID; }
  w->threads = threads < 1 ? 1 : (threads > 256 ? 256 : threads); return NB_OK;
}

int32_t nb_whitelist_destroy(nb_whitelist* w) { delete w; return NB_OK; }

int64_t nb_whitelist_size(const nb_whitelist* w) { return w ? (int64_t)w->set.size() : 0; }

// n raw barcodes of fixed width cb_len (+ their quality characters) -> corrected barcodes (same layout) and ok[i] = 1,
// or ok[i] = 0 when the pair must be dropped. Exact matches do not depend on order and are resolved in parallel; everything
// else is sequential on purpose: the cache makes the first sighting decide.
int32_t nb_whitelist_correct(nb_whitelist* w, const char* raw, const char* qual, int64_t n, int32_t cb_len, char* out, uint8_t* ok,
                             int64_t* stats /* perfect, corrected, none: cumulative */) {
  if (!w || !raw || !qual || !out || !ok || cb_len < 1) { nb::set_error("nb_whitelist_correct: bad arguments"); return NB_ERR_INVALID; }
  std::vector<uint8_t> done;
  if (w->fast_len == cb_len && n >= 4096) {
    done.assign((size_t)n, 0);
    const int T = (int)std::min<int64_t>(w->threads, std::max<int64_t>(1, n / 16384));
    std::vector<int64_t> hits((size_t)T, 0);
    auto work = [&](int t) {
      int64_t h = 0;
      for (int64_t i = n * t / T, e = n * (t + 1) / T; i < e; ++i) {
        uint64_t c;
        if (encode_cb(raw + i * cb_len, cb_len, &c) && fast_contains(w, c)) { std::memcpy(out + i * cb_len, raw + i * cb_len, (size_t)cb_len); ok[i] = 1; done[(size_t)i] = 1; ++h; }
      }
      hits[(size_t)t] = h;
    };
    if (T <= 1) work(0);
    else { std::vector<std::thread> th; for (int t = 1; t < T; ++t) th.emplace_back(work, t); work(0); for (auto& x : th) x.join(); }
    for (int64_t h : hits) w->perfect += h;
  }
  std::string key, cand, best;
  for (int64_t i = 0; i < n; ++i) {
    if (!done.empty() && done[(size_t)i]) continue;
    key.assign(raw + i * cb_len, (size_t)cb_len);
    auto it = w->cache.find(key);
    const std::string* res;
    if (it != w->cache.end()) res = &it->second;
    else {
      std::string r;
      if (w->set.count(key)) r = key;
      else if (w->fast_len == cb_len) {
        // Hamming-1 candidates through the 2-bit table. A whitelist entry is ACGT only, so a raw barcode with one foreign
        // character (N) can only be repaired at that position, and one with two or more not at all. Candidates are compared
        // as the reference does: lowest quality at the differing base, ties to the lexicographically smallest string
        // (= the smallest code: A < C < G < T, first base most significant).
        const char* q = qual + i * cb_len;
        int foreign = 0, fpos = -1; uint64_t code = 0;
        for (int p = 0; p < cb_len; ++p) {
          uint64_t c;
          switch (key[(size_t)p]) { case 'A': c = 0; break; case 'C': c = 1; break; case 'G': c = 2; break; case 'T': c = 3; break; default: c = 0; ++foreign; fpos = p; }
          code = (code << 2) | c;
        }
        if (foreign <= 1) {
          int best_q = 1 << 30; uint64_t best_code = 0; bool any = false;
          for (int p = (foreign ? fpos : 0); p < (foreign ? fpos + 1 : cb_len); ++p) {
            const int sh = 2 * (cb_len - 1 - p);
            const uint64_t cur = (code >> sh) & 3u;
            for (uint64_t b = 0; b < 4; ++b) {
              if (!foreign && b == cur) continue;
              const uint64_t cand_code = (code & ~(3ull << sh)) | (b << sh);
              if (!fast_contains(w, cand_code)) continue;
              const int ql = (int)(unsigned char)q[p] - 33;
              if (!any || ql < best_q || (ql == best_q && cand_code < best_code)) { any = true; best_q = ql; best_code = cand_code; }
            }
          }
          if (any) { r.resize((size_t)cb_len); for (int p = 0; p < cb_len; ++p) r[(size_t)p] = "ACGT"[(best_code >> (2 * (cb_len - 1 - p))) & 3u]; }
        }
      } else {
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

// Class bitsets -> feature strings (docs/SPEC.md §5): for row r, the distinct feature names of its members in ascending
// sequence-index order of first appearance, comma-joined, to blob/offsets; the same names sorted, comma-joined (the form
// `report` groups by, /root/reference/nimble/__main__.py:248), to sorted_blob/sorted_offsets; reasons[r] = NB_CALLED, or
// NB_MAX_HITS / NB_MULTI_HITS when the feature count exceeds max_hits_to_report / discard_multi_hits (> 0).
// seq_feature[i] = index into feature_names of sequence i. Returns NB_ERR_OVERFLOW when a blob is too small.
int32_t nb_class_features(int64_t n_rows, int32_t words, const uint64_t* rows, int32_t n_seqs, const int32_t* seq_feature,
                          const char* const* feature_names, int32_t n_features, int32_t max_hits, int32_t multi_hits, char* blob,
                          int64_t blob_cap, int64_t* offsets, char* sorted_blob, int64_t sorted_cap, int64_t* sorted_offsets, int32_t* reasons) {
  if (n_rows < 0 || words < 1 || !rows || !seq_feature || !feature_names || !blob || !offsets || !sorted_blob || !sorted_offsets || !reasons ||
      (int64_t)words * 64 < n_seqs) { nb::set_error("nb_class_features: bad arguments"); return NB_ERR_INVALID; }
  std::vector<int32_t> feats; std::vector<uint32_t> stamp((size_t)n_features, 0); uint32_t tick = 0;
  std::vector<size_t> name_len((size_t)n_features);
  for (int32_t f = 0; f < n_features; ++f) name_len[(size_t)f] = std::strlen(feature_names[f]);
  int64_t pos = 0, spos = 0;
  auto emit = [&](char* dst, int64_t cap, int64_t& at) -> bool {
    for (size_t t = 0; t < feats.size(); ++t) {
      const size_t l = name_len[(size_t)feats[t]];
      if (at + (int64_t)l + 1 > cap) return false;
      if (t) dst[at++] = ',';
      std::memcpy(dst + at, feature_names[feats[t]], l); at += (int64_t)l;
    }
    return true;
  };
  for (int64_t r = 0; r < n_rows; ++r) {
    feats.clear(); ++tick;
    if (tick == 0) { std::fill(stamp.begin(), stamp.end(), 0u); tick = 1; }
    const uint64_t* row = rows + r * words;
    for (int32_t w = 0; w < words; ++w) {
      uint64_t m = row[w];
      while (m) {
        const int bit = __builtin_ctzll(m); m &= m - 1;
        const int64_t seq = (int64_t)w * 64 + bit;
        if (seq >= n_seqs) { nb::set_error("nb_class_features: class row has a bit beyond n_seqs"); return NB_ERR_INVALID; }
        const int32_t f = seq_feature[seq];
        if (f < 0 || f >= n_features) { nb::set_error("nb_class_features: feature id out of range"); return NB_ERR_INVALID; }
        if (stamp[(size_t)f] != tick) { stamp[(size_t)f] = tick; feats.push_back(f); }
      }
    }
    reasons[r] = (int64_t)feats.size() > max_hits ? NB_MAX_HITS : ((multi_hits > 0 && (int64_t)feats.size() > multi_hits) ? NB_MULTI_HITS : NB_CALLED);
    offsets[r] = pos; sorted_offsets[r] = spos;
    if (!emit(blob, blob_cap, pos)) { nb::set_error("nb_class_features: blob too small"); return NB_ERR_OVERFLOW; }
    std::sort(feats.begin(), feats.end(), [&](int32_t a, int32_t b) { return std::strcmp(feature_names[a], feature_names[b]) < 0; });
    if (!emit(sorted_blob, sorted_cap, spos)) { nb::set_error("nb_class_features: blob too small"); return NB_ERR_OVERFLOW; }
  }
  offsets[n_rows] = pos; sorted_offsets[n_rows] = spos;
  return NB_OK;
}

// gzip for the per-read TSV (the reference's aligner writes .gz tables for BAM input, validate_gzip at
// /root/reference/nimble/__main__.py:35-42): the text is cut into BGZF blocks (SAM spec §4.1: gzip members of <= 64 KiB
// with their size in a 'BC' extra field), each deflated on a worker thread. The concatenation is one valid gzip file
// (RFC 1952 §2.2) for gzip / pandas, and block-addressable for bgzip-aware tools and for the parallel reader of
// nb_report_tsv.
int32_t nb_gzip_members(const char* in, int64_t n, int32_t level, int32_t threads, char* out, int64_t out_cap, int64_t* out_len) {
  if ((!in && n) || n < 0 || !out || !out_len) { nb::set_error("nb_gzip_members: bad arguments"); return NB_ERR_INVALID; }
  constexpr int64_t PIECE = 0xff00;
  const int64_t pieces = (n + PIECE - 1) / PIECE;
  std::vector<std::vector<unsigned char>> comp((size_t)pieces);
  const int T = (int)std::min<int64_t>(threads < 1 ? 1 : threads, std::max<int64_t>(1, pieces / 16));
  std::vector<int> bad((size_t)T, 0);
  auto work = [&](int t) {
    z_stream z; std::memset(&z, 0, sizeof z);
    if (deflateInit2(&z, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { bad[(size_t)t] = 1; return; }
    for (int64_t p = pieces * t / T, e = pieces * (t + 1) / T; p < e; ++p) {
      const int64_t lo = p * PIECE, len = std::min(PIECE, n - lo);
      auto& v = comp[(size_t)p]; v.resize(18 + 65536 + 8);
      deflateReset(&z);
      z.next_in = (Bytef*)(in + lo); z.avail_in = (uInt)len; z.next_out = v.data() + 18; z.avail_out = 65536 - 26;
      if (deflate(&z, Z_FINISH) != Z_STREAM_END) { bad[(size_t)t] = 1; break; }
      const size_t clen = (size_t)z.total_out, total = 18 + clen + 8;
      static const unsigned char head[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
      std::memcpy(v.data(), head, 16);
      v[16] = (unsigned char)((total - 1) & 0xff); v[17] = (unsigned char)((total - 1) >> 8);
      const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), (const Bytef*)(in + lo), (uInt)len);
      unsigned char* t8 = v.data() + 18 + clen;
      for (int j = 0; j < 4; ++j) { t8[j] = (unsigned char)(crc >> (8 * j)); t8[4 + j] = (unsigned char)((uint32_t)len >> (8 * j)); }
      v.resize(total);
    }
    deflateEnd(&z);
  };
  if (T <= 1) work(0);
  else { std::vector<std::thread> th; for (int t = 1; t < T; ++t) th.emplace_back(work, t); work(0); for (auto& x : th) x.join(); }
  for (int b : bad) if (b) { nb::set_error("nb_gzip_members: deflate failed"); return NB_ERR_INVALID; }
  int64_t pos = 0;
  for (auto& v : comp) {
    if (pos + (int64_t)v.size() > out_cap) { nb::set_error("nb_gzip_members: output buffer too small"); return NB_ERR_OVERFLOW; }
    std::memcpy(out + pos, v.data(), v.size()); pos += (int64_t)v.size();
  }
  *out_len = pos; return NB_OK;
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
