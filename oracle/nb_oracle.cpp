// nb_oracle.cpp — CPU restatement of the pseudoalignment hot path. TEST INFRASTRUCTURE ONLY.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's library.
// The product (nimble_amd/) never links, imports or calls it.
//
// PARITY UNPINNED: the algorithm lives in a third-party Rust binary, BimberLab/nimble-aligner, release asset
// `Linux.out`, version `latest` (unpinned), which /root/reference downloads at run time
// (nimble/__main__.py:115-149, call site nimble/__main__.py:195) and which is absent from this pipeline.
// No reference test or golden vector covers per-read calls. This file therefore restates the
// published algorithm of the lineage the reference credits (README.md:37-40: rust-pseudoaligner's
// map_read_to_nodes walk over a coloured compacted de Bruijn graph) as written down in docs/SPEC.md, section by
// section, in the most literal sequential form: one k-mer step at a time, a hash map of k-mers, explicit sets.
// It deliberately shares no code, data layout or k-mer encoding with the HIP product (which uses a flat
// canonical-k-mer table and a bit-parallel formulation), so agreement between the two is a real check.
//
// Encoding here: first base of a k-mer is the MOST significant 2-bit group.

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {

struct KInfo {
  uint32_t colour = 0;   // id into Index::colours
  uint8_t lext = 0;      // bit b set: base b precedes this k-mer somewhere (SPEC §2)
  uint8_t rext = 0;      // bit b set: base b follows this k-mer somewhere
  uint8_t r_internal = 0;
  uint8_t l_internal = 0;
};

struct Index {
  int k = 20;
  int n_seqs = 0;
  int words = 0;                                   // ceil(n_seqs/64)
  std::unordered_map<uint64_t, KInfo> kmers;       // SPEC §2: S with colour/exts
  std::vector<std::vector<uint32_t>> colours;      // colour id -> sorted sequence indices
};

struct Config {           // mirrors nb_config field for field (include/nimble_amd.h); SPEC §4
  int32_t k;
  int32_t score_threshold;
  int32_t score_filter;
  int32_t num_mismatches;
  int32_t discard_multiple_matches;
  int32_t intersect_level;
  int32_t require_valid_pair;
  int32_t strand_filter;  // 0 unstranded, 1 fiveprime, 2 threeprime, 3 none
  double score_percent;
};

enum Reason : int {
  CALLED = 0, TOO_SHORT = 1, NO_HIT = 2, SCORE_THRESHOLD = 3, SCORE_PERCENT = 4, EMPTY_EQCLASS = 5,
  MULTIPLE_MATCH = 6, INVALID_PAIR = 7, STRAND = 8, SCORE_FILTER = 9
};

inline int code_of(char c) {
  switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 4;  // N
  }
}

inline int popcount8(uint8_t v) { return __builtin_popcount(v); }
inline int single_bit_index(uint8_t v) { return __builtin_ctz(v); }

struct Kmerizer {
  int k; uint64_t mask;
  explicit Kmerizer(int k_) : k(k_), mask(k_ == 32 ? ~0ull : ((1ull << (2 * k_)) - 1)) {}
  uint64_t succ(uint64_t x, int b) const { return ((x << 2) | (uint64_t)b) & mask; }
  uint64_t pred(uint64_t x, int a) const { return (x >> 2) | ((uint64_t)a << (2 * (k - 1))); }
};

// SPEC §2 — build S, colour, lext/rext, r_internal/l_internal from the library's sequences.
Index* build_index(int n_seqs, const char* const* seqs, const int64_t* lens, int k) {
  Index* ix = new Index();
  ix->k = k; ix->n_seqs = n_seqs; ix->words = (n_seqs + 63) / 64;
  Kmerizer kz(k);
  std::vector<std::pair<uint64_t, uint32_t>> occ;   // (k-mer, sequence)
  for (int s = 0; s < n_seqs; ++s) {
    const char* p = seqs[s]; int64_t L = lens[s];
    uint64_t x = 0; int run = 0;  // run = number of consecutive valid bases ending here
    for (int64_t i = 0; i < L; ++i) {
      int c = code_of(p[i]);
      if (c == 4) { run = 0; x = 0; continue; }
      uint64_t prev = x;
      x = kz.succ(x, c); ++run;
      if (run >= k) {
        occ.emplace_back(x, (uint32_t)s);
        KInfo& cur = ix->kmers[x];
        if (run >= k + 1) {                 // (K+1)-mer prev·c occurs: edge-centric extension
          int first_of_prev = code_of(p[i - k]);
          cur.lext |= (uint8_t)(1u << first_of_prev);
          ix->kmers[prev].rext |= (uint8_t)(1u << c);
        }
      }
    }
  }
  std::sort(occ.begin(), occ.end());
  occ.erase(std::unique(occ.begin(), occ.end()), occ.end());
  std::map<std::vector<uint32_t>, uint32_t> intern;
  size_t i = 0;
  while (i < occ.size()) {
    size_t j = i; std::vector<uint32_t> members;
    while (j < occ.size() && occ[j].first == occ[i].first) { members.push_back(occ[j].second); ++j; }
    auto it = intern.find(members);
    uint32_t id;
    if (it == intern.end()) { id = (uint32_t)ix->colours.size(); intern.emplace(members, id); ix->colours.push_back(members); }
    else id = it->second;
    ix->kmers[occ[i].first].colour = id;
    i = j;
  }
  // unitig interior flags (SPEC §2, local rule)
  for (auto& kv : ix->kmers) {
    uint64_t x = kv.first; KInfo& xi = kv.second;
    if (popcount8(xi.rext) == 1) {
      uint64_t y = kz.succ(x, single_bit_index(xi.rext));
      auto it = ix->kmers.find(y);
      if (it != ix->kmers.end() && popcount8(it->second.lext) == 1 && it->second.colour == xi.colour && y != x)
        xi.r_internal = 1;
    }
  }
  for (auto& kv : ix->kmers) {
    uint64_t y = kv.first; KInfo& yi = kv.second;
    if (popcount8(yi.lext) == 1) {
      uint64_t x = kz.pred(y, single_bit_index(yi.lext));
      auto it = ix->kmers.find(x);
      if (it != ix->kmers.end() && it->second.r_internal) yi.l_internal = 1;
    }
  }
  return ix;
}

struct Mapping {             // SPEC §3 result
  bool found = false; int none_reason = NO_HIT;
  int cov = 0; int mm = 0;
  std::vector<uint32_t> eq;  // sorted members
};

std::vector<uint32_t> intersect_sorted(const std::vector<uint32_t>& a, const std::vector<uint32_t>& b) {
  std::vector<uint32_t> o; std::set_intersection(a.begin(), a.end(), b.begin(), b.end(), std::back_inserter(o)); return o;
}
std::vector<uint32_t> union_sorted(const std::vector<uint32_t>& a, const std::vector<uint32_t>& b) {
  std::vector<uint32_t> o; std::set_union(a.begin(), a.end(), b.begin(), b.end(), std::back_inserter(o)); return o;
}

// SPEC §3, literally.
Mapping map_sequence(const Index& ix, const std::vector<int>& s, int M) {
  Mapping out; const int K = ix.k; const int L = (int)s.size(); Kmerizer kz(K);
  if (L < K) { out.none_reason = TOO_SHORT; return out; }
  const int last = L - K;
  auto kmer_at = [&](int p, uint64_t& x) -> bool {
    x = 0;
    for (int j = 0; j < K; ++j) { if (s[p + j] == 4) return false; x = (x << 2) | (uint64_t)s[p + j]; }
    return true;
  };
  auto seed = [&](int b, uint64_t& x) -> int {
    for (int p = b; p <= last; p += 3) { if (kmer_at(p, x) && ix.kmers.count(x)) return p; }
    return -1;
  };
  uint64_t x; int p = seed(0, x);
  if (p < 0) { out.none_reason = NO_HIT; return out; }
  std::vector<uint32_t> V; int cov = 0, mm = 0;
  if (p >= L / 5) {
    uint64_t y = x; int q = p - 1;
    while (q >= 0) {
      const KInfo& yi = ix.kmers.at(y); int c = s[q];
      if (c != 4 && ((yi.lext >> c) & 1)) y = kz.pred(y, c);
      else if (yi.l_internal && mm < M) { y = kz.pred(y, single_bit_index(yi.lext)); ++mm; }
      else break;
      ++cov; V.push_back(ix.kmers.at(y).colour); --q;
    }
  }
  for (;;) {
    cov += K; V.push_back(ix.kmers.at(x).colour); int b = p + K;
    while (b < L) {
      const KInfo& xi = ix.kmers.at(x); int c = s[b];
      if (c != 4 && ((xi.rext >> c) & 1)) x = kz.succ(x, c);
      else if (xi.r_internal && mm < M) { x = kz.succ(x, single_bit_index(xi.rext)); ++mm; }
      else break;
      ++cov; V.push_back(ix.kmers.at(x).colour); ++b;
    }
    if (b > last) break;
    p = seed(b, x);
    if (p < 0) break;
  }
  std::sort(V.begin(), V.end()); V.erase(std::unique(V.begin(), V.end()), V.end());
  std::vector<uint32_t> eq = ix.colours[V[0]];
  for (size_t i = 1; i < V.size(); ++i) eq = intersect_sorted(eq, ix.colours[V[i]]);
  out.found = true; out.cov = cov; out.mm = mm; out.eq = std::move(eq);
  return out;
}

struct Aln { bool valid = false; int reason = NO_HIT; int cov = 0; std::vector<uint32_t> eq; };

// SPEC §4 validity checks 1-4.
Aln validate(const Mapping& m, int L, const Config& cfg) {
  Aln a; a.cov = m.found ? m.cov : 0;
  if (!m.found) { a.reason = m.none_reason; return a; }
  if (m.cov < cfg.score_threshold) { a.reason = SCORE_THRESHOLD; return a; }
  if ((double)m.cov / (double)L < cfg.score_percent) { a.reason = SCORE_PERCENT; return a; }
  if (m.eq.empty()) { a.reason = EMPTY_EQCLASS; return a; }
  if (cfg.discard_multiple_matches && m.eq.size() != 1) { a.reason = MULTIPLE_MATCH; return a; }
  a.valid = true; a.reason = CALLED; a.eq = m.eq; return a;
}

std::vector<int> encode(const char* p, int64_t L) { std::vector<int> v((size_t)L); for (int64_t i = 0; i < L; ++i) v[(size_t)i] = code_of(p[i]); return v; }
std::vector<int> revcomp(const std::vector<int>& s) {
  std::vector<int> r(s.size());
  for (size_t i = 0; i < s.size(); ++i) { int c = s[s.size() - 1 - i]; r[i] = (c == 4) ? 4 : 3 - c; }
  return r;
}

struct Orient { bool valid = false; int reason = NO_HIT; int score = 0; int mates = 0; std::vector<uint32_t> set; };

Orient combine(const Aln* a1, const Aln* a2, const Config& cfg) {
  Orient o;
  if (!a2) { o.valid = a1->valid; o.reason = a1->reason; o.score = a1->cov; o.mates = a1->valid ? 1 : 0; o.set = a1->eq; if (!o.valid) o.score = 0; return o; }
  if (a1->valid && a2->valid) {
    std::vector<uint32_t> is = intersect_sorted(a1->eq, a2->eq);
    if (cfg.intersect_level == 0) o.set = union_sorted(a1->eq, a2->eq);
    else if (cfg.intersect_level == 1) o.set = is.empty() ? union_sorted(a1->eq, a2->eq) : is;
    else { if (is.empty()) { o.reason = EMPTY_EQCLASS; return o; } o.set = is; }
    o.valid = true; o.reason = CALLED; o.score = a1->cov + a2->cov; o.mates = 2; return o;
  }
  if (a1->valid || a2->valid) {
    if (cfg.require_valid_pair) { o.reason = INVALID_PAIR; return o; }
    const Aln* a = a1->valid ? a1 : a2;
    o.valid = true; o.reason = CALLED; o.score = a->cov; o.mates = 1; o.set = a->eq; return o;
  }
  o.reason = std::max(a1->reason, a2->reason); return o;
}

struct TemplateResult {
  int reason; int score; int orientation; int mates; int cov[4];  // r1 fwd, r1 rev, r2 fwd, r2 rev
  std::vector<uint32_t> set;
};

// SPEC §4 — orientations, mates, strand filter, score_filter.
TemplateResult align_template(const Index& ix, const Config& cfg, const char* r1, int64_t l1, const char* r2, int64_t l2) {
  TemplateResult t{}; const bool paired = (r2 != nullptr);
  std::vector<int> s1 = encode(r1, l1), s1rc = revcomp(s1), s2, s2rc;
  if (paired) { s2 = encode(r2, l2); s2rc = revcomp(s2); }
  Mapping m1f = map_sequence(ix, s1, cfg.num_mismatches), m1r = map_sequence(ix, s1rc, cfg.num_mismatches), m2f, m2r;
  Aln a1f = validate(m1f, (int)l1, cfg), a1r = validate(m1r, (int)l1, cfg), a2f, a2r;
  t.cov[0] = m1f.found ? m1f.cov : 0; t.cov[1] = m1r.found ? m1r.cov : 0; t.cov[2] = t.cov[3] = 0;
  if (paired) {
    m2f = map_sequence(ix, s2, cfg.num_mismatches); m2r = map_sequence(ix, s2rc, cfg.num_mismatches);
    a2f = validate(m2f, (int)l2, cfg); a2r = validate(m2r, (int)l2, cfg);
    t.cov[2] = m2f.found ? m2f.cov : 0; t.cov[3] = m2r.found ? m2r.cov : 0;
  }
  Orient F = combine(&a1f, paired ? &a2r : nullptr, cfg);   // F = {map(r1), map(rc(r2))}
  Orient R = combine(&a1r, paired ? &a2f : nullptr, cfg);   // R = {map(rc(r1)), map(r2)}
  const bool allowF = cfg.strand_filter != 2, allowR = cfg.strand_filter != 1;
  if (F.valid && !allowF) { F.valid = false; F.reason = STRAND; }
  if (R.valid && !allowR) { R.valid = false; R.reason = STRAND; }
  const Orient* pick = nullptr; int orient = 0;
  if (F.valid && R.valid) { if (R.score > F.score) { pick = &R; orient = 1; } else { pick = &F; orient = 0; } }
  else if (F.valid) { pick = &F; orient = 0; }
  else if (R.valid) { pick = &R; orient = 1; }
  if (!pick) { t.reason = std::max(F.reason, R.reason); return t; }
  t.score = pick->score; t.orientation = orient; t.mates = pick->mates;
  if (pick->score < cfg.score_filter) { t.reason = SCORE_FILTER; return t; }
  t.reason = CALLED; t.set = pick->set; return t;
}

}  // namespace

extern "C" {

void* nbo_index_build(int32_t n_seqs, const char* const* seqs, const int64_t* lens, int32_t k) {
  if (k < 2 || k > 31 || n_seqs <= 0) return nullptr;
  return build_index(n_seqs, seqs, lens, k);
}
void nbo_index_destroy(void* h) { delete static_cast<Index*>(h); }
int64_t nbo_index_num_kmers(void* h) { return (int64_t)static_cast<Index*>(h)->kmers.size(); }
int64_t nbo_index_num_colours(void* h) { return (int64_t)static_cast<Index*>(h)->colours.size(); }
int32_t nbo_index_words(void* h) { return static_cast<Index*>(h)->words; }

// Membership query used by index-parity tests: returns 1 and fills fields if the (ASCII) k-mer is in S.
int32_t nbo_index_query(void* h, const char* kmer_ascii, uint8_t* lext, uint8_t* rext, uint8_t* r_int, uint8_t* l_int,
                        uint64_t* colour_bits /* words */) {
  Index* ix = static_cast<Index*>(h); uint64_t x = 0;
  for (int j = 0; j < ix->k; ++j) { int c = code_of(kmer_ascii[j]); if (c == 4) return 0; x = (x << 2) | (uint64_t)c; }
  auto it = ix->kmers.find(x); if (it == ix->kmers.end()) return 0;
  *lext = it->second.lext; *rext = it->second.rext; *r_int = it->second.r_internal; *l_int = it->second.l_internal;
  for (int w = 0; w < ix->words; ++w) colour_bits[w] = 0;
  for (uint32_t m : ix->colours[it->second.colour]) colour_bits[m >> 6] |= 1ull << (m & 63);
  return 1;
}

// Align n templates. reads1/reads2: concatenated ASCII with offsets[n+1] (reads2/off2 may be NULL = single-end).
// Outputs (caller-allocated): set_bits[n*words], score[n], reason[n], orientation[n], mates[n], cov[n*4].
int32_t nbo_align(void* h, const Config* cfg, int64_t n, const char* reads1, const int64_t* off1, const char* reads2,
                  const int64_t* off2, uint64_t* set_bits, int32_t* score, uint8_t* reason, uint8_t* orientation,
                  uint8_t* mates, int32_t* cov, int32_t n_threads) {
  Index* ix = static_cast<Index*>(h);
  if (!ix || cfg->k != ix->k) return -1;
  if (n_threads < 1) n_threads = 1;
  std::atomic<int64_t> next{0};
  auto work = [&]() {
    for (;;) {
      int64_t lo = next.fetch_add(256); if (lo >= n) break;
      int64_t hi = std::min(n, lo + 256);
      for (int64_t i = lo; i < hi; ++i) {
        const char* r2 = reads2 ? reads2 + off2[i] : nullptr; int64_t l2 = reads2 ? off2[i + 1] - off2[i] : 0;
        TemplateResult t = align_template(*ix, *cfg, reads1 + off1[i], off1[i + 1] - off1[i], r2, l2);
        score[i] = t.score; reason[i] = (uint8_t)t.reason; orientation[i] = (uint8_t)t.orientation; mates[i] = (uint8_t)t.mates;
        for (int j = 0; j < 4; ++j) cov[i * 4 + j] = t.cov[j];
        uint64_t* row = set_bits + i * ix->words;
        for (int w = 0; w < ix->words; ++w) row[w] = 0;
        for (uint32_t m : t.set) row[m >> 6] |= 1ull << (m & 63);
      }
    }
  };
  std::vector<std::thread> th;
  for (int t = 1; t < n_threads; ++t) th.emplace_back(work);
  work();
  for (auto& x : th) x.join();
  return 0;
}

}  // extern "C"
