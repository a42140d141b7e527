// nb_index.cpp — host-side build of the coloured k-mer index (product code; pure C++, no GPU needed).
//
// Replaces the graph build the Rust binary performs after parsing "-r LIB" (/root/reference/nimble/__main__.py:182-189)
// from the library's `sequence` column (/root/reference/nimble/types.py:29-32). Semantics: docs/SPEC.md §2.
// Sort-based (no hash maps): every stranded k-mer occurrence is emitted with its one-base left/right context, the
// array is sorted, and colours / extension masks fall out of a single grouped pass. The result is the flat
// canonical-k-mer table the kernels probe plus the colour bitset rows.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <unordered_map>

#include "nb_internal.h"

namespace nb {

namespace {

struct Occ {
  uint64_t kmer;
  uint32_t seq;
  uint8_t lext, rext;  // single-bit masks (or 0) contributed by this occurrence
};

struct KRec {
  uint64_t kmer;
  uint32_t colour;  // temp id, later remapped
  uint8_t lext, rext, r_int, l_int;
};

inline int code_of(char c) {
  switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 4;
  }
}

inline uint64_t mix64(uint64_t z) {
  z ^= z >> 33; z *= 0xff51afd7ed558ccdull; z ^= z >> 33; z *= 0xc4ceb9fe1a85ec53ull; z ^= z >> 33; return z;
}

int index_threads() {
  int t = (int)std::thread::hardware_concurrency();
  if (const char* e = std::getenv("NB_INDEX_THREADS")) t = std::atoi(e);
  return std::max(1, std::min(t, 16));
}

template <class F>
void parallel_ranges(size_t n, int threads, F&& f) {   // f(lo, hi) over [0, n) cut into `threads` ranges
  const int T = (int)std::min<size_t>((size_t)threads, std::max<size_t>(1, n >> 16));
  if (T <= 1) { f((size_t)0, n); return; }
  std::vector<std::thread> th;
  for (int t = 1; t < T; ++t) th.emplace_back([&f, n, t, T] { f(n * (size_t)t / (size_t)T, n * (size_t)(t + 1) / (size_t)T); });
  f((size_t)0, n / (size_t)T);
  for (auto& x : th) x.join();
}

const KRec* find_rec(const std::vector<KRec>& recs, uint64_t x) {
  auto it = std::lower_bound(recs.begin(), recs.end(), x, [](const KRec& r, uint64_t v) { return r.kmer < v; });
  return (it != recs.end() && it->kmer == x) ? &*it : nullptr;
}

}  // namespace

int build_host_index(int n_seqs, const char* const* seqs, const int64_t* lens, int k, HostIndex** out) {
  if (n_seqs <= 0 || !seqs || !lens || !out) { set_error("nb_index_build: null or empty input"); return NB_ERR_INVALID; }
  if (k < 2 || k > NB_MAX_K) { set_error("nb_index_build: k must be in 2..31"); return NB_ERR_INVALID; }
  const uint64_t mask = kmer_mask(k);
  const int top = 2 * (k - 1);
  const bool dbg = std::getenv("NB_INDEX_DEBUG") != nullptr;             // stage clock on stderr
  auto tick = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (dbg) { const auto now = std::chrono::steady_clock::now(); std::fprintf(stderr, "[nb_index] %-28s %.2f s\n", what, std::chrono::duration<double>(now - tick).count()); tick = now; }
  };

  // 1. occurrences
  int64_t total = 0;
  for (int s = 0; s < n_seqs; ++s) total += std::max<int64_t>(0, lens[s] - k + 1);
  std::vector<Occ> occ; occ.reserve((size_t)total);
  for (int s = 0; s < n_seqs; ++s) {
    const char* p = seqs[s]; const int64_t L = lens[s];
    uint64_t x = 0; int64_t run = 0;
    for (int64_t i = 0; i < L; ++i) {
      const int c = code_of(p[i]);
      if (c == 4) { run = 0; x = 0; continue; }
      x = (x >> 2) | ((uint64_t)c << top); ++run;
      if (run >= k) {
        Occ o; o.kmer = x & mask; o.seq = (uint32_t)s; o.lext = 0; o.rext = 0;
        if (run >= k + 1) o.lext = (uint8_t)(1u << code_of(p[i - k]));           // (K+1)-mer a·x occurs
        if (i + 1 < L) { const int n = code_of(p[i + 1]); if (n != 4) o.rext = (uint8_t)(1u << n); }  // x·b occurs
        occ.push_back(o);
      }
    }
  }
  if (occ.empty()) { set_error("nb_index_build: library has no k-mer of length k"); return NB_ERR_INVALID; }
  const int threads = index_threads();
  lap("occurrences");
  parallel_sort(occ, threads, [](const Occ& a, const Occ& b) { return a.kmer != b.kmer ? a.kmer < b.kmer : a.seq < b.seq; });

  lap("occurrences sorted");
  // 2. group by k-mer: merged exts, member list; intern member lists into temp colour ids
  std::vector<KRec> recs;
  std::vector<uint32_t> pool;                 // concatenated member lists of distinct colours
  std::vector<uint64_t> col_off{0};           // offsets into pool
  std::unordered_multimap<uint64_t, uint32_t> by_hash;
  std::vector<uint32_t> members;
  for (size_t i = 0; i < occ.size();) {
    size_t j = i; uint8_t le = 0, re = 0; members.clear();
    while (j < occ.size() && occ[j].kmer == occ[i].kmer) {
      le |= occ[j].lext; re |= occ[j].rext;
      if (members.empty() || members.back() != occ[j].seq) members.push_back(occ[j].seq);
      ++j;
    }
    uint64_t h = 0x1234567ull + members.size();
    for (uint32_t m : members) h = mix64(h ^ (uint64_t)m);
    uint32_t id = 0xFFFFFFFFu;
    auto range = by_hash.equal_range(h);
    for (auto it = range.first; it != range.second; ++it) {
      const uint32_t c = it->second; const uint64_t b = col_off[c], e = col_off[c + 1];
      if (e - b == members.size() && std::equal(members.begin(), members.end(), pool.begin() + (std::ptrdiff_t)b)) { id = c; break; }
    }
    if (id == 0xFFFFFFFFu) {
      id = (uint32_t)(col_off.size() - 1);
      pool.insert(pool.end(), members.begin(), members.end());
      col_off.push_back(pool.size());
      by_hash.emplace(h, id);
    }
    recs.push_back(KRec{occ[i].kmer, id, le, re, 0, 0});
    i = j;
  }
  std::vector<Occ>().swap(occ);
  lap("grouped by k-mer, classes");
  const uint32_t n_col = (uint32_t)(col_off.size() - 1);
  if (n_col > MAX_COLOURS) { set_error("nb_index_build: more than 2^22-1 equivalence classes"); return NB_ERR_UNSUPPORTED; }

  // 3. final colour ids: ascending set size, ties by member list (lexicographic): a canonical numbering that depends on
  //    the library only (not on k-mer encoding or hash order).
  std::vector<uint32_t> order(n_col); std::iota(order.begin(), order.end(), 0u);
  std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
    const uint64_t la = col_off[a + 1] - col_off[a], lb = col_off[b + 1] - col_off[b];
    if (la != lb) return la < lb;
    return std::lexicographical_compare(pool.begin() + (std::ptrdiff_t)col_off[a], pool.begin() + (std::ptrdiff_t)col_off[a + 1],
                                        pool.begin() + (std::ptrdiff_t)col_off[b], pool.begin() + (std::ptrdiff_t)col_off[b + 1]);
  });
  std::vector<uint32_t> remap(n_col);
  for (uint32_t r = 0; r < n_col; ++r) remap[order[r]] = r;
  for (auto& r : recs) r.colour = remap[r.colour];

  HostIndex* ix = new HostIndex();
  ix->k = k; ix->n_seqs = n_seqs; ix->words = (n_seqs + 63) / 64; ix->n_kmers = (int64_t)recs.size();
  ix->eq_rows.assign((size_t)n_col * (size_t)ix->words, 0ull);
  ix->eq_sizes.resize(n_col);
  for (uint32_t c = 0; c < n_col; ++c) {
    const uint32_t id = remap[c];
    ix->eq_sizes[id] = (uint32_t)(col_off[c + 1] - col_off[c]);
    uint64_t* row = ix->eq_rows.data() + (size_t)id * (size_t)ix->words;
    for (uint64_t t = col_off[c]; t < col_off[c + 1]; ++t) row[pool[t] >> 6] |= 1ull << (pool[t] & 63);
  }

  // 3a. internal sequence numbering (nb_internal.h, ColDesc): connected components of "two sequences share a k-mer" (union-find
  //     over the colours' member lists), in order of their smallest member; a component of <= 64 sequences never straddles a word
  //     boundary, a larger one starts at a boundary. Every class then lies inside its component's words.
  {
    std::vector<uint32_t> parent((size_t)n_seqs); std::iota(parent.begin(), parent.end(), 0u);
    auto find = [&](uint32_t x) { while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; } return x; };
    for (uint32_t c = 0; c < n_col; ++c) {
      const uint32_t r0 = find(pool[col_off[c]]);
      for (uint64_t t = col_off[c] + 1; t < col_off[c + 1]; ++t) { const uint32_t r = find(pool[t]); if (r != r0) parent[r] = r0; }
    }
    std::vector<std::vector<uint32_t>> comps; std::vector<int32_t> comp_of((size_t)n_seqs, -1);
    for (int sidx = 0; sidx < n_seqs; ++sidx) {
      const uint32_t r = find((uint32_t)sidx);
      if (comp_of[r] < 0) { comp_of[r] = (int32_t)comps.size(); comps.emplace_back(); }
      comps[(size_t)comp_of[r]].push_back((uint32_t)sidx);
    }
    ix->perm.assign((size_t)n_seqs, -1);
    ix->comp_w0.assign((size_t)n_seqs, 0u);
    ix->comp_nw.assign((size_t)n_seqs, 1u);
    int64_t bit = 0;
    for (const auto& comp : comps) {
      const int64_t sz = (int64_t)comp.size();
      if (sz > 64 || (bit & 63) + sz > 64) bit = (bit + 63) & ~63ll;
      const uint32_t w0c = (uint32_t)(bit >> 6);
      const int cw = (int)((bit + sz - 1) >> 6) - (int)w0c + 1;
      ix->max_comp_words = std::max(ix->max_comp_words, cw);
      for (uint32_t sidx : comp) { ix->perm[sidx] = (int32_t)bit++; ix->comp_w0[sidx] = w0c; ix->comp_nw[sidx] = (uint32_t)cw; }
    }
    ix->words_dev = (int)((bit + 63) >> 6);
    if (ix->words_dev < 1) ix->words_dev = 1;
    ix->inv_perm.assign((size_t)ix->words_dev * 64, -1);
    for (int sidx = 0; sidx < n_seqs; ++sidx) ix->inv_perm[(size_t)ix->perm[(size_t)sidx]] = sidx;
  }

  // 3b. device form of the colours: span descriptors in internal numbering (first two words inline) + packed spans for the wider ones
  ix->desc.resize(n_col);
  ix->packed.assign(1, 0ull);
  {
    std::vector<uint64_t> span;
    for (uint32_t c = 0; c < n_col; ++c) {
      const uint32_t id = remap[c];
      uint32_t lo = 0xFFFFFFFFu, hi = 0;
      for (uint64_t t = col_off[c]; t < col_off[c + 1]; ++t) { const uint32_t w = (uint32_t)ix->perm[pool[t]] >> 6; lo = std::min(lo, w); hi = std::max(hi, w); }
      span.assign((size_t)(hi - lo + 1), 0ull);
      for (uint64_t t = col_off[c]; t < col_off[c + 1]; ++t) { const uint32_t b = (uint32_t)ix->perm[pool[t]]; span[(b >> 6) - lo] |= 1ull << (b & 63); }
      ColDesc d{}; d.w0 = lo; d.nw = hi - lo + 1; d.size = ix->eq_sizes[id]; d.m0 = span[0]; d.m1 = d.nw > 1 ? span[1] : 0ull; d.off = 0;
      ix->max_desc_nw = std::max(ix->max_desc_nw, (int)d.nw);
      if (d.nw > 2) {
        if (ix->packed.size() + d.nw > 0xFFFFFFFFull) { delete ix; set_error("nb_index_build: packed colour spans exceed 2^32 words"); return NB_ERR_UNSUPPORTED; }
        d.off = (uint32_t)ix->packed.size();
        ix->packed.insert(ix->packed.end(), span.begin(), span.end());
      }
      ix->desc[id] = d;
    }
  }

  // 4. unitig-interior flags (SPEC §2, local rule). recs is sorted by k-mer.
  parallel_ranges(recs.size(), threads, [&](size_t lo, size_t hi) {
    for (size_t i = lo; i < hi; ++i) { KRec& r = recs[i];
      if (__builtin_popcount(r.rext) == 1) {
        const uint64_t y = (r.kmer >> 2) | ((uint64_t)__builtin_ctz(r.rext) << top);
        const KRec* yr = find_rec(recs, y);
        if (yr && __builtin_popcount(yr->lext) == 1 && yr->colour == r.colour && y != r.kmer) r.r_int = 1;
      } }
  });
  parallel_ranges(recs.size(), threads, [&](size_t lo, size_t hi) {   // (reads r_int of OTHER records: all of them are final by now)
    for (size_t i = lo; i < hi; ++i) { KRec& r = recs[i];
      if (__builtin_popcount(r.lext) == 1) {
        const uint64_t x = ((r.kmer << 2) | (uint64_t)__builtin_ctz(r.lext)) & mask;
        const KRec* xr = find_rec(recs, x);
        if (xr && xr->r_int) r.l_int = 1;
      } }
  });

  // 5. canonical table: 32-byte buckets of two slots, load <= TABLE_MAX_LOAD (nb_internal.h).
  struct Canon { uint64_t key; uint32_t pay[2]; };
  std::vector<Canon> canons; canons.reserve(recs.size());
  {
    std::vector<std::pair<uint64_t, std::pair<int, uint32_t>>> tmp; tmp.reserve(recs.size());
    for (const auto& r : recs) {
      const uint64_t y = rc_kmer_host(r.kmer, k), canon = std::min(r.kmer, y);
      const uint32_t pay = make_payload(r.colour, r.lext, r.rext, r.r_int, r.l_int);
      tmp.push_back({canon, {(r.kmer != canon) ? 1 : 0, pay}});
      if (r.kmer == y) tmp.push_back({canon, {1, pay}});   // palindrome: both orientations are this k-mer
    }
    parallel_sort(tmp, threads, [](const std::pair<uint64_t, std::pair<int, uint32_t>>& a, const std::pair<uint64_t, std::pair<int, uint32_t>>& b) { return a < b; });
    for (size_t i = 0; i < tmp.size();) {
      Canon c{tmp[i].first, {PAY_ABSENT, PAY_ABSENT}};
      size_t j = i;
      while (j < tmp.size() && tmp[j].first == c.key) { c.pay[tmp[j].second.first] = tmp[j].second.second; ++j; }
      canons.push_back(c); i = j;
    }
  }
  const int64_t n_canon = (int64_t)canons.size();
  // Table kind. The compact cuckoo table reads TWO random 16-byte buckets per lookup: that wins while the table stays (nearly) resident in
  // the 4 MiB L2s (measured on MI355X, profiles/r02_tuning_notes.md: c2 2.9 MB 15.4 vs 16.4 ms, c3 4.7 MB 107 vs 124 ms) and loses once
  // the lookups go to the fabric, which serves a fixed ~55-60 G requests/s whatever the table size (c5: 26 MB, 500 vs 314 ms with the
  // one-line-per-lookup wide table). NB_TABLE=wide / compact forces a kind (tests, A/B runs).
  const char* force = std::getenv("NB_TABLE");
  int64_t n_entries_est = 0;
  for (const Canon& c : canons) n_entries_est += (c.pay[0] != PAY_ABSENT) + (c.pay[1] != PAY_ABSENT);
  const bool fits = (double)n_entries_est / COMPACT_LOAD * 8.0 <= (double)COMPACT_MAX_BYTES;
  const bool want_compact = force ? std::string(force) == "compact" || (std::string(force) != "wide" && fits) : fits;
  if (k <= COMPACT_MAX_K && want_compact) {
    // 5a. compact table: (2,2) partial-key cuckoo, one entry per (canonical k-mer, strand) present (nb_internal.h)
    int64_t n_entries = 0;
    for (const Canon& c : canons) n_entries += (c.pay[0] != PAY_ABSENT) + (c.pay[1] != PAY_ABSENT);
    const uint32_t hb = 2 * k > 32 ? (uint32_t)(2 * k - 32) : 0u;
    const uint64_t nb_min = std::max<uint64_t>(256, 1ull << (hb + 2));
    static const uint32_t MULS[4][2] = {{0x9E3779B1u, 0x85EBCA77u}, {0xC2B2AE3Du, 0x27D4EB2Fu}, {0x165667B1u, 0x9E3779B9u}, {0xD3A2646Du, 0xFD7046C5u}};
    double load = COMPACT_LOAD;
    bool done = false;
    // (a canonical k-mer present on BOTH strands needs two of the four slots of its bucket pair: a library that holds sequences and their
    //  reverse complements only fills to about half the usual load, hence the long tail of attempts)
    const char* att_env = std::getenv("NB_COMPACT_ATTEMPTS");           // (tests: 0 = "placement failed", the wide table takes over)
    const int max_attempts = att_env ? std::atoi(att_env) : 40;
    for (int attempt = 0; attempt < max_attempts && !done; ++attempt, load *= (attempt % 2 == 0 ? 0.92 : 1.0)) {
      CHash h{}; h.a = MULS[attempt % 4][0]; h.c = MULS[attempt % 4][1]; h.hb = hb; h.himask = hb ? ((1u << hb) - 1u) : 0u;
      uint64_t nb64 = std::max<uint64_t>(nb_min, (uint64_t)((double)n_entries / (2.0 * load)) + 1);
      if (nb64 > 0x7FFFFFFFull) break;
      h.nb = (uint32_t)nb64; h.nbm1 = h.nb - 1;
      std::vector<CBucket> tab((size_t)nb64, CBucket{{{CKEY_EMPTY, PAY_ABSENT}, {CKEY_EMPTY, PAY_ABSENT}}});
      uint64_t rng = 0x9E3779B97F4A7C15ull ^ (uint64_t)attempt;
      bool ok = true;
      auto place = [&](uint32_t key, uint32_t pay, uint32_t b) -> bool {      // random-walk insertion; key carries which/strand bits
        for (int kick = 0; kick < 4000; ++kick) {
          CBucket& bk = tab[b];
          for (int j = 0; j < 2; ++j) if (bk.e[j].key == CKEY_EMPTY && bk.e[j].pay == PAY_ABSENT) { bk.e[j] = CEntry{key, pay}; return true; }
          if (kick == 0 && !(key & 2u)) {                                      // first try the alternate bucket before evicting anyone
            const uint32_t d = compact_alt_delta(key & ~3u, h); uint32_t b2 = b + d; if (b2 >= h.nb) b2 -= h.nb;
            CBucket& alt = tab[b2];
            for (int j = 0; j < 2; ++j) if (alt.e[j].key == CKEY_EMPTY && alt.e[j].pay == PAY_ABSENT) { alt.e[j] = CEntry{key | 2u, pay}; return true; }
          }
          rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17;
          const int j = (int)(rng >> 33) & 1;
          std::swap(key, bk.e[j].key); std::swap(pay, bk.e[j].pay);            // evict entry j; the evicted one moves to ITS other bucket
          const uint32_t d = compact_alt_delta(key & ~3u, h);
          if (key & 2u) { b = b >= d ? b - d : b + h.nb - d; key &= ~2u; }     // was in its alternate bucket: back home
          else { b += d; if (b >= h.nb) b -= h.nb; key |= 2u; }
        }
        return false;
      };
      for (const Canon& c : canons) {
        const CProbe p = compact_probe(c.key, h);
        for (int s = 0; s < 2 && ok; ++s) if (c.pay[s] != PAY_ABSENT) {
          const uint32_t key = p.want | (uint32_t)s;
          if (key == CKEY_EMPTY || (key | 2u) == CKEY_EMPTY) { ok = false; break; }   // would read as an empty entry: other multipliers
          ok = place(key, c.pay[s], p.b1);
          if (!ok && std::getenv("NB_INDEX_DEBUG")) std::fprintf(stderr, "[index]   placement failed for canon %llx strand %d b1 %u b2 %u want %08x pay %08x\n", (unsigned long long)c.key, s, p.b1, p.b2, p.want, c.pay[s]);
        }
        if (!ok) break;
      }
      if (!ok && std::getenv("NB_INDEX_DEBUG")) std::fprintf(stderr, "[index] compact table attempt %d failed: %lld entries, %llu buckets, load %.3f\n", attempt, (long long)n_entries, (unsigned long long)nb64, load);
      if (ok) { ix->compact = true; ix->chash = h; ix->ctable.swap(tab); done = true; }
    }
    // (no placement after all attempts — e.g. k = 23 with a small library, where the table stays at its minimum size and a bucket index is
    //  almost all "high bits of the k-mer": allelic variants of one k-mer pile into the same two buckets — is no error: the wide table below
    //  holds any library)
    ix->hash = KmerHash{0, 0, 0};
  }
  if (!ix->compact) {
  uint64_t nb64 = (uint64_t)((double)n_canon / (SLOTS_PER_BUCKET * TABLE_MAX_LOAD)) + 1;
  if (nb64 < 256) nb64 = 256;
  if (nb64 > 0x7FFFFFFFull) { delete ix; set_error("nb_index_build: k-mer table would exceed 2^31 buckets"); return NB_ERR_UNSUPPORTED; }
  ix->hash.mf = 0x9E3779B1u; ix->hash.m1 = 0x85EBCA77u; ix->hash.n_buckets = (uint32_t)nb64;
  Bucket empty; for (int j = 0; j < SLOTS_PER_BUCKET; ++j) { empty.slot[j].key = KEY_EMPTY; empty.slot[j].pay[0] = empty.slot[j].pay[1] = PAY_ABSENT; }
  ix->buckets.assign((size_t)nb64, empty);
  for (const Canon& c : canons) {
    uint32_t b = kmer_bucket(c.key, ix->hash);
    for (;;) {
      Bucket& bk = ix->buckets[b];
      int j = 0; while (j < SLOTS_PER_BUCKET && (bk.slot[j].key & KEY_MASK) != KEY_EMPTY) ++j;
      if (j < SLOTS_PER_BUCKET) { bk.slot[j].key = (bk.slot[j].key & KEY_OVERFLOW) | c.key; bk.slot[j].pay[0] = c.pay[0]; bk.slot[j].pay[1] = c.pay[1]; break; }
      bk.slot[0].key |= KEY_OVERFLOW;                         // full: lookups of keys homed at or before b continue
      b = (b + 1 == ix->hash.n_buckets) ? 0 : b + 1;
    }
  }
  }

  // 5b. perfect-hash table (nb_internal.h): hash and displace, one-byte pilots, groups keyed by rem. Two sizes of first level:
  //     small enough for LDS (<= PH_MAX_PILOT_BYTES: the lane-per-read kernels and the anchor kernel stage it per workgroup), or one pilot
  //     per ~3 k-mers in global memory for the libraries whose table is far beyond the L2s anyway (anchor kernel only: one more L2-resident
  //     load per lookup, but ~25 lookups per read instead of one per position).
  const char* ph_env = std::getenv("NB_PH");
  lap("layouts, k-mer table");
  if (dbg) std::fprintf(stderr, "[nb_index] k=%d max_comp_words=%d max_desc_nw=%d n_canon=%lld\n", k, ix->max_comp_words, ix->max_desc_nw, (long long)n_canon);
  if (k <= COMPACT_MAX_K && !(ph_env && ph_env[0] == '0')) {
    static const uint32_t PMULS[2] = {0x9E3779B1u, 0x85EBCA77u};
    const uint32_t hb = 2 * k > 32 ? (uint32_t)(2 * k - 32) : 0u;
    const uint64_t nb_min = std::max<uint64_t>(256, 1ull << (hb + 2));
    const uint32_t mul_a = ix->compact ? ix->chash.a : PMULS[0], mul_c = ix->compact ? ix->chash.c : PMULS[1];
    const uint32_t himask = hb ? ((1u << hb) - 1u) : 0u;
    const int64_t n = n_canon;
    int64_t pilot_cap = PH_MAX_PILOT_BYTES;
    if (const char* e = std::getenv("NB_PH_PILOT_KB")) { const int64_t v = std::atoll(e); if (v >= 1 && v <= 150) pilot_cap = v << 10; }
    // groups: as many as the LDS budget of the first level allows, but not more than one per 2.5 k-mers (smaller groups fill the
    // table higher: a group of s k-mers finds a free pilot among 256 only while (1 - fill)^s stays above a few per cent)
    int64_t G = std::max<int64_t>(1, std::min<int64_t>(pilot_cap, (int64_t)((double)n / 2.5)));
    int64_t table_cap = PH_MAX_TABLE_BYTES;
    // too many k-mers for an LDS first level (or classes the lane kernel cannot hold): first level in global memory, one pilot per 3 k-mers.
    // Also the second try when the LDS-sized first level finds no pilots: groups of 7+ k-mers (libraries of ~2 000 .. 4 000 sequences) do not
    // place with one-byte pilots at any useful load — such a library used to end up without perfect hash, i.e. on the wave-per-read kernels.
    const bool large_first = (double)n / (double)G > 16.0 || ix->max_desc_nw != 1;
    for (int variant = large_first ? 1 : 0; variant < 2 && !ix->has_ph; ++variant) {
    if (variant == 1) { G = std::max<int64_t>(1, (int64_t)((double)n / 3.0)); table_cap = PH_LARGE_MAX_TABLE_BYTES; }
    const double lambda = (double)n / (double)G;
    // (groups of ~3 k-mers with one-byte pilots fill to ~0.84: start there instead of walking down from 0.95, every failed attempt is a
    //  full pass over the k-mers; groups of 6+ never fill beyond ~0.75)
    for (double alpha = table_cap == PH_MAX_TABLE_BYTES ? (lambda > 6.0 ? 0.75 : 0.95) : 0.84; alpha >= 0.55 && !ix->has_ph; alpha -= 0.02) {
      const uint64_t N = std::max<uint64_t>({(uint64_t)((double)n / alpha) + 1, nb_min, 1024});
      if (dbg) std::fprintf(stderr, "[nb_index] perfect hash: alpha=%.2f N=%llu G=%lld lambda=%.2f\n", alpha, (unsigned long long)N, (long long)G, lambda);
      if ((int64_t)N * 16 > table_cap || N > 0x7FFFFFFFull || lambda > 16.0) break;
      PHash h{}; h.a = mul_a; h.c = mul_c; h.n_slots = (uint32_t)N; h.n_groups = (uint32_t)G; h.hb = hb; h.himask = himask; h.stride = (uint32_t)(N >> (hb + 2));
      if (h.stride >= (1u << 24) || hb + 2 > 16) break;                       // (24-bit multiply operands in ph_probe)
      std::vector<PHProbe> pr((size_t)n);
      std::vector<uint32_t> gcount((size_t)G + 1, 0);
      for (int64_t i = 0; i < n; ++i) { pr[(size_t)i] = ph_probe(canons[(size_t)i].key, h); ++gcount[pr[(size_t)i].group + 1]; }
      std::vector<uint32_t> gstart(gcount);
      for (int64_t g = 0; g < G; ++g) gstart[(size_t)g + 1] += gstart[(size_t)g];
      std::vector<uint32_t> members((size_t)n), fill(gstart.begin(), gstart.end() - 1);
      for (int64_t i = 0; i < n; ++i) members[fill[pr[(size_t)i].group]++] = (uint32_t)i;
      std::vector<uint32_t> order((size_t)G); std::iota(order.begin(), order.end(), 0u);
      std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
        const uint32_t sx = gstart[x + 1] - gstart[x], sy = gstart[y + 1] - gstart[y]; return sx != sy ? sx > sy : x < y; });
      std::vector<uint8_t> used((size_t)N, 0), pilots((size_t)G, 0);
      std::vector<uint32_t> slots;
      bool ok = true;
      for (uint32_t g : order) {
        const uint32_t b = gstart[g], e = gstart[g + 1];
        if (b == e) continue;
        bool placed = false;
        for (uint32_t pilot = 0; pilot < 256 && !placed; ++pilot) {
          slots.clear(); bool clash = false;
          for (uint32_t t = b; t < e && !clash; ++t) {
            const uint32_t sl = ph_slot(pr[members[t]], pilot, h);
            if (used[sl]) clash = true;
            for (uint32_t x : slots) if (x == sl) clash = true;
            slots.push_back(sl);
          }
          if (clash) continue;
          for (uint32_t x : slots) used[x] = 1;
          pilots[g] = (uint8_t)pilot; placed = true;
        }
        if (!placed) { ok = false; if (dbg) std::fprintf(stderr, "[nb_index]   a group of %u k-mers found no pilot\n", e - b); break; }
      }
      if (!ok) continue;
      std::vector<PHEntry> tab((size_t)N, PHEntry{0xFFFFFFFFu, PAY_ABSENT, 0ull});
      bool bad_key = false;
      for (int64_t i = 0; i < n; ++i) {
        const Canon& cn = canons[(size_t)i]; const PHProbe& q = pr[(size_t)i];
        const bool both = cn.pay[0] != PAY_ABSENT && cn.pay[1] != PAY_ABSENT;
        const int strand = cn.pay[0] != PAY_ABSENT ? 0 : 1;
        const uint32_t pay = cn.pay[strand], colour = pay >> 10;
        const ColDesc& d = ix->desc[colour];
        PHEntry en; en.key = q.want | (both ? 2u : 0u) | (uint32_t)strand;
        en.pay = (pay & 0x3FFu) | (d.nw == 1 ? (d.w0 << 11) : ((1u << 10) | (colour << 11)));
        en.m = d.nw == 1 ? d.m0 : 0ull;
        if (en.key == 0xFFFFFFFFu || d.w0 > 0x1FFFFFu || colour > 0x1FFFFFu) bad_key = true;
        tab[ph_slot(q, pilots[q.group], h)] = en;
      }
      if (bad_key) { if (dbg) std::fprintf(stderr, "[nb_index]   key / word index out of range\n"); break; }
      ix->has_ph = true; ix->ph = h; ix->ph_table.swap(tab); ix->ph_pilots.swap(pilots);
    }
    }

    lap("perfect hash");
    // 5c. anchor structures of the anchor-extend kernel (nb_internal.h): text, per-k-mer anchor (first occurrence) + room, sparse AND table
    const char* an_env = std::getenv("NB_ANCHOR");
    int64_t T = 0;
    for (int s = 0; s < n_seqs; ++s) T += lens[s];
    // words per sparse-table entry: the widest component, two at the most. Sequences of wider components (loci merged by shared k-mers,
    // repeat families) keep their anchors but are marked: a read that meets one is mapped by the wave-per-read kernel. No anchors at all
    // when that would be more than a quarter of the library.
    const int NW = std::min(ix->max_comp_words, 2);
    int64_t wide_bases = 0;
    for (int s = 0; s < n_seqs; ++s) if ((int)ix->comp_nw[(size_t)s] > NW) wide_bases += lens[s];
    if (ix->has_ph && !(an_env && an_env[0] == '0') && T < ANCHOR_MAX_TEXT && (int64_t)ANCHOR_LEVELS * T * NW * 8 <= ANCHOR_MAX_SPARSE_BYTES && wide_bases * 4 <= T) {
      std::vector<uint8_t> code((size_t)T);
      std::vector<uint32_t> vrun((size_t)T + 1, 0);          // valid bases from j to the end of its sequence / the next N
      std::vector<uint32_t> seq_at((size_t)T);
      { int64_t j0 = 0;
        for (int s = 0; s < n_seqs; ++s) {
          for (int64_t i = 0; i < lens[s]; ++i) { code[(size_t)(j0 + i)] = (uint8_t)code_of(seqs[s][i]); seq_at[(size_t)(j0 + i)] = (uint32_t)s; }
          uint32_t run = 0;
          for (int64_t i = lens[s] - 1; i >= 0; --i) { run = code[(size_t)(j0 + i)] == 4 ? 0u : run + 1u; vrun[(size_t)(j0 + i)] = run; }
          j0 += lens[s];
        } }
      ix->text_len = T; ix->anchor_nw = NW;
      ix->text.assign((size_t)((T + 15) / 16 + 4), 0u);
      for (int64_t j = 0; j < T; ++j) if (code[(size_t)j] < 4) ix->text[(size_t)(j >> 4)] |= (uint32_t)code[(size_t)j] << (2 * (j & 15));
      // canonical entry (index into canons) of the k-mer that starts at every text position
      constexpr uint32_t NONE = 0xFFFFFFFFu;
      std::vector<uint32_t> cidx((size_t)T, NONE);
      parallel_ranges((size_t)T, threads, [&](size_t lo, size_t hi) {
        for (size_t j = lo; j < hi; ++j) {
          if (vrun[j] < (uint32_t)k) continue;
          uint64_t x = 0;
          for (int t = 0; t < k; ++t) x |= (uint64_t)code[j + (size_t)t] << (2 * t);
          const uint64_t y = rc_kmer_host(x, k), canon = std::min(x, y);
          auto it = std::lower_bound(canons.begin(), canons.end(), canon, [](const Canon& c, uint64_t v) { return c.key < v; });
          cidx[j] = (uint32_t)(it - canons.begin());             // (every library k-mer is in canons)
        }
      });
      auto is_both = [&](uint32_t i) { return canons[i].pay[0] != PAY_ABSENT && canons[i].pay[1] != PAY_ABSENT; };
      std::vector<uint8_t> room((size_t)T + 1, 0);
      for (int64_t j = T - 2; j >= 0; --j) {
        const uint32_t a = cidx[(size_t)j], b = cidx[(size_t)j + 1];
        room[(size_t)j] = (a != NONE && b != NONE && !is_both(b)) ? (uint8_t)std::min<uint32_t>((uint32_t)room[(size_t)j + 1] + 1u, ANCHOR_MAX_ROOM) : (uint8_t)0;
      }
      // r_internal (SPEC 2) of the k-mer that starts at every text position, as it is oriented there: what lets the walk tolerate a
      // mismatch (num_mismatches > 0) while it follows the text — the unique right extension of such a k-mer IS the text's next base
      ix->text_rint.assign((size_t)((T + 31) / 32 + 2), 0u);
      parallel_ranges((size_t)((T + 31) / 32), threads, [&](size_t wlo, size_t whi) {          // (a range of whole words per thread)
        for (int64_t j = (int64_t)wlo * 32; j < std::min<int64_t>(T, (int64_t)whi * 32); ++j) {
          const uint32_t a = cidx[(size_t)j];
          if (a == NONE) continue;
          uint64_t x = 0;
          for (int t = 0; t < k; ++t) x |= (uint64_t)code[(size_t)j + (size_t)t] << (2 * t);
          const uint32_t pay = canons[a].pay[x != canons[a].key ? 1 : 0];
          if (pay != PAY_ABSENT && ((pay >> 8) & 1u)) ix->text_rint[(size_t)(j >> 5)] |= 1u << (j & 31);
        }
      });
      std::vector<uint8_t> lroom((size_t)T + 1, 0);          // the same towards the left (SPEC 3's left extension from the first seed)
      for (int64_t j = 1; j < T; ++j) {
        const uint32_t a = cidx[(size_t)j], b = cidx[(size_t)j - 1];
        lroom[(size_t)j] = (a != NONE && b != NONE && !is_both(b)) ? (uint8_t)std::min<uint32_t>((uint32_t)lroom[(size_t)j - 1] + 1u, ANCHOR_MAX_LROOM) : (uint8_t)0;
      }
      // Which occurrence anchors a k-mer: any would be exact; the walk runs longest along the sequence that most other sequences agree
      // with (a read leaves its anchor at the READ's private variants either way, and at the ANCHOR's private variants on top): the
      // occurrence whose next ANCHOR_LOOKAHEAD text k-mers are shared by the most sequences (sum of class sizes), first one on ties.
      std::vector<uint64_t> ahead((size_t)T + 2, 0);          // prefix sums of the class sizes along the text
      for (int64_t j = 0; j < T; ++j) {
        const uint32_t i = cidx[(size_t)j];
        uint64_t sz = 0;
        if (i != NONE) { const Canon& cn = canons[i]; sz = ix->eq_sizes[(cn.pay[0] != PAY_ABSENT ? cn.pay[0] : cn.pay[1]) >> 10]; }
        ahead[(size_t)j + 1] = ahead[(size_t)j] + sz;
      }
      lap("text, k-mer of every position");
      std::vector<uint32_t> tpos_of((size_t)n_canon, NONE);
      std::vector<uint64_t> best((size_t)n_canon, 0);
      for (int64_t j = 0; j < T; ++j) {
        const uint32_t i = cidx[(size_t)j];
        if (i == NONE) continue;
        const int64_t r = std::min<int64_t>((int64_t)room[(size_t)j], ANCHOR_LOOKAHEAD);
        const uint64_t score = ahead[(size_t)(j + 1 + r)] - ahead[(size_t)j + 1];
        if (tpos_of[i] == NONE || score > best[i]) { tpos_of[i] = (uint32_t)j; best[i] = score; }
      }
      lap("anchors");
      // level 0: the class of the k-mer at every text position as NW words from the first word of its sequence's component
      ix->sparse.resize((size_t)ANCHOR_LEVELS * (size_t)T * (size_t)NW);      // (uninitialised: level 0 is zeroed range by range below, the levels above write every word)
      lap("  sparse: allocate");
      parallel_ranges((size_t)T, threads, [&](size_t lo, size_t hi) {
        std::memset(ix->sparse.data() + lo * (size_t)NW, 0, (hi - lo) * (size_t)NW * 8);
        for (size_t j = lo; j < hi; ++j) {
          const uint32_t i = cidx[j];
          if (i == NONE) continue;
          const Canon& cn = canons[i];
          const uint32_t pay = cn.pay[0] != PAY_ABSENT ? cn.pay[0] : cn.pay[1];      // (both strands: never part of a run, any class will do)
          const ColDesc& d = ix->desc[pay >> 10];
          if ((int)ix->comp_nw[seq_at[j]] > NW) continue;      // (marked in the anchor entry: never read)
          const uint32_t base = ix->comp_w0[seq_at[j]];
          uint64_t* e = ix->sparse.data() + j * (size_t)NW;
          if (d.w0 >= base && d.w0 - base < (uint32_t)NW) e[d.w0 - base] = d.m0;
          if (d.nw > 1 && d.w0 + 1 >= base && d.w0 + 1 - base < (uint32_t)NW) e[d.w0 + 1 - base] = d.m1;
        }
      });
      lap("  sparse: level 0");
      for (int l = 1; l < ANCHOR_LEVELS; ++l) {
        const uint64_t* prev = ix->sparse.data() + (size_t)(l - 1) * (size_t)T * (size_t)NW; uint64_t* cur = ix->sparse.data() + (size_t)l * (size_t)T * (size_t)NW;
        const int64_t half = 1ll << (l - 1);
        parallel_ranges((size_t)T, threads, [&](size_t lo, size_t hi) {
          for (size_t j = lo; j < hi; ++j) for (int w = 0; w < NW; ++w)
            cur[j * (size_t)NW + (size_t)w] = ((int64_t)j + half < T) ? (prev[j * (size_t)NW + (size_t)w] & prev[(j + (size_t)half) * (size_t)NW + (size_t)w]) : 0ull;
        });
      }
      lap("sparse table");
      ix->atable.assign(ix->ph_table.size(), AEntry{0xFFFFFFFFu, PAY_ABSENT, 0u, 0u});
      bool fits = true;
      for (int64_t i = 0; i < n_canon; ++i) {
        const PHProbe q = ph_probe(canons[(size_t)i].key, ix->ph);
        const uint32_t slot = ph_slot(q, ix->ph_pilots[q.group], ix->ph);
        const PHEntry& pe = ix->ph_table[slot];
        const uint32_t t = tpos_of[(size_t)i];
        const uint32_t base = t == NONE ? 0u : ix->comp_w0[seq_at[t]];
        if (base > 0xFFFFu) fits = false;
        const uint32_t wide = (t != NONE && (int)ix->comp_nw[seq_at[t]] > NW) ? ANCHOR_WIDE : 0u;
        ix->atable[slot] = AEntry{pe.key, pe.pay, t == NONE ? 0u : t, ((t == NONE || is_both((uint32_t)i)) ? 0u : ((uint32_t)room[t] | ((uint32_t)lroom[t] << 8))) | wide | (base << 16)};
      }
      ix->akeys.resize(ix->atable.size());
      for (size_t i = 0; i < ix->atable.size(); ++i) ix->akeys[i] = ix->atable[i].key;
      lap("anchor entries");
      ix->has_anchor = fits;
      if (fits) {                                                      // presence filter (nb_internal.h): the m-mers of both orientations of every k-mer of the table
        ix->pf_s = pf_stride(k); ix->pf_m = k + 1 - ix->pf_s;
        ix->pf_wbits = pf_words_log2(n_canon);
        ix->pfilter.assign((size_t)1 << ix->pf_wbits, 0u);
        ix->lfilter.assign((size_t)1 << (LF_BITS_LOG2 - 5), 0u);
        const uint64_t mmask = (1ull << (2 * ix->pf_m)) - 1;
        for (int64_t i = 0; i < n_canon; ++i) {
          const uint64_t c = canons[(size_t)i].key, both[2] = {c, rc_kmer_host(c, k)};
          for (int o = 0; o < 2; ++o) for (int sh = 0; sh < ix->pf_s; ++sh) {
              const uint64_t mm = (both[o] >> (2 * sh)) & mmask;
              const uint32_t h0 = pf_hash(mm), li = lf_index(mm);
              ix->pfilter[pf_word(h0, ix->pf_wbits)] |= pf_mask(h0);
              ix->lfilter[li >> 5] |= 1u << (li & 31u);
            }
        }
        { int64_t set = 0; for (uint32_t w : ix->lfilter) set += __builtin_popcount(w);
          if (dbg) std::fprintf(stderr, "[nb_index] presence filter: m = %d, stride %d, %u KB, LDS stage %.1f %% full\n", ix->pf_m, ix->pf_s, (unsigned)(ix->pfilter.size() * 4 / 1024), 100.0 * (double)set / (double)((int64_t)1 << LF_BITS_LOG2));
          if ((double)set > LF_MAX_FILL * (double)((int64_t)1 << LF_BITS_LOG2)) std::vector<uint32_t>().swap(ix->lfilter); }
        lap("presence filter");
      }
      if (!fits) { std::vector<uint32_t>().swap(ix->text_rint); std::vector<uint32_t>().swap(ix->lfilter); std::vector<uint32_t>().swap(ix->pfilter); std::vector<uint32_t>().swap(ix->akeys); std::vector<AEntry>().swap(ix->atable); WordsNoInit().swap(ix->sparse); std::vector<uint32_t>().swap(ix->text); }
    }
  }
  ix->n_canonical = n_canon;
  *out = ix;
  return NB_OK;
}

}  // namespace nb
