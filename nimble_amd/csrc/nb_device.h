// nb_device.h — device-side views, wave helpers and the context object shared by the kernel translation units
// (nb_kernels.hip: one read per wave; nb_lane.hip: one read per lane). Not part of the C-ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "nb_internal.h"
#include "nb_wave.h"

namespace nb {

// ------------------------------------------------------------------------------------------------------------
// device-side views
struct DevIndex {
  const Bucket* buckets;  // wide table (k > COMPACT_MAX_K): hash.n_buckets 32-byte buckets (nb_internal.h)
  const CBucket* ctable;  // compact table (k <= COMPACT_MAX_K): chash.nb 16-byte buckets; exactly one of the two is set
  const ColDesc* desc;    // n_colours 16-byte colour descriptors (internal sequence numbering)
  const uint64_t* packed; // spans of the colours wider than one word
  KmerHash hash;
  CHash chash;
  int W;                  // u64 words per colour row
  int K;
};

// Compact table: both candidate buckets of a k-mer as they came back from memory (a = home, c = alternate; uint4 = {key0, pay0,
// key1, pay1}) -> payload on the read strand (entry whose strand bit equals o = "the read's k-mer is the reverse complement of
// the canonical key") and on the reverse-complement strand. Entries of the home bucket must carry which = 0, those of the
// alternate bucket which = 1 (nb_internal.h: that is what makes the match exact). 8 compares + 8 selects, no loop.
__device__ __forceinline__ void compact_match(const uint4& a, const uint4& c, uint32_t want, uint32_t o, bool valid, uint32_t& payF, uint32_t& payR) {
  const uint32_t wF1 = want | o, wR1 = wF1 ^ 1u, wF2 = wF1 | 2u, wR2 = wR1 | 2u;
  uint32_t f = PAY_ABSENT, r = PAY_ABSENT;
  f = a.x == wF1 ? a.y : f; r = a.x == wR1 ? a.y : r;
  f = a.z == wF1 ? a.w : f; r = a.z == wR1 ? a.w : r;
  f = c.x == wF2 ? c.y : f; r = c.x == wR2 ? c.y : r;
  f = c.z == wF2 ? c.w : f; r = c.z == wR2 ? c.w : r;
  payF = valid ? f : PAY_ABSENT; payR = valid ? r : PAY_ABSENT;
}
__device__ __forceinline__ uint4 load_cbucket(const CBucket* __restrict__ T, uint32_t b) { return *reinterpret_cast<const uint4*>(&T[b]); }

// One bucket probe for one lane. Returns true on a hit and fills p0/p1; `more` = the bucket is flagged OVERFLOW (the key
// may live in the next bucket).
__device__ __forceinline__ bool probe_bucket(const Bucket* __restrict__ B, uint32_t b, uint64_t canon, uint32_t& p0, uint32_t& p1, bool& more) {
  const uint4* kp = reinterpret_cast<const uint4*>(&B[b]);
  const uint4 a = kp[0], c = kp[1];
  const uint64_t k0 = ((uint64_t)a.y << 32) | a.x, k1 = ((uint64_t)c.y << 32) | c.x;
  const bool m0 = (k0 & KEY_MASK) == canon, m1 = k1 == canon;
  more = (k0 >> 63) != 0;
  p0 = m0 ? a.z : c.z; p1 = m0 ? a.w : c.w;
  return m0 || m1;
}

struct DevClasses {
  unsigned long long* keys;    // cap slots, 0 = free
  unsigned int* ids;           // cap slots, id+1 once published, 0 = not yet, 0xFFFFFFFF = overflow
  unsigned long long* rows;    // max_classes * W
  unsigned long long* counts;  // max_classes
  unsigned int* spans;         // max_classes: first non-zero word | last non-zero word << 16 of rows[id] (written before the id is
                               // published): lets one lane verify a row it holds as (word index, words) without reading all W words
  unsigned int* n_classes;     // allocated ids
  unsigned int* err;           // bit0: class table overflow, bit1: spin timeout
  unsigned long long* reasons; // NB_NUM_REASONS
  unsigned int cap_log2;
  unsigned int max_classes;
};

struct BatchArgs {
  const uint64_t* reads; const uint16_t* lens; int max_len; int stride; int nwb;     // r1
  const uint64_t* mates; const uint16_t* mate_lens; int mate_max_len; int mstride; int mnwb;  // r2 (optional)
  long long n;
  const uint32_t* n_dev;     // optional: the number of templates lives on the device (redo list of the lane-per-read path); n is then an upper bound
  uint32_t* out_class; uint32_t* out_info; uint64_t* out_cov;
  const uint16_t* pct_min;   // pct_min[L] = smallest cov with (double)cov/(double)L >= score_percent, L in 0..max(max_len,mate_max_len)
  int pct_fixed;             // pct_min[max_len], used when every read has length max_len (lens == NULL): no table load
  int score_threshold, score_filter, num_mismatches, discard_multiple, intersect_level, require_valid_pair, strand_filter;
};

// ------------------------------------------------------------------------------------------------------------
// wave-level helpers (wave64)
__device__ __forceinline__ uint64_t bcast_u64(uint64_t v) {  // lane 0's value to every lane
  uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z ^= z >> 33; z *= 0xff51afd7ed558ccdull; z ^= z >> 33; z *= 0xc4ceb9fe1a85ec53ull; z ^= z >> 33; return z;
}

// reverse complement of a k-mer held LSB-first (nb_internal.h encoding)
__device__ __forceinline__ uint64_t rc_kmer(uint64_t x, int K) {
  uint64_t v = __brevll(~x);
  v = ((v & 0x5555555555555555ull) << 1) | ((v >> 1) & 0x5555555555555555ull);
  return v >> (64 - 2 * K);
}

// ------------------------------------------------------------------------------------------------------------
// multiword bit masks over k-mer positions (NR words, wave-uniform values)
template <int NR> struct Mask { uint64_t w[NR]; };

// first set bit at a position p >= from with (p - from) % 3 == 0, or -1          (SPEC §3 seed(), SCAN_STRIDE = 3)
template <int NR> __device__ __forceinline__ int ffs_stride3(const Mask<NR>& m, int from) {
  const uint64_t PAT = 0x9249249249249249ull;  // bits t with t % 3 == 0
#pragma unroll
  for (int wi = 0; wi < NR; ++wi) {
    const int base = wi * 64;
    if (from < base + 64) {
      uint64_t v = m.w[wi];
      const int res = ((from % 3) - (wi % 3) + 3) % 3;  // want t with (base + t) % 3 == from % 3; 64 % 3 == 1
      v &= PAT << res;
      if (from > base) v &= ~0ull << (from - base);
      if (v) return base + __builtin_ctzll(v);
    }
  }
  return -1;
}
// first ZERO bit at a position >= from (caller guarantees one exists below NR*64)
template <int NR> __device__ __forceinline__ int ffz_from(const Mask<NR>& m, int from) {
#pragma unroll
  for (int wi = 0; wi < NR; ++wi) {
    const int base = wi * 64;
    if (from < base + 64) {
      uint64_t v = ~m.w[wi];
      if (from > base) v &= ~0ull << (from - base);
      if (v) return base + __builtin_ctzll(v);
    }
  }
  return NR * 64;
}
// highest ZERO bit at a position <= from (bit 0 of the masks passed here is always zero)
template <int NR> __device__ __forceinline__ int flz_upto(const Mask<NR>& m, int from) {
#pragma unroll
  for (int wi = NR - 1; wi >= 0; --wi) {
    const int base = wi * 64;
    if (from >= base) {
      uint64_t v = ~m.w[wi];
      if (from < base + 63) v &= (2ull << (from - base)) - 1;
      if (v) return base + 63 - __builtin_clzll(v);
    }
  }
  return 0;
}
template <int NR> __device__ __forceinline__ void set_range(Mask<NR>& m, int lo, int hi) {  // inclusive, lo <= hi
#pragma unroll
  for (int wi = 0; wi < NR; ++wi) {
    const int base = wi * 64;
    if (lo < base + 64 && hi >= base) {
      const int a = lo > base ? lo - base : 0, b = hi < base + 63 ? hi - base : 63;
      const uint64_t hi_mask = (b == 63) ? ~0ull : ((2ull << b) - 1);
      m.w[wi] |= hi_mask & (~0ull << a);
    }
  }
}
template <int NR> __device__ __forceinline__ bool any_set(const Mask<NR>& m) {
  uint64_t v = 0;
#pragma unroll
  for (int wi = 0; wi < NR; ++wi) v |= m.w[wi];
  return v != 0;
}

// ------------------------------------------------------------------------------------------------------------
// SPEC §3 on masks, exact mode (no tolerated mismatch). H = k-mer present, R = present and the next read base is a
// right extension, LX = present and the previous read base is a left extension. Positions are in walk order
// (position 0 = first k-mer of the sequence being mapped). Returns coverage (0 = NONE) and the visited positions.
template <int NR>
__device__ __forceinline__ int walk_masks(const Mask<NR>& H, const Mask<NR>& R, const Mask<NR>& LX, int L, int K, Mask<NR>& V) {
#pragma unroll
  for (int wi = 0; wi < NR; ++wi) V.w[wi] = 0;
  const int last = L - K;
  int p = ffs_stride3<NR>(H, 0);
  if (p < 0) return 0;
  int cov = 0;
  if (p >= L / 5 && p > 0) {               // left extension: consecutive LX bits going down from p
    const int z = flz_upto<NR>(LX, p);
    const int t = p - z;
    if (t > 0) { cov += t; set_range<NR>(V, p - t, p - 1); }
  }
  for (;;) {
    const int e = ffz_from<NR>(R, p);      // run p..e
    cov += K + (e - p);
    set_range<NR>(V, p, e);
    const int b = e + K;
    if (b > last) break;
    p = ffs_stride3<NR>(H, b);
    if (p < 0) break;
  }
  return cov;
}

// highest set bit at a position i <= from with (from - i) % 3 == 0, or -1
template <int NR> __device__ __forceinline__ int fls_stride3_down(const Mask<NR>& m, int from) {
  const uint64_t PAT = 0x9249249249249249ull;
#pragma unroll
  for (int wi = NR - 1; wi >= 0; --wi) {
    const int base = wi * 64;
    if (from >= base) {
      uint64_t v = m.w[wi];
      const int res = ((from % 3) - (wi % 3) + 3) % 3;
      v &= PAT << res;
      if (from < base + 63) v &= (2ull << (from - base)) - 1;
      if (v) return base + 63 - __builtin_clzll(v);
    }
  }
  return -1;
}

// SPEC §3 for the REVERSE-COMPLEMENT strand, evaluated directly on masks indexed by read position i (rc position
// j = P-1-i): the walk advances towards lower i, "left" is towards higher i. Same arithmetic as walk_masks on the
// bit-reversed masks, without reversing them. H/R/LX are the rc strand's facts at read position i.
template <int NR>
__device__ __forceinline__ int walk_masks_down(const Mask<NR>& H, const Mask<NR>& R, const Mask<NR>& LX, int L, int K, Mask<NR>& V) {
#pragma unroll
  for (int wi = 0; wi < NR; ++wi) V.w[wi] = 0;
  const int last = L - K;                       // = P-1
  int ip = fls_stride3_down<NR>(H, last);       // rc position 0 is read position last
  if (ip < 0) return 0;
  int cov = 0;
  const int pj = last - ip;
  if (pj >= L / 5 && pj > 0) {                  // left extension: consecutive LX bits going UP from ip
    const int z = ffz_from<NR>(LX, ip);
    const int t = z - ip;
    if (t > 0) { cov += t; set_range<NR>(V, ip + 1, ip + t); }
  }
  for (;;) {
    const int ei = flz_upto<NR>(R, ip);         // run ip .. ei (downwards)
    cov += K + (ip - ei);
    set_range<NR>(V, ei, ip);
    const int bj = (last - ei) + K;
    if (bj > last) break;
    ip = fls_stride3_down<NR>(H, last - bj);
    if (ip < 0) break;
  }
  return cov;
}

// ------------------------------------------------------------------------------------------------------------
// SPEC §4 checks 1-2 on a mapped strand (cheap part): returns 0 if it may be valid, else the reason.
__device__ __forceinline__ int cheap_reason_p(int cov, int pmin, const BatchArgs& a) {
  if (cov == 0) return NB_NO_HIT;
  if (cov < a.score_threshold) return NB_SCORE_THRESHOLD;
  if (cov < pmin) return NB_SCORE_PERCENT;
  return 0;
}
__device__ __forceinline__ int cheap_reason(int cov, int L, const BatchArgs& a) { return cheap_reason_p(cov, (int)a.pct_min[L], a); }

// ------------------------------------------------------------------------------------------------------------
// host side shared by the translation units
void set_error(const std::string& msg);
#define NB_HIP(call)                                                                                     \
  do {                                                                                                   \
    hipError_t e_ = (call);                                                                              \
    if (e_ != hipSuccess) { nb::set_error(std::string(#call) + ": " + hipGetErrorString(e_)); return NB_ERR_HIP; } \
  } while (0)

struct LaneLists {
  uint32_t* rc_list;      // reads to map in reverse-complement orientation (pass 1 of the lane-per-read path)
  uint32_t* redo_list;    // reads for the wave-per-read kernel
  uint32_t* counters;     // [0] = entries in rc_list, [1] = entries in redo_list
};
// (the only raw ISA string in the per-lane device code; tests/native/anchor_emul.cpp compiles that code for the host and blanks it)
#ifndef NB_WAIT_VMCNT0
#define NB_WAIT_VMCNT0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif

// everything the tail of a lane-per-read batch needs (class table, outputs, filters, lists): read from device memory at the point of use
struct LaneTail { DevClasses ct; BatchArgs a; LaneLists ll; int W; };

#if defined(__HIPCC__)
// Counts of a workgroup's hot classes in LDS (direct-mapped by the low bits of the class id, which are handed out in first-seen order: the
// classes every wave meets have small, distinct ids). With a library of a few loci every read adds to the same handful of counters, and
// atomics on one address serialise in the L2 (wave-per-read kernel, 20 sequences: 43.6 ms per 10 M reads against 16.7 for 1 000 sequences).
// tag / cnt: COUNT_CACHE words each, set to NB_NO_CLASS / 0 before use; count_cache_flush after the workgroup's last read.
constexpr int COUNT_CACHE = 256;
__device__ __forceinline__ void count_cache_init(unsigned int* tag, unsigned int* cnt) {
  for (int i = threadIdx.x; i < COUNT_CACHE; i += blockDim.x) { tag[i] = NB_NO_CLASS; cnt[i] = 0u; }
}
__device__ __forceinline__ void count_cache_add(unsigned int* tag, unsigned int* cnt, unsigned long long* counts, uint32_t cls, uint32_t mult) {
  const uint32_t slot = cls & (uint32_t)(COUNT_CACHE - 1);
  uint32_t t = tag[slot];
  if (t == NB_NO_CLASS) { t = atomicCAS(&tag[slot], NB_NO_CLASS, cls); if (t == NB_NO_CLASS) t = cls; }
  if (t == cls) atomicAdd(&cnt[slot], mult);
  else atomicAdd(&counts[cls], (unsigned long long)mult);
}
__device__ __forceinline__ void count_cache_flush(const unsigned int* tag, const unsigned int* cnt, unsigned long long* counts) {   // (after a __syncthreads)
  for (int i = threadIdx.x; i < COUNT_CACHE; i += blockDim.x)
    if (tag[i] != NB_NO_CLASS && cnt[i]) atomicAdd(&counts[tag[i]], (unsigned long long)cnt[i]);
}

// per-lane interning of a set of at most two words {word index lo: a0, lo + 1: a1} in the shared class table; returns the class id or
// NB_NO_CLASS. Same table protocol as intern_class (nb_kernels.hip): key = content hash (xor of one mixed word per non-zero (index, value)),
// full-content verification through (spans, words).
__device__ __forceinline__ uint32_t intern_lane2(const DevClasses& ct, int W, bool want, uint32_t lo, uint64_t a0, uint64_t a1) {
  const uint64_t hx = (a0 ? mix64(a0 ^ (0x9E3779B97F4A7C15ull * (uint64_t)(lo + 1u))) : 0ull) ^ (a1 ? mix64(a1 ^ (0x9E3779B97F4A7C15ull * (uint64_t)(lo + 2u))) : 0ull);
  const uint64_t key = hx ? hx : 1ull;
  const uint32_t span = (a0 ? lo : lo + 1u) | ((a1 ? lo + 1u : lo) << 16);
  const uint64_t smask = (1ull << ct.cap_log2) - 1;
  uint64_t h = (key * HASH_MULT) >> (64 - ct.cap_log2);
  uint32_t result = NB_NO_CLASS;
  uint64_t probes = 0;                              // slots passed: the table is full once every slot was seen (intern_class has the same bound)
  bool pending = want;
  for (uint32_t it = 0; it < (1u << 24) && __ballot(pending); ++it) {
    if (pending) {
      const uint64_t cur = __hip_atomic_load(&ct.keys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const uint32_t idp0 = __hip_atomic_load(&ct.ids[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // fetched with the key: one round trip on the common (already interned) path
      if (cur == 0) {
        unsigned long long expected = 0ull;
        __hip_atomic_compare_exchange_strong(&ct.keys[h], &expected, (unsigned long long)key, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (expected == 0ull) {                       // this lane owns the slot: allocate, write the row, publish
          const uint32_t id = __hip_atomic_fetch_add(ct.n_classes, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (id >= ct.max_classes) {
            atomicOr(ct.err, 1u);
            __hip_atomic_store(&ct.ids[h], 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            pending = false;
          } else {
            for (int w = 0; w < W; ++w)
              __hip_atomic_store(&ct.rows[(size_t)id * W + w], (unsigned long long)((uint32_t)w == lo ? a0 : ((uint32_t)w == lo + 1u ? a1 : 0ull)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&ct.spans[id], span, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            NB_WAIT_VMCNT0();
            __hip_atomic_store(&ct.ids[h], id + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            result = id; pending = false;
          }
        }
        // lost the race: look at the same slot again in the next iteration (its key is set now)
      } else if (cur == key) {
        const uint32_t idp = idp0;                    // (an id still 0 here is polled again in the next iteration, with a fresh load)
        if (idp == 0xFFFFFFFFu) pending = false;      // table overflow recorded by the owner
        else if (idp != 0u) {
          const uint32_t id = idp - 1u;
          const uint32_t sp = __hip_atomic_load(&ct.spans[id], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const uint64_t word = __hip_atomic_load(&ct.rows[(size_t)id * W + lo], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const uint64_t word1 = a1 ? __hip_atomic_load(&ct.rows[(size_t)id * W + lo + 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
          if (sp == span && word == a0 && word1 == a1) { result = id; pending = false; }
          else { h = (h + 1) & smask; ++probes; }     // a different set with the same 64-bit hash: keep probing
        }
        // idp == 0: the owner is still publishing: poll again in the next iteration
      } else { h = (h + 1) & smask; ++probes; }
      if (probes > smask) { atomicOr(ct.err, 1u); pending = false; }
    }
  }
  if (pending) atomicOr(ct.err, 2u);
  return result;
}
__device__ __forceinline__ uint32_t intern_lane(const DevClasses& ct, int W, bool want, uint32_t lo, uint64_t a0) { return intern_lane2(ct, W, want, lo, a0, 0ull); }

#endif
}  // namespace nb

struct nb_ctx;
namespace nb {
// nb_anchor.hip: the anchor-extend kernel (one launch; what it cannot finish exactly goes on ll.redo_list like the lane path's)
int launch_anchor(nb_ctx* c, int nr, const DevIndex& ix, const BatchArgs& a, const LaneLists& ll, hipStream_t s);
int launch_anchor_pairs(nb_ctx* c, const DevIndex& ix, const BatchArgs& a, const LaneLists& ll, void* mate_res, hipStream_t s);   // paired templates
int64_t anchor_mate_res_bytes();                           // bytes of one raw mate result (two per template in mate_res)
bool anchor_fits(const nb_ctx* c, int max_len);
bool anchor_pilots_in_lds(const nb_ctx* c, int max_len);   // which instantiation launch_anchor picks (kernel names for the profiles)
// nb_lane.hip: both lane-per-read passes of one batch, then gather of the redo reads; nb_lane_scatter puts their results back
int launch_lane(nb_ctx* c, int nr, const DevIndex& ix, const BatchArgs& a, const LaneLists& ll, hipStream_t s);
int64_t lane_lds_bytes(const nb_ctx* c, int max_len, int* waves_per_block);   // 0 = the lane path does not fit this shape
int launch_lane_gather(nb_ctx* c, const BatchArgs& a, const LaneLists& ll, uint64_t* redo_reads, uint16_t* redo_lens, hipStream_t s);
int launch_lane_scatter(nb_ctx* c, const BatchArgs& a, const LaneLists& ll, const uint32_t* redo_out, hipStream_t s);
}  // namespace nb

struct nb_index { nb::HostIndex* h; };

struct nb_ctx {
  int device = 0;
  int n_cu = 256;
  // bound index (device copies)
  nb::Bucket* d_slots = nullptr; nb::CBucket* d_ctable = nullptr; bool compact = false; nb::CHash chash{}; nb::ColDesc* d_desc = nullptr; uint64_t* d_packed = nullptr; nb::KmerHash hash{}; int W = 0 /* words per class row, INTERNAL numbering */, W_ext = 0, K = 0, n_seqs = 0; int64_t index_bytes = 0;
  std::vector<int32_t> inv_perm;   // internal bit -> original sequence index (-1 = padding): class rows leave the library in original numbering
  // class table
  nb::DevClasses ct{}; nb::DevClasses* d_ct = nullptr; int64_t max_classes = 0;
  // worklists of the lane-per-read path (nb_lane.hip): reads to map in reverse-complement orientation, reads for the wave-per-read kernel
  uint32_t* d_lists = nullptr; int64_t lists_cap = 0; uint32_t* d_list_counters = nullptr; int max_desc_nw = 0;
  uint32_t* d_scan_counts = nullptr; int64_t scan_counts_cap = 0, scan_list_cap = 0;   // anchor kernels: per-workgroup counts / capacity of the segmented scan list
  uint64_t* d_redo_mates = nullptr; int64_t redo_mate_words_cap = 0; uint16_t* d_redo_mate_lens = nullptr; void* d_mate_res = nullptr; int64_t mate_res_cap = 0;   // paired path of nb_anchor.hip
  uint64_t* d_redo_reads = nullptr; int64_t redo_words_cap = 0; uint16_t* d_redo_lens = nullptr; uint32_t* d_redo_out = nullptr;   // gathered redo reads + their results (4 u32 per read)
  void* d_lane_tail = nullptr;     // LaneTail of the current batch (nb_lane.hip)
  nb::PHEntry* d_ph_table = nullptr; uint8_t* d_ph_pilots = nullptr; nb::PHash ph{}; bool has_ph = false;   // perfect-hash table of the lane kernels
  uint32_t* d_retry_list = nullptr; uint32_t* d_retry_counts = nullptr; int64_t retry_cap = 0; uint32_t* d_retry_state = nullptr; int retry_state_dw = 0;   // stragglers of the anchor kernel's first pass (nb_anchor.hip)
  bool anchor_retry_ran = false;   // the last launch_anchor / launch_anchor_pairs ran the second pass (reported in last_kernels)
  uint32_t* d_akeys = nullptr;     // key words of d_atable alone (nb_internal.h, akeys)
  uint32_t* d_text_rint = nullptr; // r_internal bit of every text k-mer (anchor walk with num_mismatches > 0)
  uint32_t* d_lfilter = nullptr;   // first stage of the presence filter (copied to LDS by the scan kernel), null if not built
  uint32_t* d_pfilter = nullptr; uint32_t pf_wbits = 0; int pf_m = 0, pf_s = 1;   // presence filter of the anchor kernels (nb_internal.h)
  nb::AEntry* d_atable = nullptr; uint32_t* d_text = nullptr; uint64_t* d_sparse = nullptr; long long text_len = 0; bool has_anchor = false; int anchor_nw = 1;   // anchor structures (nb_anchor.hip)
  hipStream_t copy_stream = nullptr; hipStream_t last_stream = nullptr; bool last_stream_set = false;   // read-back stream; stream of the last nb_align_batch
  bool h_stage_pinned = true;
  void* h_stage = nullptr; size_t h_stage_cap = 0;   // pinned host staging buffer of nb_classes_read (grown on demand, kept between calls)
  std::string last_kernels;        // kernels launched by the last nb_align_batch (nb_last_kernel_names)   // d_ct: device copy of ct (the single-end kernel reads it through a pointer)
  // score_percent table
  uint16_t* d_pct = nullptr; int pct_len = 0; double pct_value = -1.0; std::vector<uint16_t> pct_host;
};

