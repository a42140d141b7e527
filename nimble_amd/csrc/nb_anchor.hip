// nb_anchor.hip — the anchor-extend kernel: one read per lane like nb_lane.hip, but the walk of docs/SPEC.md §3 follows the library TEXT.
//
// nb_lane.hip looks every k-mer position of a read up (131 table requests and ~380 wave-instructions per 150-bp read: it runs at the
// request rate of the memory system AND at the issue rate of the SIMDs at once, profiles/r02_tuning_notes.md §4). Fewer requests and fewer
// instructions per read need fewer POSITIONS handled one by one. Here a table hit names a text position (AEntry, nb_internal.h); from there
// the read is compared with the text 32 bases at a time, every matching base is one step of the walk (right extension by construction),
// and the AND of the classes of the k-mers passed comes from the sparse table with two loads. A lookup happens only where the read leaves
// its anchor sequence or where the walk has to seed:
//
//   record    each lane stages its packed read in its own LDS column; a read that maps in reverse-complement orientation (three probes
//             decide) rewrites the column IN PLACE as the reverse complement, so every later access is a plain 32-base window (Rec).
//   round     a lane is in SEEK (seed scan, stride 3), STEP (the k-mer that ends at the base where the last run stopped: the walk goes
//             on iff it is in the table on the stream strand and the base before it is one of its left extensions — the same (K+1)-mer
//             test as nb_lane.hip), EXT (following the text) or done: at most one lookup and one 32-base comparison per round.
//   other     SPEC §4 needs the coverage of the OTHER orientation too. It is zero unless one of that orientation's seed-scan positions
//   strand    (stream positions p = last, last - 3, ...) hits on the opposite strand. Positions inside verified runs cannot (runs never
//             enter k-mers that occur on both strands); the remaining ones are looked up after the walk, eight in flight per lane.
//   scan      a read none of whose probes is in the table at all is settled by nb_anchor_scan_kernel: every seed-scan position of both
//             orientations, behind the presence filter (one m-mer per tile of S positions: LDS stage, Bloom filter, exact lookup for what
//             passes); nothing found = NO_HIT.
//   retry     the few lanes that still walk when the rest of their wave is done hand their read over WITH the state of its walk; a second
//             launch resumes those walks in waves that hold nothing else.
//   mismatch  num_mismatches > 0 (MM): while the read follows the text a base that differs is passed over where the text k-mer is
//             r_internal, and the walk goes on from the corrected k-mer (SPEC §3).
//   redo      anything else — a k-mer on both strands, any opposite-strand hit, a first seed at or beyond L/5 (left extension), an
//             anchor in a component wider than the sparse table's entries — puts the read on the redo list of the wave-per-read kernel.
//   pairs     MATE form of both kernels + nb_anchor_pair_kernel: SPEC §4 for paired templates from the two mates' raw alignments.
// Classes of one or two 64-bit words (NW); first level of the perfect hash in LDS or, for large libraries, read from L2 (PLDS).
// Debug builds: -DNB_ANCHOR_PROF (phase clock, scripts/anchor_phases.py), -DNB_ANCHOR_CHECK (bounds-checked, scripts/anchor_check.py);
// the same device code runs on the CPU with one lane per wave under ASan / UBSan (tests/native/anchor_emul.cpp).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "nb_device.h"

namespace nb {

// Paired templates (SPEC §4): each mate is walked like a single read, but what leaves the kernel is the raw alignment of the orientation
// that hits — coverage, orientation, the class words — and the guarantee that the other orientation has no hit at all; nb_anchor_pair_kernel
// combines the two mates of a template. status: 0 = as described, 1 = cannot be settled here (the pair goes to the wave-per-read kernel),
// 2 = shorter than k, 3 = on the scan list (the scan kernel turns it into 0 with coverage 0, or into 1).
struct MateRes { uint32_t info, lo; uint64_t a0, a1; };    // info = coverage | orientation << 16 | status << 17
constexpr uint32_t MATE_OK = 0u, MATE_REDO = 1u, MATE_SHORT = 2u, MATE_SCAN = 3u;

struct AnchorHead {
  const AEntry* table; const uint32_t* keys; const uint8_t* pilots; PHash ph; int K;     // keys[slot] = table[slot].key (presence / strand tests)
  const uint32_t* text; const uint64_t* sparse; long long T;
  const uint64_t* reads; const uint16_t* lens; int max_len, stride, nwb;
  long long n;
  int pilot_lds_bytes, rec_dwords;                   // dynamic LDS: [pilots][per wave: rec_dwords x 64 dwords]
  long long text_words, sparse_words; int W;         // sizes, for the bounds-checked debug build (-DNB_ANCHOR_PROF -DNB_ANCHOR_CHECK)
  MateRes* mate_out;                                 // MATE kernels: one raw result per sequence instead of SPEC §4 / interning / outputs
  int dbg;                                           // NB_ANCHOR_DBG: 8 = no LDS cache of class counts; -DNB_ANCHOR_PROF only: 1 = sparse loads, 2 = table entries, 4 = text loads redirected to element 0 (WRONG results: cost attribution)
  uint32_t* scan_counts; long long scan_seg;         // scan list: workgroup b owns entries [b * scan_seg, ...) of ll.rc_list and counts them in
                                                     // scan_counts[b] (one LDS atomic per wave instead of one contended device atomic)
  // Stragglers. A wave-batch lasts as long as its slowest lane (c2: 18 rounds for reads that need 7 on average). First pass (retry_out set):
  // once no more than retry_lanes lanes are still walking, those reads are put on the workgroup's segment of the retry list, WITH the state of
  // their walk (mode, position, coverage, class words, the run being followed, cover mask: 64 B), and the wave moves on. Second pass (retry_in
  // set, same grid): workgroup b maps its own segment: the record is staged again, the walk resumes where it stopped, in waves
  // that hold nothing but long walks.
  uint32_t* retry_out; const uint32_t* retry_in; uint32_t* retry_counts; int retry_lanes, retry_rounds;
  uint32_t* retry_state; long long retry_seg;        // the walk of a straggler as it stood at the hand-over (RetryShape<NR, NW>::DW dwords per list entry); entries per workgroup segment
  const uint32_t* pfilter; uint32_t pf_wbits;        // presence filter over the library's oriented m-mers (nb_internal.h): 2^pf_wbits words
  int pf_s;                                          // its tile size S (m = k + 1 - S)
  const uint32_t* lfilter;                           // its LDS-sized first stage (2^20 bits), or null (library too large for it to discriminate)
  const uint32_t* text_rint; int M;                  // num_mismatches > 0: r_internal bit of every text k-mer, the per-read budget of tolerated mismatches (SPEC 3)
};

struct __attribute__((packed, aligned(4))) U3 { uint32_t a, b, c; };
// the walk of a straggler at the hand-over to the second pass: d0 = coverage | flags << 16 | mode << 24 | rc << 26; d1 = lo | pending w0 << 16;
// class words; pending run's text position; pending n | first-seed position << 8 | its left room << 24; first-seed anchor text position;
// pos | run_q << 16; run_t; run_w0 | steps << 16 | maxsteps << 24; cover mask
// cover: which seed-scan positions of the OTHER orientation (stream positions p = res + 3 j, res = last % 3) are settled — looked up, or
// inside a verified run. One bit per lattice point: 64 NR positions hold at most 22 NR of them (one word up to NR = 3).
template <int NR> struct CoverShape { static constexpr int LW = NR <= 3 ? 1 : 2; };
__device__ __forceinline__ uint32_t div3(uint32_t x) { return (x * 0xAAABu) >> 17; }          // floor(x / 3), x < 98 304
template <int NR, int NW> struct RetryShape { static constexpr int DW = (11 + 2 * NW + 2 * CoverShape<NR>::LW + 3) & ~3; };  // (+ patches (2 dwords), cv_from | ended-by-room << 16; mm in d0 bits 27..31)

#ifdef NB_ANCHOR_PROF   // phase clock for tuning (-DNB_ANCHOR_PROF through NB_EXTRA_HIPCC_FLAGS): cycles per phase summed over waves, see scripts/anchor_phases.py
__device__ unsigned long long g_anchor_prof[16];
__device__ unsigned long long g_anchor_hist[128];   // [0, 64): reads by the number of rounds their lane was busy; [64, 128): wave-batches by rounds
#define PROF_MARK(slot) do { const long long t_ = clock64(); pacc[slot] += (unsigned long long)(t_ - tprof); tprof = t_; } while (0)
#define PROF_COUNT(slot, v) do { pacc[slot] += (unsigned long long)(v); } while (0)
#ifdef NB_ANCHOR_CHECK   // every computed index is compared with its array's size; a violation is counted (slots 10..15) and the access redirected to element 0
#define CHK(ok, slot) ((ok) ? true : (atomicAdd(&g_anchor_prof[slot], 1ull), false))
#endif
#else
#define PROF_MARK(slot) do {} while (0)
#define PROF_COUNT(slot, v) do {} while (0)
#endif
#ifndef CHK
#define CHK(ok, slot) (true)
#endif


// ---- per-lane record in LDS (column `lane` of rec[]: dword j at rec[j * 64 + lane]) ------------------------------------------------------
// layout: [0, nbd) packed bases, [nbd, nbd + nd) N bits, two zero dwords. stream_rc() rewrites the column IN PLACE as the reverse complement
// of the read, so that every later access is a plain forward window whatever the orientation.
struct Rec {
  uint32_t* col; int nbd, nd, L;
  __device__ __forceinline__ uint32_t& at(int j) const { return col[(CHK(j >= 0 && j < nbd + nd + 2, 15) ? j : 0) * 64]; }
  // 32 stream bases from position q (2 bits each, first base at bit 0) and their N-or-beyond-the-read bits
  __device__ __forceinline__ void get32(int q, uint64_t& win, uint32_t& nwin) const {
    const int j = q >> 4; const uint32_t sh = 2u * ((uint32_t)q & 15u);
    const uint32_t d0 = at(j), d1 = at(j + 1), d2 = at(j + 2);
    const uint64_t lo = ((uint64_t)d1 << 32) | d0;
    win = sh ? ((lo >> sh) | ((uint64_t)d2 << (64u - sh))) : lo;
    const int jn = nbd + (q >> 5); const uint32_t shn = (uint32_t)q & 31u;
    const uint64_t nn = ((uint64_t)at(jn + 1) << 32) | at(jn);
    uint32_t n = (uint32_t)(nn >> shn);
    const int left = L - q;
    n |= left >= 32 ? 0u : (left <= 0 ? 0xFFFFFFFFu : (0xFFFFFFFFu << left));
    nwin = n;
  }
  __device__ __forceinline__ void stream_rc() const {
    // bases: reverse the dword order, reverse the bases inside each dword, complement; then shift the read's end down to position 0
    for (int j = 0; j < nbd / 2; ++j) {
      uint32_t a = at(j), b = at(nbd - 1 - j);
      a = __brev(a); a = ~(((a & 0x55555555u) << 1) | ((a >> 1) & 0x55555555u));
      b = __brev(b); b = ~(((b & 0x55555555u) << 1) | ((b >> 1) & 0x55555555u));
      at(j) = b; at(nbd - 1 - j) = a;
    }
    { const int s = 2 * (16 * nbd - L), qd = s >> 5; const uint32_t r = (uint32_t)s & 31u;
      for (int j = 0; j < nbd; ++j) {
        const uint32_t x0 = j + qd < nbd ? at(j + qd) : 0u, x1 = j + qd + 1 < nbd ? at(j + qd + 1) : 0u;
        at(j) = r ? ((x0 >> r) | (x1 << (32u - r))) : x0;
      } }
    for (int j = 0; j < nd / 2; ++j) {
      const uint32_t a = __brev(at(nbd + j)), b = __brev(at(nbd + nd - 1 - j));
      at(nbd + j) = b; at(nbd + nd - 1 - j) = a;
    }
    { const int s = 32 * nd - L, qd = s >> 5; const uint32_t r = (uint32_t)s & 31u;
      for (int j = 0; j < nd; ++j) {
        const uint32_t x0 = j + qd < nd ? at(nbd + j + qd) : 0u, x1 = j + qd + 1 < nd ? at(nbd + j + qd + 1) : 0u;
        at(nbd + j) = r ? ((x0 >> r) | (x1 << (32u - r))) : x0;
      } }
  }
};

struct Look { uint4 e; uint32_t want; bool o; };
// PLDS: the first level of the table was staged in LDS by the workgroup; otherwise it is read from global memory (L2-resident)
template <bool PLDS>
__device__ __forceinline__ Look anchor_lookup(const AnchorHead& h, const uint8_t* s_pilots, uint64_t x, bool valid) {
  const uint64_t y = rc_kmer(x, h.K);
  Look k; k.o = x > y;
  const PHProbe pr = ph_probe(k.o ? y : x, h.ph);
  k.want = pr.want;
  const bool gok = CHK(pr.group < h.ph.n_groups, 10);
  const uint32_t pilot = PLDS ? (uint32_t)s_pilots[gok ? pr.group : 0u] : ((valid && gok) ? (uint32_t)h.pilots[pr.group] : 0u);
  uint32_t slot = valid ? ph_slot(pr, pilot, h.ph) : 0u;
  if (!CHK(slot < h.ph.n_slots, 11)) slot = 0u;
#ifdef NB_ANCHOR_PROF
  if (h.dbg & 2) slot &= 1023u;
#endif
  k.e = *reinterpret_cast<const uint4*>(&h.table[slot]);
  return k;
}

// presence / strand test only: the key word of the slot, from the 4-byte-per-slot copy (a quarter of the table: stays in the L2s)
struct LookKey { uint32_t d; bool o; };         // d = key ^ want: < 4 = the k-mer is in the table; bit 1 = on both strands; bit 0 ^ o = on the other strand
template <bool PLDS>
__device__ __forceinline__ LookKey anchor_lookup_key(const AnchorHead& h, const uint8_t* s_pilots, uint64_t x, bool valid) {
  const uint64_t y = rc_kmer(x, h.K);
  LookKey k; k.o = x > y;
  const PHProbe pr = ph_probe(k.o ? y : x, h.ph);
  const bool gok = CHK(pr.group < h.ph.n_groups, 10);
  const uint32_t pilot = PLDS ? (uint32_t)s_pilots[gok ? pr.group : 0u] : ((valid && gok) ? (uint32_t)h.pilots[pr.group] : 0u);
  uint32_t slot = valid ? ph_slot(pr, pilot, h.ph) : 0u;
  if (!CHK(slot < h.ph.n_slots, 11)) slot = 0u;
  k.d = h.keys[slot] ^ pr.want;
  return k;
}

template <int NR, int NW, bool PLDS, bool MATE, bool MM>   // MM: num_mismatches > 0 (h.M): the code that passes over bases is compiled into its own instantiations
__global__ __launch_bounds__(1024) void nb_anchor_kernel(AnchorHead h, const LaneTail* __restrict__ tp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
  __shared__ unsigned int s_reason[NB_NUM_REASONS];
  __shared__ unsigned int s_scan_n, s_retry_n;
  // counts of the workgroup's hot classes (direct-mapped by the low bits of the class id; ids are handed out in first-seen order, so the
  // classes every wave meets have small, distinct ones): with a library of a few loci every wave adds to the same handful of counters,
  // and atomics on one address serialise in the L2 — 250 sequences took 3.9 ms per 10 M reads where 500 took 3.1
  constexpr int CACHE = 256;
  __shared__ unsigned int s_ctag[CACHE], s_ccnt[CACHE];
  uint8_t* s_pilots = s_dyn;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wpb = (int)(blockDim.x >> 6);
  uint32_t* rec = reinterpret_cast<uint32_t*>(s_dyn + h.pilot_lds_bytes) + (size_t)wave * h.rec_dwords * 64;
  const bool listed = h.retry_in != nullptr;     // second pass: this workgroup's segment of the retry list
  if (threadIdx.x < NB_NUM_REASONS) s_reason[threadIdx.x] = 0;
  if (threadIdx.x == 0) { s_scan_n = listed ? h.scan_counts[blockIdx.x] : 0u; s_retry_n = 0; }
  for (int i = threadIdx.x; i < CACHE; i += blockDim.x) { s_ctag[i] = NB_NO_CLASS; s_ccnt[i] = 0u; }
  if (PLDS)
    for (int i = threadIdx.x * 16; i < h.pilot_lds_bytes; i += blockDim.x * 16)
      *reinterpret_cast<uint4*>(s_pilots + i) = *reinterpret_cast<const uint4*>(h.pilots + i);
  __syncthreads();
  const long long nwaves = (long long)gridDim.x * wpb;
  const int K = h.K;
  const uint64_t kmask = (1ull << (2 * K)) - 1;
  const uint32_t nkmask = (1u << K) - 1u;
  const int nwb = h.nwb, nwn = (h.nwb + 1) >> 1;
  constexpr uint32_t F_SEEDED = 1u, F_STARTED = 2u, F_SLOW = 4u, F_HARD = 8u, F_OPP = 16u, F_LEXT = 32u, F_ERR = 64u, F_STEP = 128u;
  constexpr uint32_t F_REDO = F_SLOW | F_HARD | F_OPP | F_LEXT | F_ERR;
#ifdef NB_ANCHOR_PROF
  unsigned long long pacc[16] = {0};
#endif

  // the packed record of the NEXT batch is loaded into registers while this one is mapped (the reads stream from HBM exactly once)
  constexpr int MAXW = 3 * NR + 2;               // words of a record whose k-mer positions fit NR mask words (max_len <= 64 NR + 30)
  uint64_t nxt[MAXW];
  const uint32_t* __restrict__ seg = listed ? h.retry_in + (size_t)blockIdx.x * (size_t)h.retry_seg : nullptr;
  const long long cnt = listed ? (long long)h.retry_counts[blockIdx.x] : h.n;                       // reads of this pass / of this workgroup's segment
  const long long first = listed ? (long long)wave * 64 : ((long long)blockIdx.x * wpb + wave) * 64;
  const long long hop = listed ? (long long)wpb * 64 : nwaves * 64;
  uint32_t nxt_r = 0;
  auto fetch = [&](long long b) {
    const long long i = b + lane;
    nxt_r = i < cnt ? (listed ? seg[i] : (uint32_t)i) : 0u;
    const uint64_t* __restrict__ rp = h.reads + (size_t)nxt_r * h.stride;
#pragma unroll
    for (int j = 0; j < MAXW; ++j) nxt[j] = (j < nwb + nwn && i < cnt) ? rp[j] : 0ull;
  };
  fetch(first);

  for (long long base = first; base < cnt; base += hop) {
    const long long idx = base + lane;
    bool active = idx < cnt;
    const uint32_t r = nxt_r;
    int L = 0;
    if (active) { L = h.lens ? (int)h.lens[r] : h.max_len; if (L > h.max_len) L = h.max_len; }
    Rec R{rec + lane, 2 * nwb, 2 * nwn, L};
#ifdef NB_ANCHOR_PROF
    long long tprof = clock64();
#endif
    {   // stage the record (this lane's column of LDS), then start the loads of the next one
#pragma unroll
      for (int j = 0; j < MAXW; ++j) if (j < nwb + nwn) { R.at(2 * j) = (uint32_t)nxt[j]; R.at(2 * j + 1) = (uint32_t)(nxt[j] >> 32); }
      R.at(2 * (nwb + nwn)) = 0u; R.at(2 * (nwb + nwn) + 1) = 0u;
      if (base + hop < cnt) fetch(base + hop);
    }
    const int last = L - K;
    bool rcf = false;                            // stream = reverse complement of the read
    uint32_t fl = 0;
    bool to_scan = false;
    PROF_MARK(0);

    // ---- orientation: which strand do the k-mers at the start, the middle and the end of the read hit? A read none of whose three
    // probes is in the table at all is most likely foreign to the library: the scan kernel settles it (no walk, no divergence here) ----
    constexpr int RDW = RetryShape<NR, NW>::DW;
    uint32_t rs[RDW];                              // (second pass: the saved walk of this lane's read)
    if (listed) {
      const uint32_t* sp = h.retry_state + ((size_t)blockIdx.x * (size_t)h.retry_seg + (size_t)(idx < cnt ? idx : 0)) * RDW;
#pragma unroll
      for (int q4 = 0; q4 < RDW / 4; ++q4) { const uint4 v = *reinterpret_cast<const uint4*>(sp + 4 * q4); rs[4 * q4] = v.x; rs[4 * q4 + 1] = v.y; rs[4 * q4 + 2] = v.z; rs[4 * q4 + 3] = v.w; }
      rcf = active && ((rs[0] >> 26) & 1u);
      if (rcf) R.stream_rc();
    } else if (last >= 0) {
      int votes = 0; bool any = false;
      LookKey pk[3]; bool pv[3];
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int p = u == 0 ? 0 : (u == 1 ? last >> 1 : last);
        uint64_t w; uint32_t nn; R.get32(p, w, nn);
        pv[u] = (nn & nkmask) == 0;
        pk[u] = anchor_lookup_key<PLDS>(h, s_pilots, w & kmask, pv[u]);
      }
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const uint32_t d = pk[u].d;
        if (pv[u] && d < 4u) { any = true; if (!(d & 2u)) votes += ((d & 1u) == (pk[u].o ? 1u : 0u)) ? 1 : -1; }
      }
      rcf = votes < 0;
      to_scan = active && !any;
      if (rcf) R.stream_rc();
    }

    PROF_MARK(1);
    // ---- the walk, in rounds. A lane is seeking a seed (SEEK, stride 3), stepping over the base its last run stopped at (STEP), following
    // the text (EXT: one 32-base comparison per round) or done. One lookup and one comparison per round at the most. ----
    constexpr int SEEK = 0, STEP = 1, EXT = 2, DONE = 3;
    int mode = (active && last >= 0 && !to_scan) ? SEEK : DONE;
    int pos = 0, cov = 0;
    uint32_t lo = 0; uint64_t a0 = 0, a1 = 0;   // the running AND of the classes: NW words from class-row word lo
    uint32_t run_t = 0, run_w0 = 0; int run_q = 0, steps = 0, maxsteps = 0;   // the run being followed: text position / payload of its first k-mer, ...
    uint32_t p_t = 0, p_n = 0, p_w0 = 0;         // the last finished run, classes not yet fetched: its two sparse-table loads are issued
                                                 // together with the lane's NEXT table lookup (one wait for both), or after the loop
    auto fold = [&](uint64_t v0, uint64_t v1, uint32_t w0) {
      const bool first = !(fl & F_STARTED);
      a0 = first ? v0 : ((w0 == lo) ? (a0 & v0) : 0ull);
      if (NW > 1) a1 = first ? v1 : ((w0 == lo) ? (a1 & v1) : 0ull);
      lo = first ? w0 : lo;
      fl |= F_STARTED;
    };
    // AND of the classes of the text k-mers [t, t + n): two overlapping power-of-two blocks of the sparse table
    auto range_and = [&](uint32_t t, uint32_t n, uint64_t& v0, uint64_t& v1) {   // (t by value: the checked build redirects it)
      const int lv = 31 - __builtin_clz(n);
      uint32_t off = n - (1u << lv);
      if (!CHK(n >= 1u && n < 256u && ((long long)lv * h.T + t + off + 1) * NW <= h.sparse_words, 13)) { t = 0u; off = 0u; }
#ifdef NB_ANCHOR_PROF
      if (h.dbg & 1) { t = 0u; off = 0u; }
#endif
      const uint64_t* sp = h.sparse + ((size_t)(n >= 1u && n < 256u ? lv : 0) * (size_t)h.T + t) * NW;
      if (NW == 1) { v0 = sp[0] & sp[off]; v1 = 0ull; }
      else {
        const uint4 x = *reinterpret_cast<const uint4*>(sp), y = *reinterpret_cast<const uint4*>(sp + 2 * (size_t)off);
        v0 = (((uint64_t)x.y << 32) | x.x) & (((uint64_t)y.y << 32) | y.x);
        v1 = (((uint64_t)x.w << 32) | x.z) & (((uint64_t)y.w << 32) | y.z);
      }
    };
    constexpr int LW = CoverShape<NR>::LW;
    uint64_t cover[LW];                         // settled seed-scan positions of the other orientation (CoverShape)
#pragma unroll
    for (int wi = 0; wi < LW; ++wi) cover[wi] = 0;
    const int res = last >= 0 ? last % 3 : 0;
    auto cover_set = [&](int j0, int j1) {      // lattice points j0 .. j1 (nothing if j1 < j0)
#pragma unroll
      for (int wi = 0; wi < LW; ++wi) {
        const int a = j0 - 64 * wi, b = j1 - 64 * wi;
        const uint64_t hi = b >= 63 ? ~0ull : (b < 0 ? 0ull : ((2ull << b) - 1)), lo_ = a <= 0 ? ~0ull : (a >= 64 ? 0ull : (~0ull << a));
        cover[wi] |= (j1 >= j0) ? (hi & lo_) : 0ull;
      }
    };
    const int L5 = L / 5;
    int ls_q = 0, ls_room = 0; uint32_t ls_t = 0;   // first seed with a possible left extension: its position, anchor and left room
    int rounds = 0;
#ifdef NB_ANCHOR_PROF
    int my_rounds = 0;
#endif
    if (listed && active) {                        // the walk resumes where the first pass left it
      constexpr int B = 2 + 2 * NW;
      cov = (int)(rs[0] & 0xFFFFu); fl = (rs[0] >> 16) & 0xFFu; mode = (int)((rs[0] >> 24) & 3u);
      lo = rs[1] & 0xFFFFu; p_w0 = rs[1] >> 16;
      a0 = ((uint64_t)rs[3] << 32) | rs[2]; if (NW > 1) a1 = ((uint64_t)rs[B - 1] << 32) | rs[B - 2];
      p_t = rs[B]; p_n = rs[B + 1] & 0xFFu; ls_q = (int)((rs[B + 1] >> 8) & 0xFFFFu); ls_room = (int)(rs[B + 1] >> 24); ls_t = rs[B + 2];
      pos = (int)(rs[B + 3] & 0xFFFFu); run_q = (int)(rs[B + 3] >> 16); run_t = rs[B + 4];
      run_w0 = rs[B + 5] & 0xFFFFu; steps = (int)((rs[B + 5] >> 16) & 0xFFu); maxsteps = (int)(rs[B + 5] >> 24);
#pragma unroll
      for (int wi = 0; wi < LW; ++wi) cover[wi] = ((uint64_t)rs[B + 7 + 2 * wi] << 32) | rs[B + 6 + 2 * wi];
    }
    // tolerated mismatches (num_mismatches = h.M > 0, SPEC 3): while the read follows the text, a base that differs (or an N) is passed over if
    // the text k-mer before it is r_internal — its unique right extension is the text's next base — and the budget allows. The read's own
    // k-mers over that base were never looked up: their seed-scan positions stay unsettled for the other orientation (cv_from).
    int mm = 0, cv_from = 0; bool er = false;      // mismatches used; first position of the run whose k-mers equal the read's; last run ended for lack of room
    uint64_t patches = 0;                          // the four latest bases passed over: position (9 bits) | read base XOR text base << 9 | 1 << 15 each (the
                                                   // walk goes on from the corrected k-mer: the next STEP lookups see the text's base there, SPEC 3's succ(x, u))
    if (listed && active) { mm = (int)(rs[0] >> 27); cv_from = (int)(rs[RDW - 1] & 0xFFFFu); er = ((rs[RDW - 1] >> 16) & 1u) != 0; patches = ((uint64_t)rs[RDW - 2] << 32) | rs[RDW - 3]; }
    uint64_t busy;
    while ((busy = __ballot(mode != DONE)) != 0) {
      if (h.retry_out && rounds >= h.retry_rounds && __popcll(busy) <= h.retry_lanes) {   // the few lanes left hold the whole wave: second pass
        // the read and its walk as it stands go on the workgroup's segment of the retry list (one LDS atomic per wave)
        const bool to_retry = mode != DONE;
        uint32_t b0 = 0;
        if (lane == 0) b0 = atomicAdd(&s_retry_n, (uint32_t)__popcll(busy));
        b0 = __builtin_amdgcn_readfirstlane(b0);
        if (to_retry) {
          const size_t slot = (size_t)blockIdx.x * (size_t)h.retry_seg + b0 + (uint32_t)__popcll(busy & ((1ull << lane) - 1));
          h.retry_out[slot] = r;
          constexpr int B = 2 + 2 * NW;
          uint32_t d[RDW];
#pragma unroll
          for (int j = 0; j < RDW; ++j) d[j] = 0u;
          d[0] = (uint32_t)(cov & 0xFFFF) | ((fl & 0xFFu) << 16) | ((uint32_t)mode << 24) | ((rcf ? 1u : 0u) << 26) | ((uint32_t)mm << 27);
          d[RDW - 1] = ((uint32_t)cv_from & 0xFFFFu) | ((er ? 1u : 0u) << 16); d[RDW - 3] = (uint32_t)patches; d[RDW - 2] = (uint32_t)(patches >> 32);
          d[1] = (lo & 0xFFFFu) | (p_w0 << 16);
          d[2] = (uint32_t)a0; d[3] = (uint32_t)(a0 >> 32);
          if (NW > 1) { d[B - 2] = (uint32_t)a1; d[B - 1] = (uint32_t)(a1 >> 32); }
          d[B] = p_t; d[B + 1] = (p_n & 0xFFu) | (((uint32_t)ls_q & 0xFFFFu) << 8) | (((uint32_t)ls_room & 0xFFu) << 24); d[B + 2] = ls_t;
          d[B + 3] = ((uint32_t)pos & 0xFFFFu) | ((uint32_t)run_q << 16); d[B + 4] = run_t;
          d[B + 5] = (run_w0 & 0xFFFFu) | (((uint32_t)steps & 0xFFu) << 16) | (((uint32_t)maxsteps & 0xFFu) << 24);
#pragma unroll
          for (int wi = 0; wi < LW; ++wi) { d[B + 6 + 2 * wi] = (uint32_t)cover[wi]; d[B + 7 + 2 * wi] = (uint32_t)(cover[wi] >> 32); }
          uint32_t* sp = h.retry_state + slot * RDW;
#pragma unroll
          for (int q4 = 0; q4 < RDW / 4; ++q4) { uint4 v; v.x = d[4 * q4]; v.y = d[4 * q4 + 1]; v.z = d[4 * q4 + 2]; v.w = d[4 * q4 + 3]; *reinterpret_cast<uint4*>(sp + 4 * q4) = v; }
          mode = DONE; p_n = 0u; fl = 0u; active = false;                      // (nothing below touches this read: no outputs, no lists)
        }
        break;
      }
#ifdef NB_ANCHOR_PROF
      my_rounds += mode != DONE ? 1 : 0;
#endif
      if (__ballot(mode == SEEK || mode == STEP)) {
        const bool act = mode == SEEK || mode == STEP;
        const bool step = mode == STEP;
        const int q = act ? (step ? pos - K + 1 : pos) : 0;           // k-mer position of this round's lookup
        uint64_t w; uint32_t nn; R.get32(q - (step ? 1 : 0), w, nn);  // (STEP: one base earlier, for the left-extension test; q >= 1 there)
        int pmax = -1;                                                 // the last corrected base inside the k-mer looked up (STEP with corrections)
        if (MM && __ballot(step && patches != 0ull)) {            // STEP continues from the CORRECTED k-mer: bases passed over take the text's value
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const uint32_t e = (uint32_t)(patches >> (16 * i)) & 0xFFFFu;
            const int pp = (int)(e & 0x1FFu), dd = pp - (q - 1);
            if (step && (e & 0x8000u) && dd >= 0 && dd <= K) { w ^= (uint64_t)((e >> 9) & 3u) << (2 * dd); nn &= ~(1u << dd); if (dd >= 1 && pp > pmax) pmax = pp; }
          }
        }
        const uint32_t pb = (uint32_t)w & 3u;
        const uint64_t x = (step ? (w >> 2) : w) & kmask;
        const bool valid = act && (((step ? (nn >> 1) : nn) & nkmask) == 0);
        const bool havep = act && p_n != 0u;
        // the pending run's two sparse-table operands: loaded unconditionally (entry 0 for lanes without a pending run) and combined only
        // AFTER the table lookup has been issued — inside a conditional block the compiler waited for them before it issued the lookup,
        // two dependent round trips per round instead of one
        uint64_t x0 = 0, x1 = 0, y0 = 0, y1 = 0;
        {
          const uint32_t n = havep ? p_n : 1u;
          const int lv = 31 - __builtin_clz(n);
          uint32_t off = n - (1u << lv), t = havep ? p_t : 0u;
          if (!CHK(n >= 1u && n < 256u && ((long long)lv * h.T + t + off + 1) * NW <= h.sparse_words, 13)) { t = 0u; off = 0u; }
#ifdef NB_ANCHOR_PROF
          if (h.dbg & 1) { t = 0u; off = 0u; }
#endif
          const uint64_t* sp = h.sparse + ((size_t)(havep && n < 256u ? lv : 0) * (size_t)h.T + t) * NW;
          if (NW == 1) { x0 = sp[0]; y0 = sp[off]; }
          else {
            const uint4 xa = *reinterpret_cast<const uint4*>(sp), ya = *reinterpret_cast<const uint4*>(sp + 2 * (size_t)off);
            x0 = ((uint64_t)xa.y << 32) | xa.x; x1 = ((uint64_t)xa.w << 32) | xa.z; y0 = ((uint64_t)ya.y << 32) | ya.x; y1 = ((uint64_t)ya.w << 32) | ya.z;
          }
        }
        const Look k = anchor_lookup<PLDS>(h, s_pilots, x, valid);
        const uint32_t d = k.e.x ^ k.want, pay = k.e.y;
        if (havep) { fold(x0 & y0, x1 & y1, p_w0); p_n = 0u; }
        const bool found = valid && d < 4u, both = found && (d & 2u);
        const bool hS = found && !both && ((d & 1u) == (k.o ? 1u : 0u));
        fl |= (both ? F_HARD : 0u) | ((found && !both && !hS) ? F_OPP : 0u);
        const bool ok = hS && (!step || ((pay >> (4u + pb)) & 1u));
        if (ok) {
          if (!step && !(fl & F_SEEDED) && q >= L5 && q > 0) {                // the first seed may have a left extension (SPEC §3): after the rounds
            fl |= F_LEXT; ls_q = q; ls_t = k.e.z; ls_room = (int)((k.e.w >> 8) & 0x7Fu);
          }
          fl = (fl & ~F_STEP) | (step ? F_STEP : 0u) | F_SEEDED;
          run_t = k.e.z; run_w0 = k.e.w >> 16; run_q = q; steps = 0; cv_from = pmax >= q ? pmax + 1 : q; er = false;   // (k-mers over a corrected base are not the read's own)
          fl |= (k.e.w & ANCHOR_WIDE) ? F_SLOW : 0u;                          // a component wider than the sparse table's entries: wave-per-read kernel
          const int room = (int)(k.e.w & 0xFFu), rem = L - (q + K);
          maxsteps = room < rem ? room : rem;
          mode = EXT;
        } else if (act) {
          if (pmax < 0) { const uint32_t j = div3((uint32_t)(q + 3 - res)); if ((int)(3u * j) == q + 3 - res) cover_set((int)j - 1, (int)j - 1); }     // (a looked-up position on the lattice; not if a corrected k-mer was looked up instead of the read's)
          if (MM && step && er && mm < h.M) fl |= F_SLOW;              // (the run ended for lack of room, not at a mismatch: whether SPEC 3 would pass over this base is not known here)
          if (step) mode = pos > last ? DONE : SEEK;                          // the walk broke at base pos: seed scan from there (SPEC §3)
          else { pos += 3; mode = pos > last ? DONE : SEEK; }
        }
      }
      if (__ballot(mode == EXT)) {
        const bool ex = mode == EXT;
        bool fin = ex;
        if (ex && steps < maxsteps) {
          uint64_t sw; uint32_t sn; R.get32(run_q + K + steps, sw, sn);
          uint32_t tb = run_t + (uint32_t)(K + steps);
          if (!CHK((long long)(tb >> 4) + 3 <= h.text_words, 12)) tb = 0u;
#ifdef NB_ANCHOR_PROF
          if (h.dbg & 4) tb &= 4095u;
#endif
          const U3 tw3 = *reinterpret_cast<const U3*>(h.text + (tb >> 4));
          const uint32_t sh = 2u * (tb & 15u);
          const uint64_t t01 = ((uint64_t)tw3.b << 32) | tw3.a;
          const uint64_t tw = sh ? ((t01 >> sh) | ((uint64_t)tw3.c << (64u - sh))) : t01;
          const uint64_t df = sw ^ tw;
          const uint64_t m2 = (df | (df >> 1)) & 0x5555555555555555ull;
          const int fm = m2 ? (__builtin_ctzll(m2) >> 1) : 32;
          const int fn = sn ? __builtin_ctz(sn) : 32;
          const int m = fm < fn ? fm : fn;
          const int room_left = maxsteps - steps;
          uint64_t rw = 0;                                                    // bit i: the text k-mer at run_t + steps + i is r_internal
          if (MM) { const uint32_t tx = run_t + (uint32_t)steps; const uint32_t* rq = h.text_rint + (tx >> 5); rw = ((((uint64_t)rq[1]) << 32) | rq[0]) >> (tx & 31u); }
          steps += m < room_left ? m : room_left;
          fin = (m < 32) || steps >= maxsteps;
          if (MM && m < 32 && m < room_left && mm < h.M && ((rw >> m) & 1ull)) {   // pass over the base at run_q + K + steps
            cover_set((int)div3((uint32_t)(cv_from - res + 2 + 3)) - 1, (int)div3((uint32_t)(run_q + steps - res + 3)) - 1);
            cv_from = run_q + steps + K + 1;                                  // (the read's k-mers over that base are not the text's)
            { const int b_ = run_q + K + steps;                                 // the base passed over: remember what the text has there
              const uint32_t old_ = (uint32_t)(patches >> 48) & 0xFFFFu;
              if ((old_ & 0x8000u) && b_ - (int)(old_ & 0x1FFu) <= K) fl |= F_SLOW;   // (a fifth one within k bases: the list holds four)
              const uint32_t delta = (uint32_t)((sw ^ tw) >> (2 * m)) & 3u;
              patches = (patches << 16) | (uint64_t)(0x8000u | (delta << 9) | ((uint32_t)b_ & 0x1FFu)); }
            steps += 1; mm += 1;
            fin = steps >= maxsteps;
          }
        }
        if (fin) {                                                            // the run [run_q, run_q + steps] is complete
          cov += ((fl & F_STEP) ? 1 : K) + steps;
          p_t = run_t; p_n = (uint32_t)steps + 1u; p_w0 = run_w0;
          er = steps >= maxsteps;
          cover_set((int)div3((uint32_t)(cv_from - res + 2 + 3)) - 1, (int)div3((uint32_t)(run_q + steps - res + 3)) - 1);   // lattice points inside [cv_from, run_q + steps]
          pos = run_q + K + steps;                                            // the base the run stopped at
          mode = pos >= L ? DONE : STEP;
        }
      }
      if (++rounds > 6 * h.max_len + 16) { fl |= (mode != DONE) ? F_ERR : 0u; mode = DONE; }   // (cannot happen: every round advances a lane; an exit every wave reaches)
    }

    if (__ballot(p_n != 0u)) {
      uint64_t s0 = ~0ull, s1 = ~0ull;
      if (p_n != 0u) { range_and(p_t, p_n, s0, s1); fold(s0, s1, p_w0); }
    }
    // ---- left extension from the first seed (SPEC §3), the common shape: the read follows the anchor's text to the left for a few bases and
    // then stops at a base that no sequence continues with (the sequencing error that killed the earlier seeds), at an N, or at its own
    // start. One comparison, one lookup (the left-extension mask of the k-mer at which the comparison stopped), one AND; if some sequence
    // does continue there, the read stays with the wave-per-read kernel. ----
    if (!MM && __ballot((fl & F_REDO) == F_LEXT)) {                      // (with tolerated mismatches the left extension is the wave-per-read kernel's)
      const bool le = (fl & F_REDO) == F_LEXT;
      int lm = 0; bool verify = false, solved = false;
      if (le) {
        const int s0 = ls_q > 32 ? ls_q - 32 : 0, na = ls_q - s0;                               // stream bases [s0, ls_q)
        uint64_t sw; uint32_t sn; R.get32(s0, sw, sn);
        const uint32_t t0 = ls_t > 32u ? ls_t - 32u : 0u; const int nb = (int)(ls_t - t0);       // text bases [t0, ls_t)
        const U3 tw3 = *reinterpret_cast<const U3*>(h.text + (t0 >> 4));
        const uint32_t sh = 2u * (t0 & 15u);
        const uint64_t t01 = ((uint64_t)tw3.b << 32) | tw3.a;
        const uint64_t tw = sh ? ((t01 >> sh) | ((uint64_t)tw3.c << (64u - sh))) : t01;
        // align both windows so that the base just left of the seed sits in the top two bits, then count equal bases from the top
        const uint64_t swt = sw << (2 * (32 - na)), twt = nb ? (tw << (2 * (32 - nb))) : ~swt;
        const uint64_t df = swt ^ twt;
        const uint64_t mm = (df | (df << 1)) & 0xAAAAAAAAAAAAAAAAull;
        const int lead = mm ? (__builtin_clzll(mm) >> 1) : 32;
        const uint32_t snt = na < 32 ? (sn << (32 - na)) : sn;                                  // N bits of [s0, ls_q), the last one on top
        const int leadn = snt ? __builtin_clz(snt) : 32;
        int cap = na < nb ? na : nb; cap = cap < ls_room ? cap : ls_room; cap = cap < 31 ? cap : 31;
        lm = lead < leadn ? lead : leadn; lm = lm < cap ? lm : cap;
        const int r = ls_q - 1 - lm;                                                            // the base the extension stopped at
        if (r < 0 || leadn == lm) solved = true;                                                // the start of the read, or an N: SPEC's loop ends
        else verify = true;                                                                     // does ANY sequence continue with that base?
      }
      if (__ballot(verify)) {
        const int qy = ls_q - lm;                                                               // k-mer where the text was left (a text k-mer, on this strand)
        uint64_t w; uint32_t nn; R.get32(verify ? qy - 1 : 0, w, nn);                           // its window, one base earlier for the candidate left extension
        const uint32_t cb = (uint32_t)w & 3u;
        const Look k = anchor_lookup<PLDS>(h, s_pilots, (w >> 2) & kmask, verify);
        const uint32_t d = k.e.x ^ k.want;
        const bool mine = verify && d < 4u && !(d & 2u) && ((d & 1u) == (k.o ? 1u : 0u));
        if (verify && mine && !((k.e.y >> (4u + cb)) & 1u)) solved = true;                      // nobody continues with that base: the extension ends here
        // (somebody does — another sequence, or this one beyond the left room / the 31 bases compared: the wave-per-read kernel walks it)
      }
      if (__ballot(solved && lm > 0)) {
        uint64_t s0v = ~0ull, s1v = ~0ull;
        if (solved && lm > 0) { range_and(ls_t - (uint32_t)lm, (uint32_t)lm, s0v, s1v); fold(s0v, s1v, lo); cov += lm; }
      }
      if (solved) fl &= ~F_LEXT;
    }
    PROF_MARK(2); PROF_COUNT(8, rounds); PROF_COUNT(9, 1);
#ifdef NB_ANCHOR_PROF
    if (active) atomicAdd(&g_anchor_hist[my_rounds < 63 ? my_rounds : 63], 1ull);
    if (lane == 0) atomicAdd(&g_anchor_hist[64 + (rounds < 63 ? rounds : 63)], 1ull);
#endif

    // ---- the other orientation: does any of ITS seed-scan positions hit? (stream positions p = last, last - 3, ... not yet known) ----
    {
      uint64_t todo[LW];                                                        // lattice points (p = res + 3 j, j <= last / 3) not settled yet
      const bool need = active && last >= 0 && !to_scan && !(fl & F_REDO);
      const int jmax = last >= 0 ? (int)div3((uint32_t)last) : -1;
#pragma unroll
      for (int wi = 0; wi < LW; ++wi) {
        const int top = jmax - 64 * wi;
        const uint64_t m = top >= 63 ? ~0ull : (top < 0 ? 0ull : ((2ull << top) - 1));
        todo[wi] = need ? (m & ~cover[wi]) : 0ull;
      }
      auto any_todo = [&]() { uint64_t v = 0;
#pragma unroll
        for (int wi = 0; wi < LW; ++wi) v |= todo[wi];
        return v != 0; };
      // Most of these positions are the k-mers over a sequencing error (the run before stopped there, the seed scan started after it):
      // consecutive, and absent for one and the same reason. ONE m-mer of the presence filter (nb_internal.h) that lies inside all the
      // k-mers at positions [p, p + S - 1] — the one that starts at p + S - 1, the first of them being the lowest unsettled point — clears
      // their lattice points together (three at S = 8) with one 4-byte load and no hash of the table; what it cannot clear (an N inside,
      // or an m-mer of the library: the tile right behind the error) takes the exact lookup below. One word of lattice points, S >= 4.
      // Only where an exact lookup is expensive — first level of the hash in global memory (!PLDS: large libraries, c5 55.4 -> 53.7 ms); with
      // the first level in LDS the extra round trip costs what the cheaper tests save (c2 2.89 against 2.84 ms; a filter over k-mers, one
      // test per position, had been slower still: 3.52 against 3.21).
      if (!PLDS && LW == 1 && h.pf_s >= 4 && !(h.dbg & 16)) {
        const int S_ = h.pf_s, M_ = K + 1 - S_;
        const uint64_t mmask_ = (1ull << (2 * M_)) - 1; const uint32_t nmm_ = (1u << M_) - 1u;
        const uint64_t tile = S_ >= 8 ? 7ull : 3ull;                            // lattice points inside S consecutive positions, from the first
        uint64_t exact = 0;
        while (__ballot(todo[0] != 0)) {
          constexpr int U = 4;
          uint32_t fw[U], fm[U]; uint64_t bits[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const bool on = todo[0] != 0;
            const int j = on ? __builtin_ctzll(todo[0]) : 0;
            bits[u] = on ? (todo[0] & (tile << j)) : 0ull;
            todo[0] &= ~bits[u];
            const int q = res + 3 * j + S_ - 1;                                 // (q + M = p + k <= L: inside the read)
            uint64_t w; uint32_t nn; R.get32(q, w, nn);
            const bool clean = on && (nn & nmm_) == 0;
            const uint32_t hh = pf_hash(w & mmask_);
            fm[u] = clean ? pf_mask(hh) : 0u; fw[u] = h.pfilter[clean ? pf_word(hh, h.pf_wbits) : 0u];
          }
#pragma unroll
          for (int u = 0; u < U; ++u) exact |= ((fw[u] & fm[u]) == fm[u]) ? bits[u] : 0ull;
        }
        todo[0] = exact;
      }
      while (__ballot(any_todo())) {
        constexpr int U = 8;
        LookKey kk[U]; bool vv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          int j = -1;
#pragma unroll
          for (int wi = 0; wi < LW; ++wi) if (j < 0 && todo[wi]) { j = wi * 64 + __builtin_ctzll(todo[wi]); todo[wi] &= todo[wi] - 1; }
          const int p = j < 0 ? 0 : res + 3 * j;
          uint64_t w; uint32_t nn; R.get32(p, w, nn);
          vv[u] = j >= 0 && (nn & nkmask) == 0;
          kk[u] = anchor_lookup_key<PLDS>(h, s_pilots, w & kmask, vv[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint32_t d = kk[u].d;
          const bool found = vv[u] && d < 4u;
          const bool mine = found && !(d & 2u) && ((d & 1u) == (kk[u].o ? 1u : 0u));
          if (found && !mine) fl |= F_OPP;
        }
        if (fl & F_OPP) {
#pragma unroll
          for (int wi = 0; wi < LW; ++wi) todo[wi] = 0;
        }
      }
    }

    PROF_MARK(3);
    if (MATE) {                                                               // paired path: the raw alignment of this mate, no decision here
      if (active) {
        uint32_t st = MATE_OK;
        if (L < K) st = MATE_SHORT; else if (to_scan) st = MATE_SCAN; else if (fl & F_REDO) st = MATE_REDO;
        MateRes mr; mr.info = (uint32_t)(cov & 0xFFFF) | ((rcf ? 1u : 0u) << 16) | (st << 17); mr.lo = lo; mr.a0 = a0; mr.a1 = NW > 1 ? a1 : 0ull;
        h.mate_out[r] = mr;
      }
      const uint64_t ms = __ballot(to_scan);
      if (ms) {
        uint32_t b0 = 0;
        if (lane == 0) b0 = atomicAdd(&s_scan_n, (uint32_t)__popcll(ms));
        b0 = __builtin_amdgcn_readfirstlane(b0);
        if (to_scan) tp->ll.rc_list[(size_t)blockIdx.x * (size_t)h.scan_seg + b0 + (uint32_t)__popcll(ms & ((1ull << lane) - 1))] = r;
      }
    } else {
    // ---- SPEC §4 per lane, outputs, lists ----
      asm volatile("" ::: "memory");
      const LaneTail& T = *tp;
      const BatchArgs& a = T.a;
      const int pmin = a.lens ? (int)a.pct_min[L] : a.pct_fixed;
      int reason = NB_TOO_SHORT, score = 0, orient = 0, mates = 0;
      uint32_t cls = NB_NO_CLASS;
      bool to_redo = false, final_ = false, call = false;
      if (active && L >= K && !to_scan) {
#if defined(NB_ANCHOR_PROF) && !defined(NB_ANCHOR_CHECK)
        PROF_COUNT(10, __popcll(__ballot((fl & F_HARD) != 0))); PROF_COUNT(11, __popcll(__ballot((fl & F_OPP) != 0))); PROF_COUNT(12, __popcll(__ballot((fl & F_LEXT) != 0)));
        PROF_COUNT(13, __popcll(__ballot((fl & F_SLOW) != 0))); PROF_COUNT(14, __popcll(__ballot((fl & F_REDO) != 0)));
#endif
        if (fl & F_REDO) to_redo = true;
        else {
          final_ = true;                                                        // the other orientation has no hit: its coverage is 0 (NO_HIT)
          if (a.out_cov) a.out_cov[r] = rcf ? ((uint64_t)(cov & 0xFFFF) << 16) : (uint64_t)(cov & 0xFFFF);
          const int cheapS = cheap_reason_p(cov, pmin, a);
          const uint32_t n = (uint32_t)__popcll(a0) + (NW > 1 ? (uint32_t)__popcll(a1) : 0u);
          int rs = 0;
          if (cheapS != 0) rs = cheapS;
          else if (n == 0) rs = NB_EMPTY_EQCLASS;
          else if (a.discard_multiple && n != 1) rs = NB_MULTIPLE_MATCH;
          else if ((rcf && a.strand_filter == NB_STRAND_FIVEPRIME) || (!rcf && a.strand_filter == NB_STRAND_THREEPRIME)) rs = NB_STRAND;
          if (rs) reason = rs;
          else { score = cov; orient = rcf ? 1 : 0; mates = 1; reason = score < a.score_filter ? NB_SCORE_FILTER : NB_CALLED; call = reason == NB_CALLED; }
        }
      } else if (active && !to_scan) { final_ = true; if (a.out_cov) a.out_cov[r] = 0ull; }
      PROF_MARK(4);
      // equal result sets inside the wave are interned and counted once: with real data most reads of a batch map to a few classes, and 64
      // lanes adding to one counter serialise in the L2 (a handful of leader rounds; whoever is left after them goes alone)
      uint32_t leader = (uint32_t)lane, mult = call ? 1u : 0u;
      bool lead = call;
      {
        uint64_t pendm = __ballot(call);
        for (int it = 0; it < 6 && pendm; ++it) {
          const int ld = __builtin_ctzll(pendm);
          const uint32_t va_lo = __builtin_amdgcn_readlane((uint32_t)a0, ld), va_hi = __builtin_amdgcn_readlane((uint32_t)(a0 >> 32), ld);
          const uint32_t vl = __builtin_amdgcn_readlane(lo, ld);
          bool eq = call && (uint32_t)a0 == va_lo && (uint32_t)(a0 >> 32) == va_hi && lo == vl;
          if (NW > 1) eq = eq && (uint32_t)a1 == (uint32_t)__builtin_amdgcn_readlane((uint32_t)a1, ld) && (uint32_t)(a1 >> 32) == (uint32_t)__builtin_amdgcn_readlane((uint32_t)(a1 >> 32), ld);
          const uint64_t same = __ballot(eq) & pendm;
          if ((same >> lane) & 1ull) { leader = (uint32_t)ld; lead = lane == ld; mult = lane == ld ? (uint32_t)__popcll(same) : 0u; }
          pendm &= ~same;
        }
      }
      if (lead && !CHK(lo < (uint32_t)T.W && (!(NW > 1 && a1) || lo + 1u < (uint32_t)T.W), 14)) lead = false;
      cls = intern_lane2(T.ct, T.W, lead, lo, a0, NW > 1 ? a1 : 0ull);
      cls = (uint32_t)__shfl((int)cls, (int)leader);
      PROF_MARK(5);
      if (lead && cls != NB_NO_CLASS) {
        const uint32_t slot = cls & (uint32_t)(CACHE - 1);
        uint32_t tag = (h.dbg & 8) ? ~cls : s_ctag[slot];                            // (NB_ANCHOR_DBG=8: every count straight to memory, A/B runs)
        if (tag == NB_NO_CLASS) { tag = atomicCAS(&s_ctag[slot], NB_NO_CLASS, cls); if (tag == NB_NO_CLASS) tag = cls; }
        if (tag == cls) atomicAdd(&s_ccnt[slot], mult);                                // (flushed once per workgroup, below)
        else atomicAdd(&T.ct.counts[cls], (unsigned long long)mult);
      }
      if (active && final_) {
        atomicAdd(&s_reason[reason], 1u);
        if (a.out_class) a.out_class[r] = (reason == NB_CALLED) ? cls : NB_NO_CLASS;
        if (a.out_info) a.out_info[r] = (uint32_t)(score & 0xFFFF) | ((uint32_t)reason << 16) | ((uint32_t)orient << 24) | ((uint32_t)mates << 25);
      }
      {   // worklists: wave ballot + prefix count, one atomic per wave and list
        const uint64_t ms = __ballot(to_scan), mredo = __ballot(to_redo);
        if (ms) {
          uint32_t b0 = 0;
          if (lane == 0) b0 = atomicAdd(&s_scan_n, (uint32_t)__popcll(ms));
          b0 = __builtin_amdgcn_readfirstlane(b0);
          if (to_scan) T.ll.rc_list[(size_t)blockIdx.x * (size_t)h.scan_seg + b0 + (uint32_t)__popcll(ms & ((1ull << lane) - 1))] = r;
        }
        if (mredo) {
          uint32_t b0 = 0;
          if (lane == 0) b0 = atomicAdd(&T.ll.counters[1], (uint32_t)__popcll(mredo));
          b0 = __builtin_amdgcn_readfirstlane(b0);
          if (to_redo) T.ll.redo_list[b0 + (uint32_t)__popcll(mredo & ((1ull << lane) - 1))] = r;
        }
      }
    }
    PROF_MARK(6);
  }
#ifdef NB_ANCHOR_PROF
  if (lane == 0) for (int i = 0; i < 16; ++i) if (pacc[i]) atomicAdd(&g_anchor_prof[i], pacc[i]);
#endif
  __syncthreads();
  if (threadIdx.x == 0) { h.scan_counts[blockIdx.x] = s_scan_n; if (h.retry_out) h.retry_counts[blockIdx.x] = s_retry_n; }
  if (!MATE)
    for (int i = threadIdx.x; i < CACHE; i += blockDim.x)
      if (s_ctag[i] != NB_NO_CLASS && s_ccnt[i]) atomicAdd(&tp->ct.counts[s_ctag[i]], (unsigned long long)s_ccnt[i]);
  if (threadIdx.x < NB_NUM_REASONS && s_reason[threadIdx.x]) atomicAdd(&tp->ct.reasons[threadIdx.x], (unsigned long long)s_reason[threadIdx.x]);
}

// The reads none of whose three probes was in the table (ll.rc_list) — for a library of a few gene families against whole-transcriptome
// reads, nearly all of the input: every seed-scan position of BOTH orientations must be shown absent (88 positions of a 150-bp read at
// k = 20). An exact lookup per position made this kernel run at the rate at which the L2s accept requests (~100 four-byte requests per read:
// 210 G requests/s, profiles/r03_tuning_notes.md). Here (presence filter, nb_internal.h) the read is cut into tiles of S k-mer positions; ONE
// m-mer per tile (m = k + 1 - S: it lies inside every k-mer of the tile) goes through
//   LDS    the one-bit-per-slot table of 2^20 bits over the library's oriented m-mers, copied into LDS by every workgroup (when the
//          library is small enough for it to discriminate: HostIndex::lfilter), then, if it passes,
//   L2     the blocked Bloom filter in global memory (one 4-byte load),
// and only the seed-scan positions of tiles that pass both (< 1 % for a foreign read) take the exact lookup (two-level perfect hash, its
// first level read from L2 here: LDS holds the filter). The record stays in REGISTERS: tile positions are the same for all lanes, so an
// m-mer is a funnel shift of three register dwords by a constant, sixteen bases per block, fully unrolled.
// Nothing found = NO_HIT in both orientations, final; anything found = redo list (wave-per-read kernel).
constexpr int SCAN_LF_WORDS = 1 << (LF_BITS_LOG2 - 5);       // 2^20 bits = 128 KB of LDS
template <int NR, int S, bool MATE>
__global__ __launch_bounds__(1024) void nb_anchor_scan_kernel(AnchorHead h, const LaneTail* __restrict__ tp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
  __shared__ unsigned int s_nohit;
  uint32_t* s_lf = reinterpret_cast<uint32_t*>(s_dyn);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wpb = (int)(blockDim.x >> 6);
  const LaneTail& T = *tp;
  const long long count = (long long)h.scan_counts[blockIdx.x];                // this workgroup's segment of the scan list (written by the same block index)
  const uint32_t* __restrict__ seg = T.ll.rc_list + (size_t)blockIdx.x * (size_t)h.scan_seg;
  const bool lds_stage = h.lfilter != nullptr;
  if (threadIdx.x == 0) s_nohit = 0;
  if (lds_stage && count > 0)
    for (int i = threadIdx.x * 4; i < SCAN_LF_WORDS; i += blockDim.x * 4)
      *reinterpret_cast<uint4*>(s_lf + i) = *reinterpret_cast<const uint4*>(h.lfilter + i);
  __syncthreads();
  const int K = h.K, M = K + 1 - S;                                             // (S = the index's pf_s: launch_anchor_t picks the instantiation)
  const uint64_t kmask = (1ull << (2 * K)) - 1, mmask = (1ull << (2 * M)) - 1;
  const uint32_t nkmask = (1u << K) - 1u, nmmask = (1u << M) - 1u;
  const int nwb = h.nwb, nwn = (h.nwb + 1) >> 1, nbd = 2 * nwb, nd = 2 * nwn;
  const uint64_t lt = (1ull << lane) - 1;
  constexpr int MAXNWB = 2 * NR + 1, MAXNWN = NR + 1, MAXBLK = 4 * NR + 1;      // (at most 64 NR positions; the last tile's m-mer may start up to S - 1 bases later)
  constexpr int TPB = 16 / S;                                                   // tiles per block of sixteen bases
  for (long long base = (long long)wave * 64; base < count; base += (long long)wpb * 64) {
    const long long idx = base + lane;
    const bool active = idx < count;
    const uint32_t r = active ? seg[idx] : 0u;
    int L = 0;
    if (active) { L = h.lens ? (int)h.lens[r] : h.max_len; if (L > h.max_len) L = h.max_len; }
    const uint64_t* __restrict__ rp = h.reads + (size_t)r * h.stride;
    uint64_t wb[MAXNWB + 2], wn[MAXNWN + 2];
#pragma unroll
    for (int j = 0; j < MAXNWB + 2; ++j) wb[j] = (active && j < nwb) ? rp[j] : 0ull;
#pragma unroll
    for (int j = 0; j < MAXNWN + 2; ++j) wn[j] = (active && j < nwn) ? rp[nwb + j] : 0ull;
    const int last = L - K;
    const bool act = active && last >= 0;
    // the seed-scan positions: p % 3 == 0 (forward) or p % 3 == last % 3 (the other orientation's), p <= last
    Mask<NR> latt;
    {
      const uint64_t PAT = 0x9249249249249249ull;
#pragma unroll
      for (int wi = 0; wi < NR; ++wi) {
        const int r0 = (3 - (wi % 3)) % 3, r1 = ((last % 3 + 3) % 3 - (wi % 3) + 3) % 3;
        uint64_t m = (PAT << r0) | (PAT << r1);
        const int top = last - wi * 64;
        m &= top >= 63 ? ~0ull : (top < 0 ? 0ull : ((2ull << top) - 1));
        latt.w[wi] = act ? m : 0ull;
      }
    }
    Mask<NR> cand;                                                              // positions of the tiles that passed both filter stages
#pragma unroll
    for (int wi = 0; wi < NR; ++wi) cand.w[wi] = 0;
    const int nblk = h.max_len >= K ? (h.max_len - K + S - 1) / 16 + 1 : 0;     // blocks that hold a tile's m-mer start
    // chunks of blocks (~10-16 tiles): first every LDS word of the chunk, then every filter word, then the verdicts — straight-line code
    // with unconditional loads (an index that does not matter is 0), so that a chunk costs two round trips whatever it holds
    constexpr int CH = S == 8 ? 5 : (S == 4 ? 3 : (S == 2 ? 2 : 1)), NT = CH * TPB;
#pragma unroll
    for (int c0 = 0; c0 < MAXBLK; c0 += CH) {
      if (c0 < nblk && __ballot(act && 16 * c0 - (S - 1) <= last)) {
        uint32_t hh[NT], li[NT], lw[NT]; uint32_t needm = 0, cleanm = 0;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int b = c0 + t / TPB, j = t % TPB;
          hh[t] = 0u; li[t] = 0u; lw[t] = ~0u;
          if (b < MAXBLK) {
            const uint32_t d0 = (uint32_t)(wb[b >> 1] >> (32 * (b & 1))), d1 = (uint32_t)(wb[(b + 1) >> 1] >> (32 * ((b + 1) & 1))), d2 = (uint32_t)(wb[(b + 2) >> 1] >> (32 * ((b + 2) & 1)));
            const uint64_t lo64 = ((uint64_t)d1 << 32) | d0, hi64 = ((uint64_t)d2 << 32) | d1;
            const uint64_t nn64 = (b & 2) ? ((wn[b >> 2] >> 32) | (wn[(b >> 2) + 1] << 32)) : wn[b >> 2];       // N bits from position 32 (b >> 1)
            const int i = S * j + S - 1, q = 16 * b + i;                        // the tile's m-mer starts at base q and clears the k-mer positions [q - (S - 1), q]
            const uint32_t sh = 2u * (uint32_t)i;
            const uint64_t x = ((((uint64_t)(uint32_t)(hi64 >> sh)) << 32) | (uint32_t)(lo64 >> sh)) & mmask;
            needm |= (act && b < nblk && q - (S - 1) <= last) ? (1u << t) : 0u;
            cleanm |= (((uint32_t)(nn64 >> (16 * (b & 1) + i)) & nmmask) == 0) ? (1u << t) : 0u;                // (an N inside the m-mer: nothing is known about the tile)
            hh[t] = pf_hash(x); li[t] = lf_index(x);
          }
        }
        if (lds_stage) {
#pragma unroll
          for (int t = 0; t < NT; ++t) lw[t] = s_lf[li[t] >> 5];
        }
        uint32_t okm = 0;                                                      // tiles still undecided after the LDS stage
#pragma unroll
        for (int t = 0; t < NT; ++t) okm |= (((lw[t] >> (li[t] & 31u)) & 1u) || !((cleanm >> t) & 1u)) ? (1u << t) : 0u;
        okm &= needm;
        uint32_t fw[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) fw[t] = h.pfilter[pf_word(hh[t], h.pf_wbits) & (((okm & cleanm) >> t) & 1u ? ~0u : 0u)];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int b = c0 + t / TPB, j = t % TPB, p0 = 16 * b + S * j;        // first position of the tile (a multiple of S: a tile never straddles a mask word)
          const uint32_t fm = pf_mask(hh[t]);
          const bool pass = ((okm >> t) & 1u) && (!((cleanm >> t) & 1u) || (fw[t] & fm) == fm);
          if (p0 < 64 * NR) cand.w[p0 >> 6] |= pass ? (((1ull << S) - 1) << (p0 & 63)) : 0ull;
        }
      }
    }
#pragma unroll
    for (int wi = 0; wi < NR; ++wi) cand.w[wi] &= latt.w[wi];
    // exact lookups for the seed-scan positions of the tiles that passed (windows straight from the record in global memory: they are rare)
    bool hit = false;
    const uint32_t* rp32 = reinterpret_cast<const uint32_t*>(rp);
    while (__ballot(any_set<NR>(cand))) {
      constexpr int U = 6;                                                      // (a tile that passed by chance has ~S / 3 * 2 seed-scan positions: one round trip for all of them)
      LookKey kk[U]; bool vv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        int p = -1;
#pragma unroll
        for (int wi = 0; wi < NR; ++wi) if (p < 0 && cand.w[wi]) { p = wi * 64 + __builtin_ctzll(cand.w[wi]); cand.w[wi] &= cand.w[wi] - 1; }
        const int q = p < 0 ? 0 : p, j = q >> 4; const uint32_t sh = 2u * ((uint32_t)q & 15u);
        const U3 t3 = *reinterpret_cast<const U3*>(rp32 + (CHK(j + 2 < nbd + nd, 15) ? j : 0));            // (dwords j .. j + 2 <= nbd + 1 < nbd + nd)
        const uint64_t lo = ((uint64_t)t3.b << 32) | t3.a;
        const uint64_t w = sh ? ((lo >> sh) | ((uint64_t)t3.c << (64u - sh))) : lo;
        const int an = q >> 5;
        const uint32_t n0 = rp32[nbd + (an < nd ? an : 0)], n1 = an + 1 < nd ? rp32[nbd + an + 1] : 0u;
        const uint32_t nn = (uint32_t)((((uint64_t)n1 << 32) | n0) >> ((uint32_t)q & 31u));
        vv[u] = p >= 0 && (nn & nkmask) == 0;
        kk[u] = anchor_lookup_key<false>(h, nullptr, w & kmask, vv[u]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) hit = hit || (vv[u] && kk[u].d < 4u);
      if (hit) {
#pragma unroll
        for (int wi = 0; wi < NR; ++wi) cand.w[wi] = 0;
      }
    }
    if (MATE) {
      if (active) { MateRes mr; mr.info = (hit ? MATE_REDO : MATE_OK) << 17; mr.lo = 0u; mr.a0 = 0ull; mr.a1 = 0ull; h.mate_out[r] = mr; }
    } else {
      const BatchArgs& a = T.a;
      if (active && !hit) {
        if (a.out_class) a.out_class[r] = NB_NO_CLASS;
        if (a.out_info) a.out_info[r] = (uint32_t)NB_NO_HIT << 16;
        if (a.out_cov) a.out_cov[r] = 0ull;
      }
      const uint64_t mno = __ballot(active && !hit);
      if (mno && lane == 0) atomicAdd(&s_nohit, (unsigned int)__popcll(mno));
      const uint64_t mredo = __ballot(active && hit);
      if (mredo) {
        uint32_t b0 = 0;
        if (lane == 0) b0 = atomicAdd(&T.ll.counters[1], (uint32_t)__popcll(mredo));
        b0 = __builtin_amdgcn_readfirstlane(b0);
        if (active && hit) T.ll.redo_list[b0 + (uint32_t)__popcll(mredo & lt)] = r;
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0 && s_nohit) atomicAdd(&tp->ct.reasons[NB_NO_HIT], (unsigned long long)s_nohit);
}

// SPEC §4 for paired templates from the two mates' raw alignments (MateRes): one pair per lane. A mate has ONE orientation with hits (the
// other is NONE: coverage 0, NO_HIT) or none at all, so the four alignments map(r1), map(rc(r1)), map(r2), map(rc(r2)) are all known; they
// are checked and combined exactly as nb_align_general_kernel does (evaluate_alignment, combine_mates). Pairs with a mate the anchor
// kernels could not settle, or whose union would span two separate word ranges (mates on different components), go on the redo list.
template <int NW>
__global__ __launch_bounds__(256) void nb_anchor_pair_kernel(const MateRes* __restrict__ m1, const MateRes* __restrict__ m2, int K, const LaneTail* __restrict__ tp) {
  __shared__ unsigned int s_reason[NB_NUM_REASONS];
  constexpr int CACHE = 256;                                  // counts of the workgroup's hot classes, as in nb_anchor_kernel
  __shared__ unsigned int s_ctag[CACHE], s_ccnt[CACHE];
  if (threadIdx.x < NB_NUM_REASONS) s_reason[threadIdx.x] = 0;
  for (int i = threadIdx.x; i < CACHE; i += blockDim.x) { s_ctag[i] = NB_NO_CLASS; s_ccnt[i] = 0u; }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const LaneTail& T = *tp;
  const BatchArgs& a = T.a;
  const long long nwaves = (long long)gridDim.x * (blockDim.x >> 6);
  struct Aln { int rs, cov; uint32_t lo; uint64_t s0, s1; };
  for (long long base = ((long long)blockIdx.x * (blockDim.x >> 6) + wave) * 64; base < a.n; base += nwaves * 64) {
    const long long r = base + lane;
    const bool active = r < a.n;
    MateRes x{}, y{};
    if (active) { x = m1[r]; y = m2[r]; }
    const uint32_t st1 = (x.info >> 17) & 3u, st2 = (y.info >> 17) & 3u;
    bool redo = active && (st1 == MATE_REDO || st1 == MATE_SCAN || st2 == MATE_REDO || st2 == MATE_SCAN);
    int L1 = 0, L2 = 0;
    if (active) {
      L1 = a.lens ? (int)a.lens[r] : a.max_len; if (L1 > a.max_len) L1 = a.max_len;
      L2 = a.mate_lens ? (int)a.mate_lens[r] : a.mate_max_len; if (L2 > a.mate_max_len) L2 = a.mate_max_len;
    }
    // alignment j: 0 = map(r1), 1 = map(rc(r1)), 2 = map(r2), 3 = map(rc(r2))
    Aln al[4];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const MateRes& v = m ? y : x; const int L = m ? L2 : L1;
      const uint32_t st = m ? st2 : st1; const int o = (int)((v.info >> 16) & 1u), c = (int)(v.info & 0xFFFFu);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        Aln& A = al[2 * m + t];
        A.cov = 0; A.lo = 0; A.s0 = 0; A.s1 = 0;
        if (st == MATE_SHORT || L < K) { A.rs = NB_TOO_SHORT; continue; }
        if (t != o || c == 0) { A.rs = NB_NO_HIT; continue; }                 // the orientation without hits (or a mate without any)
        A.cov = c; A.lo = v.lo; A.s0 = v.a0; A.s1 = NW > 1 ? v.a1 : 0ull;
        int rs = cheap_reason_p(c, (int)a.pct_min[L], a);
        if (!rs) {
          const uint32_t n = (uint32_t)__popcll(A.s0) + (uint32_t)__popcll(A.s1);
          if (n == 0) rs = NB_EMPTY_EQCLASS; else if (a.discard_multiple && n != 1) rs = NB_MULTIPLE_MATCH;
        }
        A.rs = rs;
      }
    }
    struct Ori { bool valid; int reason, score, mates; uint32_t lo; uint64_t s0, s1; };
    auto combine = [&](const Aln& p, const Aln& q) {                          // p: the r1 member of the orientation, q: the r2 member
      Ori o{false, p.rs, 0, 0, 0u, 0ull, 0ull};
      const bool v1 = p.rs == 0, v2 = q.rs == 0;
      if (v1 && v2) {
        const bool same = p.lo == q.lo;
        const uint64_t i0 = same ? (p.s0 & q.s0) : 0ull, i1 = same ? (p.s1 & q.s1) : 0ull;
        const uint32_t ni = (uint32_t)__popcll(i0) + (uint32_t)__popcll(i1);
        const bool use_inter = (a.intersect_level == 2) || (a.intersect_level == 1 && ni != 0);
        if (a.intersect_level == 2 && ni == 0) { o.reason = NB_EMPTY_EQCLASS; return o; }
        if (!use_inter && !same) redo = true;                                 // a union over two separate word ranges: wave-per-read kernel
        o.lo = p.lo; o.s0 = use_inter ? i0 : (p.s0 | q.s0); o.s1 = use_inter ? i1 : (p.s1 | q.s1);
        o.valid = true; o.reason = 0; o.score = p.cov + q.cov; o.mates = 2; return o;
      }
      if (v1 || v2) {
        if (a.require_valid_pair) { o.reason = NB_INVALID_PAIR; return o; }
        const Aln& w = v1 ? p : q;
        o.valid = true; o.reason = 0; o.score = w.cov; o.mates = 1; o.lo = w.lo; o.s0 = w.s0; o.s1 = w.s1; return o;
      }
      o.reason = p.rs > q.rs ? p.rs : q.rs; return o;
    };
    Ori F = combine(al[0], al[3]), R = combine(al[1], al[2]);                 // F = {map(r1), map(rc(r2))}, R = {map(rc(r1)), map(r2)}
    if (F.valid && a.strand_filter == NB_STRAND_THREEPRIME) { F.valid = false; F.reason = NB_STRAND; }
    if (R.valid && a.strand_filter == NB_STRAND_FIVEPRIME) { R.valid = false; R.reason = NB_STRAND; }
    int reason, score = 0, orient = 0, mates = 0;
    const bool pickR = R.valid && (!F.valid || R.score > F.score);
    bool call = false;
    if (F.valid || R.valid) {
      score = pickR ? R.score : F.score; orient = pickR ? 1 : 0; mates = pickR ? R.mates : F.mates;
      reason = score < a.score_filter ? NB_SCORE_FILTER : NB_CALLED;
      call = reason == NB_CALLED;
    } else reason = F.reason > R.reason ? F.reason : R.reason;
    const bool fin = active && !redo;
    const Ori& P = pickR ? R : F;
    uint32_t cls = intern_lane2(T.ct, T.W, fin && call, P.lo, P.s0, NW > 1 ? P.s1 : 0ull);
    if (fin && call && cls != NB_NO_CLASS) {
      const uint32_t slot = cls & (uint32_t)(CACHE - 1);
      uint32_t tag = s_ctag[slot];
      if (tag == NB_NO_CLASS) { tag = atomicCAS(&s_ctag[slot], NB_NO_CLASS, cls); if (tag == NB_NO_CLASS) tag = cls; }
      if (tag == cls) atomicAdd(&s_ccnt[slot], 1u);
      else atomicAdd(&T.ct.counts[cls], 1ull);
    }
    if (fin) {
      atomicAdd(&s_reason[reason], 1u);
      if (a.out_class) a.out_class[r] = (reason == NB_CALLED) ? cls : NB_NO_CLASS;
      if (a.out_info) a.out_info[r] = (uint32_t)(score & 0xFFFF) | ((uint32_t)reason << 16) | ((uint32_t)orient << 24) | ((uint32_t)mates << 25);
      if (a.out_cov) a.out_cov[r] = (uint64_t)(al[0].cov & 0xFFFF) | ((uint64_t)(al[1].cov & 0xFFFF) << 16) | ((uint64_t)(al[2].cov & 0xFFFF) << 32) | ((uint64_t)(al[3].cov & 0xFFFF) << 48);
    }
    const uint64_t mredo = __ballot(active && redo);
    if (mredo) {
      uint32_t b0 = 0;
      if (lane == 0) b0 = atomicAdd(&T.ll.counters[1], (uint32_t)__popcll(mredo));
      b0 = __builtin_amdgcn_readfirstlane(b0);
      if (active && redo) T.ll.redo_list[b0 + (uint32_t)__popcll(mredo & ((1ull << lane) - 1))] = (uint32_t)r;
    }
  }
  __syncthreads();
  if (threadIdx.x < NB_NUM_REASONS && s_reason[threadIdx.x]) atomicAdd(&tp->ct.reasons[threadIdx.x], (unsigned long long)s_reason[threadIdx.x]);
  for (int i = threadIdx.x; i < CACHE; i += blockDim.x)
    if (s_ctag[i] != NB_NO_CLASS && s_ccnt[i]) atomicAdd(&tp->ct.counts[s_ctag[i]], (unsigned long long)s_ccnt[i]);
}

// dynamic LDS of one workgroup: first level of the table (if it fits next to >= 8 waves of records: *plds) + record staging of up to 16
// waves (the kernels need ~110 VGPRs: 4 waves per SIMD)
constexpr int64_t ANCHOR_LDS_BUDGET = (160 << 10) - 256 - 2048;   // dynamic LDS of the anchor kernels: 160 KB minus their static LDS (reason and list
                                                                   // counters, 2 KB of hot-class counts)
static int64_t anchor_lds_bytes(const nb_ctx* c, int max_len, int* waves_per_block, bool* plds) {
  const int nwb = (max_len + 31) / 32, nwn = (max_len + 63) / 64;
  const int64_t rec = (int64_t)(2 * (nwb + nwn) + 2) * 64 * 4;
  int64_t pil = ((int64_t)c->ph.n_groups + 15) / 16 * 16;
  const int64_t budget = ANCHOR_LDS_BUDGET;
  *plds = pil + 8 * rec <= budget;
  if (const char* e = std::getenv("NB_ANCHOR_PLDS")) { if (e[0] == '0') *plds = false; }     // (A/B: first level read from L2 although it would fit)
  if (!*plds) pil = 0;
  int w = 16;                                  // (measured on c2, final kernels: 10 / 11 / 12 / 14 / 16 waves 4.43 / 4.10 / 3.80 / 3.77 / 3.41 ms: once the contended
                                               //  atomics were gone and half the lookups read the small key array, the walk scales with the waves in flight)
  if (!*plds) w = 8;                           // no first level in LDS: two workgroups of 8 waves per CU (launch_anchor_t), 4 waves per SIMD
  if (const char* e = std::getenv("NB_ANCHOR_WAVES")) { const int v = std::atoi(e); if (v >= 1 && v <= 16) w = v; }
  while (w > 0 && pil + (int64_t)w * rec > budget) --w;
  if (w < 4) return 0;
  if (waves_per_block) *waves_per_block = w;
  return pil + (int64_t)w * rec;
}

template <int NR, int S, bool MATE>
static int launch_scan_t(const AnchorHead& h, const LaneTail* d_tail, long long blocks, int waves, hipStream_t s) {
  NB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&nb_anchor_scan_kernel<NR, S, MATE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
  hipLaunchKernelGGL((nb_anchor_scan_kernel<NR, S, MATE>), dim3((unsigned)blocks), dim3((unsigned)(64 * waves)), (size_t)(h.lfilter ? SCAN_LF_WORDS * 4 : 0), s, h, d_tail);   // (same grid: block b scans block b's segment)
  NB_HIP(hipGetLastError());
  return NB_OK;
}
template <int NR, bool MATE>
static int launch_scan_s(nb_ctx* c, const AnchorHead& h, const LaneTail* d_tail, long long blocks, int waves, hipStream_t s) {
  switch (c->pf_s) {                                                  // stride of the presence filter's tiles (pf_stride(k), nb_internal.h)
    case 8: return launch_scan_t<NR, 8, MATE>(h, d_tail, blocks, waves, s);
    case 4: return launch_scan_t<NR, 4, MATE>(h, d_tail, blocks, waves, s);
    case 2: return launch_scan_t<NR, 2, MATE>(h, d_tail, blocks, waves, s);
    default: return launch_scan_t<NR, 1, MATE>(h, d_tail, blocks, waves, s);
  }
}

template <int NR, int NW, bool PLDS, bool MATE, bool MM>
static int launch_anchor_m(nb_ctx* c, const AnchorHead& h_in, const LaneTail* d_tail, int waves, int64_t lds, hipStream_t s) {
  NB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&nb_anchor_kernel<NR, NW, PLDS, MATE, MM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ANCHOR_LDS_BUDGET));
  long long blocks = c->n_cu;                                        // one workgroup per CU: the first level of the table is staged once per workgroup
  if (!PLDS && waves <= 8) blocks *= 2;                              // first level in global memory (large libraries: fabric-bound): 16 waves per CU
  if (const char* e = std::getenv("NB_ANCHOR_BLOCKS_PER_CU")) { const int v = std::atoi(e); if (v >= 1 && v <= 4) blocks = (long long)c->n_cu * v; }
  const long long need = (h_in.n + 64LL * waves - 1) / (64LL * waves);
  if (blocks > need) blocks = need;
  if (blocks < 1) blocks = 1;
  AnchorHead h = h_in;                                               // scan-list segment of a workgroup: as many entries as it has reads
  h.scan_seg = ((h.n + blocks * waves * 64 - 1) / (blocks * waves * 64)) * waves * 64;
  if (blocks * h.scan_seg > c->scan_list_cap || blocks > c->scan_counts_cap) { set_error("launch_anchor: scan list too small"); return NB_ERR_INVALID; }
  int retry_lanes = 12;                                              // (c2: 8.7 % of the reads walk for more than 12 rounds, waves for 18; with the walk's state handed over the thresholds
                                                                     //  hardly matter: 6 .. 28 lanes after 8 .. 12 rounds all give 2.98 .. 3.05 ms, profiles/r03c_retry_sweep.txt)
  if (const char* e = std::getenv("NB_ANCHOR_RETRY_LANES")) { const int v = std::atoi(e); if (v >= 0 && v < 64) retry_lanes = v; }
  long long retry_min = 64LL * 64 * waves;                           // (small batches: nothing to gain from a second launch)
  if (const char* e = std::getenv("NB_ANCHOR_RETRY_MIN")) retry_min = std::atoll(e);
  // (first level in global memory = a large library whose walk is bound by fabric requests, which a second walk of the stragglers only adds to:
  //  c5 58.8 -> 60.9 ms with the pass; c2 3.37 -> 3.19 ms, paired templates 1 249 -> 1 343 M pairs/s, c3 unchanged)
  c->anchor_retry_ran = PLDS && retry_lanes > 0 && h.n >= retry_min;
  if (c->anchor_retry_ran) {
    // a wave-batch hands over at most retry_lanes reads: that bounds a workgroup's segment of the list (ids + RetryShape::DW dwords of state each)
    h.retry_seg = (h.scan_seg / 64) * retry_lanes + 64;
    const int64_t need = (int64_t)blocks * h.retry_seg;
    constexpr int RDW = RetryShape<NR, NW>::DW;
    if (c->retry_cap < need || c->retry_state_dw < RDW || !c->d_retry_counts) {
      (void)hipFree(c->d_retry_list); (void)hipFree(c->d_retry_counts); (void)hipFree(c->d_retry_state); c->d_retry_list = nullptr; c->d_retry_counts = nullptr; c->d_retry_state = nullptr; c->retry_cap = 0;
      NB_HIP(hipMalloc((void**)&c->d_retry_list, (size_t)need * 4)); NB_HIP(hipMalloc((void**)&c->d_retry_counts, (size_t)c->scan_counts_cap * 4));
      NB_HIP(hipMalloc((void**)&c->d_retry_state, (size_t)need * RDW * 4));
      c->retry_cap = need; c->retry_state_dw = RDW;
    }
    h.retry_state = c->d_retry_state;
    h.retry_out = c->d_retry_list; h.retry_in = nullptr; h.retry_counts = c->d_retry_counts; h.retry_lanes = retry_lanes;
    h.retry_rounds = 10;                                             // (short reads — c3, 91 bp — finish in fewer rounds than that: nothing handed over, nothing lost)
    if (const char* e = std::getenv("NB_ANCHOR_RETRY_ROUNDS")) h.retry_rounds = std::atoi(e);
  }
  hipLaunchKernelGGL((nb_anchor_kernel<NR, NW, PLDS, MATE, MM>), dim3((unsigned)blocks), dim3((unsigned)(64 * waves)), (size_t)lds, s, h, d_tail);
  NB_HIP(hipGetLastError());
  if (h.retry_out) {                                                 // second pass over the stragglers: same grid, workgroup b maps its own segment
    h.retry_in = h.retry_out; h.retry_out = nullptr;
    hipLaunchKernelGGL((nb_anchor_kernel<NR, NW, PLDS, MATE, MM>), dim3((unsigned)blocks), dim3((unsigned)(64 * waves)), (size_t)lds, s, h, d_tail);
    NB_HIP(hipGetLastError());
    h.retry_in = nullptr;
  }
  return launch_scan_s<NR, MATE>(c, h, d_tail, blocks, waves, s);
  return NB_OK;
}
template <int NR, int NW, bool PLDS, bool MATE>
static int launch_anchor_t(nb_ctx* c, const AnchorHead& h, const LaneTail* d_tail, int waves, int64_t lds, hipStream_t s) {
  return h.M > 0 ? launch_anchor_m<NR, NW, PLDS, MATE, true>(c, h, d_tail, waves, lds, s) : launch_anchor_m<NR, NW, PLDS, MATE, false>(c, h, d_tail, waves, lds, s);
}
template <int NR, bool MATE>
static int launch_anchor_n(nb_ctx* c, const AnchorHead& h, const LaneTail* d_tail, int waves, int64_t lds, bool plds, hipStream_t s) {
  if (c->anchor_nw == 1) return plds ? launch_anchor_t<NR, 1, true, MATE>(c, h, d_tail, waves, lds, s) : launch_anchor_t<NR, 1, false, MATE>(c, h, d_tail, waves, lds, s);
  return plds ? launch_anchor_t<NR, 2, true, MATE>(c, h, d_tail, waves, lds, s) : launch_anchor_t<NR, 2, false, MATE>(c, h, d_tail, waves, lds, s);
}
template <bool MATE>
static int launch_anchor_nr(nb_ctx* c, int nr, const AnchorHead& h, const LaneTail* dt, int waves, int64_t lds, bool plds, hipStream_t s) {
  switch (nr) {
    case 1: return launch_anchor_n<1, MATE>(c, h, dt, waves, lds, plds, s);
    case 2: return launch_anchor_n<2, MATE>(c, h, dt, waves, lds, plds, s);
    case 3: return launch_anchor_n<3, MATE>(c, h, dt, waves, lds, plds, s);
    case 4: return launch_anchor_n<4, MATE>(c, h, dt, waves, lds, plds, s);
    default: return launch_anchor_n<5, MATE>(c, h, dt, waves, lds, plds, s);
  }
}

bool anchor_fits(const nb_ctx* c, int max_len) {
  int w = 0; bool plds = false;
  return c->has_anchor && anchor_lds_bytes(c, max_len, &w, &plds) > 0;
}

bool anchor_pilots_in_lds(const nb_ctx* c, int max_len) {
  int w = 0; bool plds = false;
  return anchor_lds_bytes(c, max_len, &w, &plds) > 0 && plds;
}

static AnchorHead make_head(const nb_ctx* c, int K, int W, const uint64_t* reads, const uint16_t* lens, int max_len, long long n, bool plds, MateRes* mate_out, int num_mismatches) {
  const int nwb = (max_len + 31) / 32, nwn = (nwb + 1) >> 1;
  return AnchorHead{c->d_atable, c->d_akeys, c->d_ph_pilots, c->ph, K, c->d_text, c->d_sparse, c->text_len, reads, lens, max_len, nwb + (max_len + 63) / 64, nwb, n,
                    plds ? (int)(((int64_t)c->ph.n_groups + 15) / 16 * 16) : 0, 2 * (nwb + nwn) + 2,
                    (c->text_len + 15) / 16 + 4, (long long)ANCHOR_LEVELS * c->text_len * c->anchor_nw, W, mate_out,
                    std::getenv("NB_ANCHOR_DBG") ? std::atoi(std::getenv("NB_ANCHOR_DBG")) : 0, c->d_scan_counts, 0, nullptr, nullptr, nullptr, 0, 0, nullptr, 0, c->d_pfilter, c->pf_wbits, c->pf_s, c->d_lfilter, c->d_text_rint, num_mismatches};
}

int launch_anchor(nb_ctx* c, int nr, const DevIndex& ix, const BatchArgs& a, const LaneLists& ll, hipStream_t s) {
  int waves = 0; bool plds = false;
  const int64_t lds = anchor_lds_bytes(c, a.max_len, &waves, &plds);
  if (lds <= 0 || !c->has_anchor) { set_error("launch_anchor: shape does not fit"); return NB_ERR_UNSUPPORTED; }
  NB_HIP(hipMemsetAsync(ll.counters, 0, 8, s));
  if (!c->d_lane_tail) NB_HIP(hipMalloc((void**)&c->d_lane_tail, sizeof(LaneTail)));
  LaneTail t{c->ct, a, ll, ix.W};
  NB_HIP(hipMemcpyAsync(c->d_lane_tail, &t, sizeof t, hipMemcpyHostToDevice, s));
  const AnchorHead h = make_head(c, ix.K, ix.W, a.reads, a.lens, a.max_len, a.n, plds, nullptr, a.num_mismatches);
  return launch_anchor_nr<false>(c, nr, h, (const LaneTail*)c->d_lane_tail, waves, lds, plds, s);
}

// Paired templates: both mates through the MATE kernels (raw alignments into mate_res[0 .. n) and mate_res[n .. 2n)), then the pair kernel.
// ll.redo_list / counters[1] receive the pairs for the wave-per-read (general) kernel.
int launch_anchor_pairs(nb_ctx* c, const DevIndex& ix, const BatchArgs& a, const LaneLists& ll, void* mate_res, hipStream_t s) {
  if (!c->has_anchor) { set_error("launch_anchor_pairs: no anchor structures"); return NB_ERR_UNSUPPORTED; }
  MateRes* mr = static_cast<MateRes*>(mate_res);
  NB_HIP(hipMemsetAsync(ll.counters, 0, 8, s));
  if (!c->d_lane_tail) NB_HIP(hipMalloc((void**)&c->d_lane_tail, sizeof(LaneTail)));
  LaneTail t{c->ct, a, ll, ix.W};
  NB_HIP(hipMemcpyAsync(c->d_lane_tail, &t, sizeof t, hipMemcpyHostToDevice, s));
  const LaneTail* dt = (const LaneTail*)c->d_lane_tail;
  for (int m = 0; m < 2; ++m) {
    const int max_len = m ? a.mate_max_len : a.max_len;
    int waves = 0; bool plds = false;
    const int64_t lds = anchor_lds_bytes(c, max_len, &waves, &plds);
    if (lds <= 0) { set_error("launch_anchor_pairs: shape does not fit"); return NB_ERR_UNSUPPORTED; }
    const int positions = max_len - ix.K + 1, nr = positions < 1 ? 1 : (positions + 63) / 64;
    const AnchorHead h = make_head(c, ix.K, ix.W, m ? a.mates : a.reads, m ? a.mate_lens : a.lens, max_len, a.n, plds, mr + (size_t)m * (size_t)a.n, a.num_mismatches);
    if (m) NB_HIP(hipMemsetAsync(ll.counters, 0, 4, s));                    // the scan list of the second mate starts empty
    const int rc = launch_anchor_nr<true>(c, nr, h, dt, waves, lds, plds, s);
    if (rc != NB_OK) return rc;
  }
  const long long blocks = std::min<long long>((long long)c->n_cu * 8, (a.n + 255) / 256);
  if (c->anchor_nw == 1) hipLaunchKernelGGL((nb_anchor_pair_kernel<1>), dim3((unsigned)(blocks < 1 ? 1 : blocks)), dim3(256), 0, s, mr, mr + a.n, ix.K, dt);
  else hipLaunchKernelGGL((nb_anchor_pair_kernel<2>), dim3((unsigned)(blocks < 1 ? 1 : blocks)), dim3(256), 0, s, mr, mr + a.n, ix.K, dt);
  NB_HIP(hipGetLastError());
  return NB_OK;
}
int64_t anchor_mate_res_bytes() { return (int64_t)sizeof(MateRes); }

}  // namespace nb

#ifdef NB_ANCHOR_PROF
extern "C" int nb_debug_anchor_prof(unsigned long long* out16) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(nb::g_anchor_prof), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
  unsigned long long z[16] = {0};
  return hipMemcpyToSymbol(HIP_SYMBOL(nb::g_anchor_prof), z, sizeof z) == hipSuccess ? 0 : -1;
}
extern "C" int nb_debug_anchor_hist(unsigned long long* out128) {
  if (hipMemcpyFromSymbol(out128, HIP_SYMBOL(nb::g_anchor_hist), sizeof(unsigned long long) * 128) != hipSuccess) return -1;
  unsigned long long z[128] = {0};
  return hipMemcpyToSymbol(HIP_SYMBOL(nb::g_anchor_hist), z, sizeof z) == hipSuccess ? 0 : -1;
}
#endif
