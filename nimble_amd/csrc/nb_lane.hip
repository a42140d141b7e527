// nb_lane.hip — the lane-per-read kernels: 64 reads per wave, one read per lane (single-end, exact mode, compact table, classes of one
// 64-bit word — i.e. every BASELINE library up to the L2-resident table sizes; everything else stays with nb_kernels.hip).
//
// Why: the wave-per-read kernel spends ~1 100 wave-instructions per read, 470 of them scalar (mask walk, control) with 64 lanes
// waiting, and uses 131 of 192 lane slots of a 150-bp read. Here a wave issues ~130 vector instructions per k-mer POSITION for 64
// reads at once (~250 per read), the walk of docs/SPEC.md §3 runs ONLINE per lane as the positions stream by, and no lane idles.
//
//   stream    every lane stages its packed record in LDS (transposed, conflict-free) and rolls its k-mer and the reverse
//             complement forward one base per step (2-bit shift-in; N bases reset a run-length counter);
//   lookup    canonical k-mer -> compact_probe -> both candidate buckets of the (2,2) cuckoo table (nb_internal.h), four positions in
//             flight per lane, then the 16-byte colour descriptor of the hit (renumbered sequences: one word per class);
//   walk      the identity R[i] = LX[i+1] (the (K+1)-mer that makes k-mer i right-extendable by the next base is the one that makes
//             k-mer i+1 left-extendable by the previous base) lets every fact of position i be computed from the bases seen so far,
//             so seed scan (stride 3), runs, re-seeding and the running AND of the visited classes advance in lock-step with the
//             stream; the opposite strand's hit / extension bits are collected as per-lane masks and walked once at the end
//             (coverage only);
//   decide    SPEC §4 per lane. A read whose better strand is the opposite one goes on a worklist (wave ballot + prefix count: hit
//             compaction) and is mapped by a second launch in reverse-complement orientation; reads the online walk cannot finish
//             exactly here (a non-empty left extension, both strands viable) go on the redo list of the wave-per-read kernel;
//   intern    per lane, lock-free, in the same class table as nb_kernels.hip: one state-machine step per loop iteration for every
//             lane (no lane ever spins inside the loop on a slot another lane of ITS OWN wave is still publishing).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <string>

#include "nb_device.h"

namespace nb {

// highest set bit at a position i <= from with (from - i) % 3 == 0 etc. come from nb_device.h (Mask<NR>, walk_masks_down): they are
// plain arithmetic on 64-bit words and work per lane as they do per wave.

// What the main loop needs, by value (SGPRs); everything the tail of a batch needs (class table, outputs, filters, lists) is read from
// device memory at the point of use (LaneTail): kernel arguments that stay live across the position loop would spill.
struct LaneHead {
  const PHEntry* table; const uint8_t* pilots; PHash ph; int K;
  const uint64_t* reads; const uint16_t* lens; int max_len, stride, nwb;
  long long n;
  const uint32_t* list; const uint32_t* list_n;      // pass 1: reads (bit 31 = map in reverse-complement orientation) and their count
  int pilot_lds_bytes, rec_dwords;                   // dynamic LDS: [pilots][per wave: rec_dwords x 64 dwords]
};

constexpr int LANE_MAX_WAVES = 12;   // = the launch bound of nb_lane_kernel (768 threads: 3 waves per SIMD, up to 168 VGPRs)

// one lookup in the perfect-hash table (nb_internal.h): pilot from LDS, ONE dwordx4 from the table
__device__ __forceinline__ uint32_t ph_lookup_slot(const LaneHead& h, const uint8_t* s_pilots, const PHProbe& pr) {
  return ph_slot(pr, (uint32_t)s_pilots[pr.group], h.ph);
}

// PASS 0: every read of the batch, orientation chosen per lane by three probes (start, middle, end of the read): the strand that hits
// is streamed, the other one is only recorded as masks. PASS 1: the reads of the worklist, in the orientation the list entry names.
template <int NR, int PASS>
__global__ __launch_bounds__(768) void nb_lane_kernel(LaneHead h, const LaneTail* __restrict__ tp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
  __shared__ unsigned int s_reason[NB_NUM_REASONS];
  __shared__ unsigned int s_ctag[COUNT_CACHE], s_ccnt[COUNT_CACHE];                        // hot class counts (nb_device.h)
  uint8_t* s_pilots = s_dyn;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wpb = (int)(blockDim.x >> 6);
  uint32_t* rec = reinterpret_cast<uint32_t*>(s_dyn + h.pilot_lds_bytes) + (size_t)wave * h.rec_dwords * 64;   // dword j of this lane's record at rec[j * 64 + lane]
  if (threadIdx.x < NB_NUM_REASONS) s_reason[threadIdx.x] = 0;
  count_cache_init(s_ctag, s_ccnt);
  for (int i = threadIdx.x * 16; i < h.pilot_lds_bytes; i += blockDim.x * 16)      // first level of the table -> LDS (16 bytes per thread and step)
    *reinterpret_cast<uint4*>(s_pilots + i) = *reinterpret_cast<const uint4*>(h.pilots + i);
  __syncthreads();
  const long long nwaves = (long long)gridDim.x * wpb;
  const int K = h.K;
  const uint64_t kmask = (1ull << (2 * K)) - 1;
  const int top = 2 * (K - 1);
  const long long count = PASS == 0 ? h.n : (long long)*h.list_n;
  const int nwb = h.nwb, nwn = (h.nwb + 1) >> 1;
  const int Pmax = h.max_len - K + 1;           // k-mer positions of the longest read of the batch

  for (long long base = ((long long)blockIdx.x * wpb + wave) * 64; base < count; base += nwaves * 64) {
    const long long idx = base + lane;
    const bool active = idx < count;
    uint32_t r = 0; bool rcf = false;            // read index; stream = reverse complement of the read
    if (active) { if (PASS == 0) r = (uint32_t)idx; else { const uint32_t e = h.list[idx]; r = e & 0x7FFFFFFFu; rcf = (e >> 31) != 0; } }
    int L = 0;
    if (active) { L = h.lens ? (int)h.lens[r] : h.max_len; if (L > h.max_len) L = h.max_len; }
    // ---- stage the record (this lane's column of LDS) ----
    {
      const uint64_t* __restrict__ rp = h.reads + (size_t)r * h.stride;
      for (int j = 0; j < nwb + nwn; ++j) {
        const uint64_t w = active ? rp[j] : 0ull;
        rec[(2 * j) * 64 + lane] = (uint32_t)w; rec[(2 * j + 1) * 64 + lane] = (uint32_t)(w >> 32);
      }
      rec[(2 * (nwb + nwn)) * 64 + lane] = 0u; rec[(2 * (nwb + nwn) + 1) * 64 + lane] = 0u;
    }
    const int last = L - K;
    uint32_t fl = 0;                             // per-lane flags kept in ONE VGPR (bools would live as 64-bit lane masks in SGPRs and spill):
    constexpr uint32_t F_SEEDED = 1u, F_STARTED = 2u, F_SLOW = 4u, F_HARD = 8u;   // F_HARD: a k-mer on both strands was met (wave-per-read kernel maps the read)
    // ---- orientation (PASS 0): which strand do the k-mers at the start, the middle and the end of the read hit? ----
    if (PASS == 0 && last >= 0) {
      int votes = 0;
      uint4 pe[3]; uint32_t pw[3], pm[3];
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int p = u == 0 ? 0 : (u == 1 ? last >> 1 : last);
        const int j0 = p >> 4, sh = (p & 15) * 2;
        const uint64_t lo64 = ((uint64_t)rec[(j0 + 1) * 64 + lane] << 32) | rec[j0 * 64 + lane];
        const uint64_t hi64 = rec[(j0 + 2) * 64 + lane];
        const uint64_t x = (sh ? ((lo64 >> sh) | (hi64 << (64 - sh))) : lo64) & kmask;
        const int n0 = 2 * nwb + (p >> 5), nsh = p & 31;
        const uint64_t nm = (((uint64_t)rec[(n0 + 1) * 64 + lane] << 32) | rec[n0 * 64 + lane]) >> nsh;
        const bool valid = (nm & ((1ull << K) - 1)) == 0;
        const uint64_t y = rc_kmer(x, K);
        const bool o = x > y;
        const PHProbe pr = ph_probe(o ? y : x, h.ph);
        pw[u] = pr.want; pm[u] = (valid ? 1u : 0u) | (o ? 2u : 0u);
        pe[u] = *reinterpret_cast<const uint4*>(&h.table[valid ? ph_lookup_slot(h, s_pilots, pr) : 0u]);
      }
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const uint32_t d = pe[u].x ^ pw[u];
        if ((pm[u] & 1u) && d < 4u && !(d & 2u)) votes += ((d & 1u) == ((pm[u] >> 1) & 1u)) ? 1 : -1;
      }
      rcf = votes < 0;
    }
    // ---- stream state: the next bases / N bits of the stream in two per-lane shift registers, refilled 16 at a time ----
    uint64_t fw = 0, rv = 0;
    int nvalid = 0;
    uint32_t prevb = 0;
    uint64_t bq = 0; uint32_t nq = 0;            // upcoming stream bases (2 bits each, next one at bit 0) / their N-or-beyond-the-read bits
    int avail = 0, snext = 0;                    // wave-uniform: bases held, stream index of the next chunk
    auto refill = [&]() {                        // append stream bases [snext, snext + 16): forward lanes read one dword of their record,
      int hi = L - 1 - snext; hi = hi < 0 ? 0 : hi;                  // reverse lanes the 16 bases ENDING at read base hi, reversed and complemented
      const int jb_f = snext >> 4, jb_r = hi >> 4, jb_r1 = jb_r > 0 ? jb_r - 1 : 0;
      const uint32_t A = rec[(rcf ? jb_r : jb_f) * 64 + lane], Bw = rec[(rcf ? jb_r1 : jb_f) * 64 + lane];
      uint32_t win = (uint32_t)((((uint64_t)A << 32) | Bw) >> (rcf ? 2 * ((hi & 15) + 1) : 32));
      uint32_t rev = __brev(win); rev = ((rev & 0x55555555u) << 1) | ((rev >> 1) & 0x55555555u);
      win = rcf ? ~rev : win;
      const int nb0 = 2 * nwb, jn_f = snext >> 5, jn_r = hi >> 5, jn_r1 = jn_r > 0 ? jn_r - 1 : 0;
      const uint32_t NA = rec[(nb0 + (rcf ? jn_r : jn_f)) * 64 + lane], NB = rec[(nb0 + (rcf ? jn_r1 : jn_f)) * 64 + lane];
      uint32_t nwin = (uint32_t)((((uint64_t)NA << 32) | NB) >> (rcf ? (hi & 31) + 17 : 32 + (snext & 16))) & 0xFFFFu;
      nwin = rcf ? (__brev(nwin) >> 16) : nwin;
      const int left = L - snext;                                     // stream positions of this chunk that lie inside the read
      nwin |= left >= 16 ? 0u : (left <= 0 ? 0xFFFFu : ((0xFFFFu << left) & 0xFFFFu));
      bq |= (uint64_t)win << (2 * avail); nq |= nwin << avail;
      avail += 16; snext += 16;
    };
    auto advance = [&]() {                       // consume the next stream base: fw / rv = the k-mer ending there and its reverse complement
      const uint32_t b = (uint32_t)bq & 3u;
      const bool isn = nq & 1u;
      bq >>= 2; nq >>= 1; --avail;
      prevb = (uint32_t)fw & 3u;                // first base of the k-mer that ended one base earlier = the base before the new k-mer
      fw = (fw >> 2) | ((uint64_t)b << top);
      rv = ((rv << 2) | (uint64_t)(3u - b)) & kmask;
      nvalid = isn ? 0 : nvalid + 1;
    };
    for (int t = 0; t < K - 1; ++t) { if (avail == 0) refill(); advance(); }

    // ---- online walk of the stream strand (docs/SPEC.md §3) ----
    int mode = 0, cnt = 0, cov = 0, cl = 0;     // mode 0 = seeking a seed, 1 = inside a run, 2 = done; cl = consecutive LX bits ending here
    uint32_t lo = 0; uint64_t a0 = 0;
    Mask<NR> HO, RO;                            // opposite strand (PASS 0 only): hit bits, right-extension bits (rR of nb_kernels.hip)
#pragma unroll
    for (int wi = 0; wi < NR; ++wi) { HO.w[wi] = 0; RO.w[wi] = 0; }

    // software pipeline over blocks of four positions, two register sets (A, B) used alternately: the table loads of the next block are
    // in flight while the current block's four walk steps run; no copies between the sets. The walk step is branch-free (selects).
    const int L5 = L / 5;
    struct Blk { uint32_t want[4], meta[4]; uint4 e[4]; };
    Blk A, B;
    auto probe_block = [&](Blk& X, int p0) {
      if (avail < 4) refill();
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        advance();
        const bool valid = nvalid >= K;
        const bool o = fw > rv;
        const PHProbe pr = ph_probe(o ? rv : fw, h.ph);
        X.want[u] = pr.want;
        const uint32_t nv = (uint32_t)(nvalid < 31 ? nvalid : 31);          // meta: run of valid bases (clamped) | o << 5 | previous base << 6
        X.meta[u] = nv | (o ? 32u : 0u) | (prevb << 6);
        X.e[u] = *reinterpret_cast<const uint4*>(&h.table[valid ? ph_lookup_slot(h, s_pilots, pr) : 0u]);
      }
    };
    uint64_t curH = 0, curR = 0;
    auto walk_block = [&](const Blk& X, int p0, int q) {
      uint32_t bitsH = 0, bitsR = 0;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int p = p0 + u;
        const uint32_t d = X.e[u].x ^ X.want[u], mt = X.meta[u], pay = X.e[u].y;
        const int nv = (int)(mt & 31u);
        const bool found = nv >= K && d < 4u;
        const bool both = found && (d & 2u);
        const bool hS = found && !both && ((d & 1u) == ((mt >> 5) & 1u));
        const bool hO = found && !both && !hS;
        fl |= both ? F_HARD : 0u;
        const bool prev_ok = nv >= K + 1;
        const uint32_t pb = (mt >> 6) & 3u;
        const bool lxS = hS && prev_ok && ((pay >> (4u + pb)) & 1u);
        if (PASS == 0) {
          const bool rrO = hO && prev_ok && ((pay >> (3u - pb)) & 1u);
          bitsH |= (hO ? 1u : 0u) << u;          // (u is a compile-time constant: one v_lshl_or each; folded into the 64-bit word once per block)
          bitsR |= (rrO ? 1u : 0u) << u;
        }
        // one step of SPEC §3 at position p, as selects
        const bool is_run = mode == 1;
        const bool cont = is_run && lxS, ended = is_run && !lxS;
        mode = ended ? ((p + K - 1 > last) ? 2 : 0) : mode;
        cnt = ended ? K - 1 : cnt;
        cl = lxS ? cl + 1 : 0;
        const bool seed = mode == 0 && cnt == 0 && hS;
        const bool lext = seed && !(fl & F_SEEDED) && p >= L5 && p > 0 && cl > 0;   // left extension: its length is exact here (coverage, cheap
        cov += (cont ? 1 : 0) + (seed ? K : 0) + (lext ? cl : 0);                   // checks); its classes are not in the running AND: such a read goes
        mode = seed ? 1 : mode;                                                     // to the redo list if it has to be EVALUATED on this strand
        cnt = mode == 0 ? (cnt == 0 ? 2 : cnt - 1) : cnt;
        const bool visit = cont || seed;
        const uint32_t w0 = pay >> 11;
        const uint64_t m = ((uint64_t)X.e[u].w << 32) | X.e[u].z;
        const bool first = visit && !(fl & F_STARTED);
        const uint64_t anded = (w0 == lo) ? (a0 & m) : 0ull;
        a0 = first ? m : (visit ? anded : a0);
        lo = first ? w0 : lo;
        fl |= (seed ? F_SEEDED : 0u) | (visit ? F_STARTED : 0u) | ((lext || (visit && (pay & (1u << 10)))) ? F_SLOW : 0u);   // (a class wider than one word: slow too)
      }
      if (PASS == 0) { curH |= (uint64_t)bitsH << q; curR |= (uint64_t)bitsR << q; }
    };
    if (Pmax > 0) probe_block(A, 0);
#pragma unroll 1
    for (int wi = 0; wi < NR; ++wi) {            // (not unrolled: 64 positions of straight-line code per mask word is enough for the instruction cache)
      curH = 0; curR = 0;
      for (int q = 0; q < 64 && wi * 64 + q < Pmax; q += 8) {
        const int p0 = wi * 64 + q;
        if (p0 + 4 < Pmax) probe_block(B, p0 + 4);
        walk_block(A, p0, q);
        if (p0 + 4 < Pmax) {
          if (p0 + 8 < Pmax) probe_block(A, p0 + 8);
          walk_block(B, p0 + 4, q + 4);
        }
      }
#pragma unroll
      for (int j = 0; j < NR; ++j) if (j == wi) { HO.w[j] = curH; RO.w[j] = curR; }
    }

    // ---- tail: the other strand's coverage from its masks (PASS 0), SPEC §4 per lane, outputs, worklists ----
    asm volatile("" ::: "memory");
    const LaneTail& T = *tp;
    const BatchArgs& a = T.a;
    const int pmin = a.lens ? (int)a.pct_min[L] : a.pct_fixed;
    int reason = NB_TOO_SHORT, score = 0, orient = 0, mates = 0;
    uint32_t cls = NB_NO_CLASS;
    bool to_other = false, to_redo = false, final_ = false, call = false;
    if (active && L >= K) {
      int covO = 0, cheapO = NB_NO_HIT;
      if (PASS == 0) {
        Mask<NR> LO, V;                          // LX of the opposite strand at i = its R at i + 1 (header)
#pragma unroll
        for (int wi = 0; wi < NR; ++wi) LO.w[wi] = (RO.w[wi] >> 1) | (wi + 1 < NR ? (RO.w[wi + 1] << 63) : 0ull);
        covO = any_set<NR>(HO) ? walk_masks_down<NR>(HO, RO, LO, L, K, V) : 0;
        cheapO = cheap_reason_p(covO, pmin, a);
        const int covF = rcf ? covO : cov, covR = rcf ? cov : covO;
        if (a.out_cov) a.out_cov[r] = (uint64_t)(covF & 0xFFFF) | ((uint64_t)(covR & 0xFFFF) << 16);
      }
      const int cheapS = cheap_reason_p(cov, pmin, a);
      if (fl & F_HARD) to_redo = true;
      else if (PASS == 0 && cheapS != 0 && cheapO != 0) { final_ = true; reason = cheapS > cheapO ? cheapS : cheapO; }
      else if (PASS == 0 && cheapS == 0 && cheapO == 0) to_redo = true;      // both strands viable: lazy two-candidate evaluation
      else if (PASS == 0 && cheapS != 0) to_other = true;                    // only the strand that was not streamed can be called
      else if (fl & F_SLOW) to_redo = true;
      else {                                                                  // evaluate the stream strand
        final_ = true;
        const uint32_t n = (uint32_t)__popcll(a0);
        int rs = 0;
        if (cheapS != 0) rs = cheapS;                                         // (pass 1 only, and then never: pass 0 saw the same coverage)
        else if (n == 0) rs = NB_EMPTY_EQCLASS;
        else if (a.discard_multiple && n != 1) rs = NB_MULTIPLE_MATCH;
        else if ((rcf && a.strand_filter == NB_STRAND_FIVEPRIME) || (!rcf && a.strand_filter == NB_STRAND_THREEPRIME)) rs = NB_STRAND;
        if (rs) reason = rs;
        else { score = cov; orient = rcf ? 1 : 0; mates = 1; reason = score < a.score_filter ? NB_SCORE_FILTER : NB_CALLED; call = reason == NB_CALLED; }
      }
    } else if (active) final_ = true;                                         // shorter than k: TOO_SHORT
    cls = intern_lane(T.ct, T.W, call, lo, a0);                               // (every lane takes part in the loop, see intern_lane)
    if (call && cls != NB_NO_CLASS) count_cache_add(s_ctag, s_ccnt, T.ct.counts, cls, 1u);
    if (active && final_) {
      atomicAdd(&s_reason[reason], 1u);
      if (a.out_class) a.out_class[r] = (reason == NB_CALLED) ? cls : NB_NO_CLASS;
      if (a.out_info) a.out_info[r] = (uint32_t)(score & 0xFFFF) | ((uint32_t)reason << 16) | ((uint32_t)orient << 24) | ((uint32_t)mates << 25);
      if (PASS == 0 && a.out_cov && L < K) a.out_cov[r] = 0ull;
    }
    {   // worklists: wave ballot + prefix count, one atomic per wave and list
      const uint64_t mo = __ballot(to_other), mredo = __ballot(to_redo);
      if (mo) {
        uint32_t b0 = 0;
        if (lane == 0) b0 = atomicAdd(&T.ll.counters[0], (uint32_t)__popcll(mo));
        b0 = __builtin_amdgcn_readfirstlane(b0);
        if (to_other) T.ll.rc_list[b0 + (uint32_t)__popcll(mo & ((1ull << lane) - 1))] = r | (rcf ? 0u : 0x80000000u);
      }
      if (mredo) {
        uint32_t b0 = 0;
        if (lane == 0) b0 = atomicAdd(&T.ll.counters[1], (uint32_t)__popcll(mredo));
        b0 = __builtin_amdgcn_readfirstlane(b0);
        if (to_redo) T.ll.redo_list[b0 + (uint32_t)__popcll(mredo & ((1ull << lane) - 1))] = r;
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < NB_NUM_REASONS && s_reason[threadIdx.x]) atomicAdd(&tp->ct.reasons[threadIdx.x], (unsigned long long)s_reason[threadIdx.x]);
  count_cache_flush(s_ctag, s_ccnt, tp->ct.counts);
}

// dynamic LDS of one workgroup for reads up to max_len: first level of the table + record staging of `waves` waves; 0 if it cannot fit
int64_t lane_lds_bytes(const nb_ctx* c, int max_len, int* waves_per_block) {
  if (!c->has_ph) return 0;
  const int nwb = (max_len + 31) / 32, nwn = (max_len + 63) / 64;
  const int64_t rec = (int64_t)(2 * (nwb + nwn) + 2) * 64 * 4, pil = ((int64_t)c->ph.n_groups + 15) / 16 * 16;
  const int64_t budget = (160 << 10) - 256 - 2048;   // (minus the kernel's static LDS: reason counters, 2 KB of hot-class counts)
  int w = 12;                                  // measured on c2: 12 waves per workgroup 8.2 ms, 16 waves 8.5 ms, 8 waves 8.5 ms
  if (const char* e = std::getenv("NB_LANE_WAVES")) { const int v = std::atoi(e); if (v >= 1 && v <= LANE_MAX_WAVES) w = v; }
  while (w > 0 && pil + (int64_t)w * rec > budget) --w;
  if (w < 8) return 0;                         // fewer than two waves per SIMD: the wave-per-read kernels take the batch
  if (waves_per_block) *waves_per_block = w;
  return pil + (int64_t)w * rec;
}

template <int NR, int PASS>
static int launch_lane_t(nb_ctx* c, const LaneHead& h, const LaneTail* d_tail, int waves, int64_t lds, hipStream_t s) {
  // (per launch: the attribute belongs to the function ON THE CURRENT DEVICE, and one process may hold contexts on several)
  NB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&nb_lane_kernel<NR, PASS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256 - 2048));
  long long blocks = c->n_cu;                                        // one workgroup per CU: the first level of the table is staged once per workgroup
  const long long need = PASS == 0 ? (h.n + 64LL * waves - 1) / (64LL * waves) : blocks;     // pass 1: the list length is only known on the device
  if (blocks > need) blocks = need;
  hipLaunchKernelGGL((nb_lane_kernel<NR, PASS>), dim3((unsigned)(blocks < 1 ? 1 : blocks)), dim3((unsigned)(64 * waves)), (size_t)lds, s, h, d_tail);
  NB_HIP(hipGetLastError());
  return NB_OK;
}

// Both passes of the lane-per-read path for one batch (the caller then runs the wave-per-read kernel over ll.redo_list).
int launch_lane(nb_ctx* c, int nr, const DevIndex& ix, const BatchArgs& a, const LaneLists& ll, hipStream_t s) {
  int waves = 0;
  const int64_t lds = lane_lds_bytes(c, a.max_len, &waves);
  if (lds <= 0) { set_error("launch_lane: shape does not fit"); return NB_ERR_UNSUPPORTED; }
  NB_HIP(hipMemsetAsync(ll.counters, 0, 8, s));
  if (!c->d_lane_tail) NB_HIP(hipMalloc((void**)&c->d_lane_tail, sizeof(LaneTail)));
  LaneTail t{c->ct, a, ll, ix.W};
  NB_HIP(hipMemcpyAsync(c->d_lane_tail, &t, sizeof t, hipMemcpyHostToDevice, s));    // (pageable source: staged by the runtime before the call returns)
  const int nwn = (a.nwb + 1) >> 1;
  LaneHead h{c->d_ph_table, c->d_ph_pilots, c->ph, ix.K, a.reads, a.lens, a.max_len, a.stride, a.nwb, a.n, ll.rc_list, ll.counters,
             (int)(((int64_t)c->ph.n_groups + 15) / 16 * 16), 2 * (a.nwb + nwn) + 2};
  const LaneTail* dt = (const LaneTail*)c->d_lane_tail;
  int rc;
  switch (nr) {
    case 1: rc = launch_lane_t<1, 0>(c, h, dt, waves, lds, s); if (rc == NB_OK) rc = launch_lane_t<1, 1>(c, h, dt, waves, lds, s); break;
    case 2: rc = launch_lane_t<2, 0>(c, h, dt, waves, lds, s); if (rc == NB_OK) rc = launch_lane_t<2, 1>(c, h, dt, waves, lds, s); break;
    case 3: rc = launch_lane_t<3, 0>(c, h, dt, waves, lds, s); if (rc == NB_OK) rc = launch_lane_t<3, 1>(c, h, dt, waves, lds, s); break;
    case 4: rc = launch_lane_t<4, 0>(c, h, dt, waves, lds, s); if (rc == NB_OK) rc = launch_lane_t<4, 1>(c, h, dt, waves, lds, s); break;
    default: rc = launch_lane_t<5, 0>(c, h, dt, waves, lds, s); if (rc == NB_OK) rc = launch_lane_t<5, 1>(c, h, dt, waves, lds, s); break;
  }
  return rc;
}

// The redo reads are gathered into a dense buffer so that the wave-per-read kernel runs on them unchanged (its batch length then comes from
// the device: BatchArgs::n_dev); its per-read results land in a dense array and are scattered back to the reads' own slots.
__global__ __launch_bounds__(256) void nb_lane_gather_kernel(const uint64_t* __restrict__ reads, const uint16_t* __restrict__ lens, int stride, const uint32_t* __restrict__ list,
                                                              const uint32_t* __restrict__ n_dev, uint64_t* __restrict__ out_reads, uint16_t* __restrict__ out_lens) {
  const long long n = (long long)*n_dev, total = n * stride;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long t = i / stride; const int w = (int)(i - t * stride);
    out_reads[i] = reads[(size_t)list[t] * stride + w];
    if (w == 0 && lens) out_lens[t] = lens[list[t]];
  }
}
__global__ __launch_bounds__(256) void nb_lane_scatter_kernel(const uint32_t* __restrict__ list, const uint32_t* __restrict__ n_dev, const uint32_t* __restrict__ tmp,
                                                               long long cap, uint32_t* out_class, uint32_t* out_info, uint64_t* out_cov) {
  const long long n = (long long)*n_dev;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const uint32_t r = list[i];
    if (out_class) out_class[r] = tmp[i];
    if (out_info) out_info[r] = tmp[cap + i];
    if (out_cov) out_cov[r] = reinterpret_cast<const uint64_t*>(tmp + 2 * cap)[i];
  }
}

int launch_lane_gather(nb_ctx* c, const BatchArgs& a, const LaneLists& ll, uint64_t* redo_reads, uint16_t* redo_lens, hipStream_t s) {
  hipLaunchKernelGGL(nb_lane_gather_kernel, dim3((unsigned)(c->n_cu * 4)), dim3(256), 0, s, a.reads, a.lens, a.stride, ll.redo_list, ll.counters + 1, redo_reads, redo_lens);
  NB_HIP(hipGetLastError());
  return NB_OK;
}
int launch_lane_scatter(nb_ctx* c, const BatchArgs& a, const LaneLists& ll, const uint32_t* redo_out, hipStream_t s) {
  hipLaunchKernelGGL(nb_lane_scatter_kernel, dim3((unsigned)(c->n_cu * 2)), dim3(256), 0, s, ll.redo_list, ll.counters + 1, redo_out, a.n, a.out_class, a.out_info, a.out_cov);
  NB_HIP(hipGetLastError());
  return NB_OK;
}

}  // namespace nb
