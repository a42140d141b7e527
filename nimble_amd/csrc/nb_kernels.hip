// nb_kernels.hip — hand-written gfx950 kernels of the pseudoalignment hot path + the C-ABI that launches them.
//
// Replaces the per-read work of the Rust `nimble-aligner` binary spawned at /root/reference/nimble/__main__.py:195
// (k-mer extraction, index lookup, equivalence-class intersection, scoring, filtering, counting). Semantics:
// docs/SPEC.md. Formulation: ONE READ PER WAVE, ONE K-MER POSITION PER LANE (three consecutive positions
// per lane for reads with 129..192 positions).
//   1. every lane extracts its k-mer from the 2-bit packed read, canonicalises it and reads its 32-byte home bucket of
//      the canonical-k-mer table; the slot carries the payload of BOTH strands, so one lookup serves read and reverse
//      complement;
//   2. wave ballots turn per-position facts into bit masks (hit, right-extendable, left-extendable);
//   3. the SPEC §3 walk (stride-3 seeding, left extension, runs, re-seeding) becomes a handful of find-first-bit
//      operations on those masks, executed uniformly by the wave (the reverse strand is walked downwards over the same
//      read-position order, no mask reversal);
//   4. the colours of the visited positions are intersected from their 32-byte span descriptors: the common word
//      window is the intersection of the spans (two wave reductions), and per word of that window a wave AND-reduction
//      folds the lanes' class words (one memory round trip for classes that live inside one locus);
//   5. the result set is interned in a lock-free class table (64-bit content hash + full compare) and counted.
// Integer/bit work only: no MFMA. Wave = 64 everywhere.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>


#include "nb_device.h"

namespace nb {

// per-lane facts of one k-mer position for both strands
struct PosFacts {
  uint32_t payF, payR;   // payload for the read strand / for the reverse-complement strand (PAY_ABSENT if none)
  bool hF, rF, lF, hR, rR, lR;
};

template <bool CT>
__device__ __forceinline__ PosFacts position_facts(const DevIndex& ix, const uint64_t* __restrict__ rp, int nwb, int L, int i, bool active) {
  PosFacts f; f.payF = PAY_ABSENT; f.payR = PAY_ABSENT; f.hF = f.rF = f.lF = f.hR = f.rR = f.lR = false;
  if (!active) return f;
  const int K = ix.K;
  const uint64_t* np = rp + nwb;
  // window of 32 bases starting at base i, and the N bits of bases i..i+K
  const int wi = i >> 5, sh = (i & 31) * 2;
  const uint64_t lo = rp[wi], hi = (wi + 1 < nwb) ? rp[wi + 1] : 0ull;
  const uint64_t win = sh ? ((lo >> sh) | (hi << (64 - sh))) : lo;
  const int nwi = i >> 6, nsh = i & 63, nwn = (nwb + 1) >> 1;
  const uint64_t nlo = np[nwi], nhi = (nwi + 1 < nwn) ? np[nwi + 1] : 0ull;
  const uint64_t nwin = nsh ? ((nlo >> nsh) | (nhi << (64 - nsh))) : nlo;
  const bool validk = (nwin & ((1ull << K) - 1)) == 0;
  if (!validk) return f;
  const uint64_t x = win & ((1ull << (2 * K)) - 1);
  const bool next_ok = (i + K < L) && !((nwin >> K) & 1);
  const uint32_t nextb = (uint32_t)(win >> (2 * K)) & 3u;       // K <= 31 keeps this inside the 64-bit window
  bool prev_ok = false; uint32_t prevb = 0;
  if (i >= 1) {
    const int j = i - 1;
    prevb = (uint32_t)(rp[j >> 5] >> ((j & 31) * 2)) & 3u;
    prev_ok = !((np[j >> 6] >> (j & 63)) & 1);
  }
  const uint64_t y = rc_kmer(x, K);
  const uint64_t canon = x < y ? x : y;
  const int o = (x != canon) ? 1 : 0;
  if constexpr (CT) {
    const CProbe p = compact_probe(canon, ix.chash);
    const uint4 a = load_cbucket(ix.ctable, p.b1), c = load_cbucket(ix.ctable, p.b2);
    compact_match(a, c, p.want, (uint32_t)o, true, f.payF, f.payR);
  } else {
    uint32_t b = kmer_bucket(canon, ix.hash), p0 = PAY_ABSENT, p1 = PAY_ABSENT;
    for (;;) {
      bool more;
      if (probe_bucket(ix.buckets, b, canon, p0, p1, more)) { f.payF = o ? p1 : p0; f.payR = o ? p0 : p1; break; }
      if (!more) break;
      b = (b + 1 == ix.hash.n_buckets) ? 0u : b + 1;
    }
  }
  if (f.payF != PAY_ABSENT) {
    f.hF = true;
    f.rF = next_ok && ((f.payF >> nextb) & 1u);
    f.lF = prev_ok && ((f.payF >> (4 + prevb)) & 1u);
  }
  if (f.payR != PAY_ABSENT) {           // rc strand: its next base is comp(prev), its previous base is comp(next)
    f.hR = true;
    f.rR = prev_ok && ((f.payR >> (3u - prevb)) & 1u);
    f.lR = next_ok && ((f.payR >> (4 + (3u - nextb))) & 1u);
  }
  return f;
}

// One mapped strand of one sequence: coverage and the visited positions with their colours (lane-resident).
template <int NR> struct Strand {
  int cov;             // 0 = NONE
  Mask<NR> V;          // visited positions, in read-position (lane) order
  uint32_t col[NR];    // this lane's colour at position lane + 64 r on this strand
};

// Step 4: eq = AND of the colour sets of the visited positions, from the colours' span descriptors (ColDesc). Every
// visited lane holds the descriptors of its own positions; the common window [lo, hi) of word indices is the
// intersection of the spans (max of the starts, min of the ends — two wave reductions); for each word of that window
// every lane contributes its colour's word (inline for the first two words of a span, else one load from `packed`) and a
// wave AND-reduction folds them. Typical reads (all classes inside one locus) need ONE memory round trip here (the
// descriptors) instead of the seed-row + per-word gather of a dense-row layout. Prefetching the descriptors before the
// walk was tried and lost (112 VGPRs -> 4 waves/SIMD, 18.8 ms vs 17.8 ms). acc[t] (lane l) = word l + 64 t. Returns |eq|.
struct DescRegs { uint32_t w0, nw, off; uint64_t m0, m1; };

__device__ __forceinline__ DescRegs load_desc(const DevIndex& ix, uint32_t colour) {
  const uint4* p = reinterpret_cast<const uint4*>(&ix.desc[colour]);
  const uint4 a = p[0], b = p[1];
  DescRegs d; d.w0 = a.x; d.nw = a.y; d.off = a.z; d.m0 = ((uint64_t)b.y << 32) | b.x; d.m1 = ((uint64_t)b.w << 32) | b.z;
  return d;
}
// bit `lane` of a wave-UNIFORM 64-bit mask as a per-lane boolean: the mask's SGPR pair is used as the lane mask of a
// v_cndmask directly (no shift / and / compare per test)
__device__ __forceinline__ bool lane_bit(uint64_t uniform_mask) { return __builtin_amdgcn_inverse_ballot_w64(uniform_mask); }

// descriptors of this lane's positions on strand S; lanes/rounds without a hit read descriptor 0 (valid memory, unused)
template <int NR>
__device__ __forceinline__ void load_descs(const DevIndex& ix, const Strand<NR>& S, const Mask<NR>& hit, int lane, DescRegs (&d)[NR]) {
#pragma unroll
  for (int r = 0; r < NR; ++r) d[r] = load_desc(ix, lane_bit(hit.w[r]) ? S.col[r] : 0u);
}

template <int NR, int TW>
__device__ __forceinline__ uint32_t intersect_descs(const DevIndex& ix, const Strand<NR>& S, const DescRegs (&d)[NR], int lane, uint64_t (&acc)[TW]) {
  uint32_t lo_l = 0u, hi_l = 0xFFFFFFFFu;
#pragma unroll
  for (int r = 0; r < NR; ++r) if (lane_bit(S.V.w[r])) { lo_l = d[r].w0 > lo_l ? d[r].w0 : lo_l; const uint32_t e = d[r].w0 + d[r].nw; hi_l = e < hi_l ? e : hi_l; }
  const uint32_t lo = wave_max_u32(lo_l), hi = wave_min_u32(hi_l);
#pragma unroll
  for (int t = 0; t < TW; ++t) acc[t] = 0ull;
  uint32_t n = 0;
  for (uint32_t j = lo; j < hi; ++j) {            // wave-uniform; one word for components of <= 64 sequences, two up to 128
    uint64_t v = ~0ull; bool need = false;
#pragma unroll
    for (int r = 0; r < NR; ++r) if (lane_bit(S.V.w[r])) { const uint32_t idx = j - d[r].w0; need |= idx >= 2u; v &= idx == 0u ? d[r].m0 : (idx == 1u ? d[r].m1 : ~0ull); }
    if (__ballot(need)) {                          // some class is wider than its two inline words here
#pragma unroll
      for (int r = 0; r < NR; ++r) if (lane_bit(S.V.w[r])) { const uint32_t idx = j - d[r].w0; if (idx >= 2u) v &= ix.packed[(size_t)d[r].off + idx]; }
    }
    v = wave_and_u64(v);
    n += (uint32_t)__popcll(v);
#pragma unroll
    for (int t = 0; t < TW; ++t) if ((uint32_t)t == (j >> 6) && (uint32_t)lane == (j & 63u)) acc[t] = v;
  }
  return n;
}

// on-demand form (general kernel, and the non-prefetched strand of the hot kernel)
template <int NR, int TW>
__device__ __forceinline__ uint32_t intersect_visited(const DevIndex& ix, const Strand<NR>& S, int lane, uint64_t (&acc)[TW]) {
  DescRegs d[NR];
  load_descs<NR>(ix, S, S.V, lane, d);
  return intersect_descs<NR, TW>(ix, S, d, lane, acc);
}

// Steps 1-3 of the hot kernel for one read: the whole packed record is fetched with wave-uniform (scalar) loads first,
// every round's k-mer is prepared, then the bucket loads of ALL rounds are issued together (one dependent memory round
// trip for the lookups of a read), ballots build the masks and both strands are walked. The general kernel reaches the
// same facts through position_facts(); tests hold both kernels to the oracle.
template <int NR, bool CT>
__device__ __forceinline__ int map_both_strands_fast(const DevIndex& ix, const uint64_t* __restrict__ rp, int nwb, int L, int lane,
                                                     Strand<NR>& F, Strand<NR>& R, uint4 (*lds)[64] = nullptr) {
  constexpr int NWB = 2 * NR + 1, NWN = NR + 1;
  const int K = ix.K, P = L - K + 1, nwn = (nwb + 1) >> 1;
  uint64_t w[NWB + 1], nm[NWN + 1];
#pragma unroll
  for (int j = 0; j < NWB; ++j) w[j] = (j < nwb) ? rp[j] : 0ull;
  w[NWB] = 0;
#pragma unroll
  for (int j = 0; j < NWN; ++j) nm[j] = (j < nwn) ? rp[nwb + j] : 0ull;
  nm[NWN] = 0;
  const bool hiHalf = lane >= 32;
  const int sh = (lane & 31) * 2;
  const uint64_t kmask = (1ull << (2 * K)) - 1, nkmask = (1ull << K) - 1;
  uint64_t canon[CT ? 1 : NR]; uint32_t bk[NR]; uint32_t bk2[CT ? NR : 1]; uint32_t want[CT ? NR : 1]; uint32_t meta[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int i = lane + 64 * r;
    const uint64_t lo = hiHalf ? w[2 * r + 1] : w[2 * r], hi = hiHalf ? w[2 * r + 2] : w[2 * r + 1];
    const uint64_t win = sh ? ((lo >> sh) | (hi << (64 - sh))) : lo;
    const uint64_t nlo = nm[r], nhi = nm[r + 1];
    const uint64_t nwin = lane ? ((nlo >> lane) | (nhi << (64 - lane))) : nlo;
    const bool validk = (i < P) && ((nwin & nkmask) == 0);
    const uint64_t x = win & kmask;
    const bool next_ok = (i + K < L) && !((nwin >> K) & 1);
    const uint32_t nextb = (uint32_t)(win >> (2 * K)) & 3u;
    uint32_t prevb; bool prevN;
    if (lane & 31) prevb = (uint32_t)(lo >> (sh - 2)) & 3u;
    else prevb = (uint32_t)((hiHalf ? w[2 * r] : (r ? w[2 * r - 1 < 0 ? 0 : 2 * r - 1] : 0ull)) >> 62) & 3u;
    if (lane) prevN = (nlo >> (lane - 1)) & 1;
    else prevN = r ? ((nm[r - 1 < 0 ? 0 : r - 1] >> 63) & 1) : false;
    const bool prev_ok = (i >= 1) && !prevN;
    const uint64_t y = rc_kmer(x, K);
    const uint64_t c = x < y ? x : y;
    if constexpr (CT) {
      const CProbe pr = compact_probe(c, ix.chash);
      want[r] = pr.want; bk[r] = validk ? pr.b1 : 0u; bk2[r] = validk ? pr.b2 : 0u;
    } else {
      canon[r] = c;
      bk[r] = validk ? kmer_bucket(c, ix.hash) : 0u;   // idle lanes read bucket 0: every load is unconditional, so the
    }                                                   // key halves of all NR buckets of a read are in flight together
    meta[r] = (validk ? 1u : 0u) | ((x != c) ? 2u : 0u) | (next_ok ? 4u : 0u) | (nextb << 3) | (prev_ok ? 32u : 0u) | (prevb << 6);
  }
  // bucket lookup (nb_internal.h): both 16-byte slots of the home bucket, key and payload together
  const Bucket* __restrict__ B = ix.buckets;
  uint4 ka[NR], kb[NR];
  uint32_t payF[NR], payR[NR];
  if constexpr (CT) {   // compact table: home and alternate bucket, one dwordx4 each, all rounds in flight together; no chain to follow
#pragma unroll
    for (int r = 0; r < NR; ++r) { ka[r] = load_cbucket(ix.ctable, bk[r]); kb[r] = load_cbucket(ix.ctable, bk2[r]); }
#pragma unroll
    for (int r = 0; r < NR; ++r) compact_match(ka[r], kb[r], want[r], (meta[r] >> 1) & 1u, meta[r] & 1u, payF[r], payR[r]);
  } else {
#pragma unroll
  for (int r = 0; r < NR; ++r) { const uint4* kp = reinterpret_cast<const uint4*>(&B[bk[r]]); ka[r] = kp[0]; kb[r] = kp[1]; }
  bool ovf[NR]; bool any_ovf = false;
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const uint64_t k0 = ((uint64_t)ka[r].y << 32) | ka[r].x, k1 = ((uint64_t)kb[r].y << 32) | kb[r].x;
    const bool valid = meta[r] & 1u, o = meta[r] & 2u;
    const bool m0 = (k0 & KEY_MASK) == canon[r], m1 = k1 == canon[r];
    const bool found = valid && (m0 || m1);
    ovf[r] = valid && !found && (k0 >> 63);
    any_ovf |= ovf[r];
    const uint32_t p0 = m0 ? ka[r].z : kb[r].z, p1 = m0 ? ka[r].w : kb[r].w;
    payF[r] = found ? (o ? p1 : p0) : PAY_ABSENT;
    payR[r] = found ? (o ? p0 : p1) : PAY_ABSENT;
  }
  if (__ballot(any_ovf)) {   // rare (load <= 0.10): some key overflowed its home bucket, follow the chain
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      uint32_t b = kmer_bucket(canon[r], ix.hash); bool pending = ovf[r];
      while (pending) {
        b = (b + 1 == ix.hash.n_buckets) ? 0u : b + 1;
        uint32_t p0, p1; bool more;
        if (probe_bucket(B, b, canon[r], p0, p1, more)) { const bool o = meta[r] & 2u; payF[r] = o ? p1 : p0; payR[r] = o ? p0 : p1; pending = false; }
        else pending = more;
      }
    }
  }
  }
  // one strand after the other (fewer masks live at once); a read maps to one strand at most, so the strand without hits
  // stops at its empty hit mask and never builds its extension masks. With `lds`, the colour descriptors of the strand that
  // has hits go straight into this wave's LDS rows behind the walk (see tri_finish); returns that strand (0 none, 1 F, 2 R).
  int pre = 0;
  auto fetch_descs = [&](const Mask<NR>& H, const uint32_t (&col)[NR]) {
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const uint32_t colour = lane_bit(H.w[r]) ? col[r] : 0u;
      const char* g = reinterpret_cast<const char*>(&ix.desc[colour]);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)&lds[2 * r][0], 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + 16), (__attribute__((address_space(3))) void*)&lds[2 * r + 1][0], 16, 0, 0);
    }
  };
  {
    Mask<NR> HF, RF, LF;
#pragma unroll
    for (int r = 0; r < NR; ++r) { F.col[r] = payF[r] >> 10; HF.w[r] = __ballot(payF[r] != PAY_ABSENT); }
    if (any_set<NR>(HF)) {
      if (lds) { pre = 1; fetch_descs(HF, F.col); }
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const bool next_ok = meta[r] & 4u, prev_ok = meta[r] & 32u;
        const uint32_t nextb = (meta[r] >> 3) & 3u, prevb = (meta[r] >> 6) & 3u;
        const bool hF = payF[r] != PAY_ABSENT;
        RF.w[r] = __ballot(hF && next_ok && ((payF[r] >> nextb) & 1u));
        LF.w[r] = __ballot(hF && prev_ok && ((payF[r] >> (4 + prevb)) & 1u));
      }
      F.cov = walk_masks<NR>(HF, RF, LF, L, K, F.V);
    } else {
      F.cov = 0;
#pragma unroll
      for (int r = 0; r < NR; ++r) F.V.w[r] = 0;
    }
  }
  {
    Mask<NR> HR, RR, LR;
#pragma unroll
    for (int r = 0; r < NR; ++r) { R.col[r] = payR[r] >> 10; HR.w[r] = __ballot(payR[r] != PAY_ABSENT); }
    if (any_set<NR>(HR)) {
      if (lds && !pre) { pre = 2; fetch_descs(HR, R.col); }
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const bool next_ok = meta[r] & 4u, prev_ok = meta[r] & 32u;
        const uint32_t nextb = (meta[r] >> 3) & 3u, prevb = (meta[r] >> 6) & 3u;
        const bool hR = payR[r] != PAY_ABSENT;
        RR.w[r] = __ballot(hR && prev_ok && ((payR[r] >> (3u - prevb)) & 1u));
        LR.w[r] = __ballot(hR && next_ok && ((payR[r] >> (4 + (3u - nextb))) & 1u));
      }
      R.cov = walk_masks_down<NR>(HR, RR, LR, L, K, R.V);
    } else {
      R.cov = 0;
#pragma unroll
      for (int r = 0; r < NR; ++r) R.V.w[r] = 0;
    }
  }
  return pre;
}

// ------------------------------------------------------------------------------------------------------------
// "Three positions per lane" front end for reads with 129..192 k-mer positions (150-bp reads at k = 20 have 131):
// lane l owns positions 3l, 3l+1, 3l+2. One window extraction and ONE reverse complement serve the three k-mers, and —
// the point — every mask is a single 64-bit word per residue class (bit l of m[r] <-> position 3l + r) instead of a
// three-word bit string: the SPEC §3 walk needs no multiword arithmetic, and the stride-3 seed scan of SPEC §3 is a
// plain find-first-bit on one word because a residue class IS a word. Downstream code is untouched: Strand<3>.V.w[r] /
// col[r] simply mean "residue r" instead of "round r".
struct Tri { uint64_t m[3]; };

// Positions are kept as (lane, residue) pairs so the walk never divides by three; lin() gives the linear position.
struct TPos { int l, r; };
__device__ __forceinline__ int lin(TPos p) { return 3 * p.l + p.r; }
__device__ __forceinline__ uint64_t tri_word(const Tri& t, int r) { return r == 0 ? t.m[0] : (r == 1 ? t.m[1] : t.m[2]); }
__device__ __forceinline__ TPos tpos_add(TPos p, int dq, int dr) {   // p + (3 dq + dr), 0 <= dr < 3
  int r = p.r + dr, l = p.l + dq;
  if (r >= 3) { r -= 3; ++l; }
  return TPos{l, r};
}
__device__ __forceinline__ TPos tpos_sub(TPos p, int dq, int dr) {   // p - (3 dq + dr); l < 0 when the result is negative
  int r = p.r - dr, l = p.l - dq;
  if (r < 0) { r += 3; --l; }
  return TPos{l, r};
}
// first position >= from with from's residue whose bit is set; l = -1 if none
__device__ __forceinline__ TPos tri_ffs_stride3(const Tri& H, TPos from) {
  const uint64_t m = from.l >= 64 ? 0ull : (tri_word(H, from.r) >> from.l);
  return TPos{m ? from.l + __builtin_ctzll(m) : -1, from.r};
}
// last position <= from with from's residue whose bit is set; l = -1 if none
__device__ __forceinline__ TPos tri_fls_stride3_down(const Tri& H, TPos from) {
  const uint64_t w = tri_word(H, from.r);
  const uint64_t m = from.l >= 63 ? w : (w & ((2ull << from.l) - 1));
  return TPos{m ? 63 - __builtin_clzll(m) : -1, from.r};
}
// first position >= p whose bit is ZERO (a zero is guaranteed at or before position 191)
__device__ __forceinline__ TPos tri_ffz_from(const Tri& X, TPos p) {
#pragma unroll
  for (int rr = 0; rr < 3; ++rr) if (rr >= p.r && !((X.m[rr] >> p.l) & 1)) return TPos{p.l, rr};
  if (p.l >= 63) return TPos{64, 0};
  const uint64_t m = ~(X.m[0] & X.m[1] & X.m[2]) >> (p.l + 1);
  if (!m) return TPos{64, 0};
  const int zl = p.l + 1 + __builtin_ctzll(m);
  return TPos{zl, !((X.m[0] >> zl) & 1) ? 0 : (!((X.m[1] >> zl) & 1) ? 1 : 2)};
}
// last position <= p whose bit is ZERO (position 0 is guaranteed zero in the masks passed here)
__device__ __forceinline__ TPos tri_flz_upto(const Tri& X, TPos p) {
#pragma unroll
  for (int rr = 2; rr >= 0; --rr) if (rr <= p.r && !((X.m[rr] >> p.l) & 1)) return TPos{p.l, rr};
  const uint64_t m = p.l ? (~(X.m[0] & X.m[1] & X.m[2]) & ((1ull << p.l) - 1)) : 0ull;
  if (!m) return TPos{0, 0};
  const int zl = 63 - __builtin_clzll(m);
  return TPos{zl, !((X.m[2] >> zl) & 1) ? 2 : (!((X.m[1] >> zl) & 1) ? 1 : 0)};
}
__device__ __forceinline__ void tri_set_range(Tri& V, TPos a, TPos b) {   // positions a..b inclusive, lin(a) <= lin(b)
#pragma unroll
  for (int rr = 0; rr < 3; ++rr) {
    const int la = a.l + (rr < a.r ? 1 : 0), lb = b.l - (rr > b.r ? 1 : 0);
    if (la > lb) continue;
    const uint64_t upto = lb >= 63 ? ~0ull : ((2ull << lb) - 1);
    V.m[rr] |= upto & (~0ull << la);
  }
}
// SPEC §3 on residue-interleaved masks, read strand (same arithmetic as walk_masks); kq, kr = K / 3, K % 3
__device__ __forceinline__ int tri_walk(const Tri& H, const Tri& R, const Tri& LX, int L, int K, int kq, int kr, Tri& V) {
  V.m[0] = V.m[1] = V.m[2] = 0;
  const int last = L - K;
  TPos p = tri_ffs_stride3(H, TPos{0, 0});
  if (p.l < 0) return 0;
  int cov = 0;
  if (lin(p) >= L / 5 && lin(p) > 0) {
    const TPos z = tri_flz_upto(LX, p);
    const int t = lin(p) - lin(z);
    if (t > 0) { cov += t; tri_set_range(V, z, tpos_sub(p, 0, 1)); }
  }
  for (;;) {
    const TPos e = tri_ffz_from(R, p);
    cov += K + (lin(e) - lin(p));
    tri_set_range(V, p, e);
    const TPos b = tpos_add(e, kq, kr);
    if (lin(b) > last) break;
    p = tri_ffs_stride3(H, b);
    if (p.l < 0) break;
  }
  return cov;
}
// reverse-complement strand, walked downwards over read positions (same arithmetic as walk_masks_down)
__device__ __forceinline__ int tri_walk_down(const Tri& H, const Tri& R, const Tri& LX, int L, int K, int kq, int kr, TPos plast, Tri& V) {
  V.m[0] = V.m[1] = V.m[2] = 0;
  const int last = L - K;
  TPos ip = tri_fls_stride3_down(H, plast);
  if (ip.l < 0) return 0;
  int cov = 0;
  const int pj = last - lin(ip);
  if (pj >= L / 5 && pj > 0) {
    const TPos z = tri_ffz_from(LX, ip);
    const int t = lin(z) - lin(ip);
    if (t > 0) { cov += t; tri_set_range(V, tpos_add(ip, 0, 1), z); }
  }
  for (;;) {
    const TPos ei = tri_flz_upto(R, ip);
    cov += K + (lin(ip) - lin(ei));
    tri_set_range(V, ei, ip);
    const TPos tgt = tpos_sub(ei, kq, kr);          // read position of the next rc seed scan: ei - K
    if (tgt.l < 0) break;
    ip = tri_fls_stride3_down(H, tgt);
    if (ip.l < 0) break;
  }
  return cov;
}

// The part of a packed record one lane of the three-per-lane front end needs: 32 bases and their N bits starting at base
// 3*lane, plus the base before. Loaded one read AHEAD (the kernel prefetches the next record while it works on the
// current one), which takes the HBM-cold record fetch out of the per-read dependency chain.
struct TriRec { uint64_t W, NW; uint32_t prevb0; bool prevok0; };

__device__ __forceinline__ TriRec tri_load_record(const uint64_t* __restrict__ rp, int nwb, int L, int K, int lane) {
  TriRec t; t.W = 0; t.NW = 0; t.prevb0 = 0; t.prevok0 = false;
  const int P = L - K + 1, nwn = (nwb + 1) >> 1, i0 = 3 * lane;
  const uint64_t* __restrict__ np = rp + nwb;
  if (i0 < P) {
    const int wi = i0 >> 5, sh = (i0 & 31) * 2;
    const uint64_t lo = rp[wi], hi = rp[wi + 1];           // wi + 1 <= nwb: still inside this read's record
    t.W = sh ? ((lo >> sh) | (hi << (64 - sh))) : lo;       // bases i0 .. i0+31
    const int nwi = i0 >> 6, nsh = i0 & 63;
    const uint64_t nlo = np[nwi], nhi = (nwi + 1 < nwn) ? np[nwi + 1] : 0ull;
    t.NW = nsh ? ((nlo >> nsh) | (nhi << (64 - nsh))) : nlo;  // N bits of bases i0 ..
    if (i0 >= 1) { const int j = i0 - 1; t.prevb0 = (uint32_t)(rp[j >> 5] >> ((j & 31) * 2)) & 3u; t.prevok0 = !((np[j >> 6] >> (j & 63)) & 1); }
  }
  return t;
}

// Front end for 129..192 positions, k <= 29, in two halves so that the kernel can issue the NEXT read's record loads
// between them: tri_issue prepares the three k-mers of the lane and issues their bucket loads; tri_finish consumes the
// buckets, builds the masks, walks both strands and fills Strand<3> with w[r] / col[r] indexed by RESIDUE r.
// (vmcnt retires in issue order: record loads issued BEFORE the bucket loads would make the bucket wait include the
// HBM-cold record fetch; issued after, they overlap the mask/walk work.)
template <bool CT> struct TriProbe { uint64_t canon[3]; uint32_t meta[3]; uint4 ka[3], kb[3]; };   // (the bucket index is recomputed on the rare overflow path: 3 VGPRs less)
template <> struct TriProbe<true> { uint32_t want[3]; uint32_t meta[3]; uint4 ka[3], kb[3]; };      // compact table: ka = home bucket, kb = alternate bucket

template <bool CT>
__device__ __forceinline__ void tri_issue(const DevIndex& ix, const TriRec& rec, int L, int lane, TriProbe<CT>& q) {
  const int K = ix.K, P = L - K + 1;
  const int i0 = 3 * lane;
  const uint64_t W = rec.W, NW = rec.NW; const uint32_t prevb0 = rec.prevb0; const bool prevok0 = rec.prevok0;
  const uint64_t kmask = (1ull << (2 * K)) - 1, nkmask = (1ull << K) - 1;
  const uint64_t RCW = rc_kmer(W & ((1ull << (2 * (K + 2))) - 1), K + 2);   // one reverse complement for the three k-mers
  uint32_t bk[3]; [[maybe_unused]] uint32_t bk2[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const int i = i0 + r;
    const bool validk = (i < P) && (((NW >> r) & nkmask) == 0);
    const uint64_t x = (W >> (2 * r)) & kmask;
    const bool next_ok = (i + K < L) && !((NW >> (K + r)) & 1);
    const uint32_t nextb = (uint32_t)(W >> (2 * (K + r))) & 3u;
    const uint32_t prevb = r ? ((uint32_t)(W >> (2 * (r - 1))) & 3u) : prevb0;
    const bool prev_ok = r ? !((NW >> (r - 1)) & 1) : prevok0;
    const uint64_t y = (RCW >> (2 * (2 - r))) & kmask;
    const uint64_t c = x < y ? x : y;
    if constexpr (CT) {
      const CProbe pr = compact_probe(c, ix.chash);
      q.want[r] = pr.want; bk[r] = validk ? pr.b1 : 0u; bk2[r] = validk ? pr.b2 : 0u;
    } else {
      q.canon[r] = c;
#if defined(NB_ABLATE) && NB_ABLATE >= 3
      bk[r] = validk ? (kmer_bucket(c, ix.hash) & 1023u) : 0u;   // timing-only build: every lookup lands in 32 KB of the table
#else
      bk[r] = validk ? kmer_bucket(c, ix.hash) : 0u;
#endif
    }
    q.meta[r] = (validk ? 1u : 0u) | ((x != c) ? 2u : 0u) | (next_ok ? 4u : 0u) | (nextb << 3) | (prev_ok ? 32u : 0u) | (prevb << 6);
  }
  if constexpr (CT) {
#pragma unroll
    for (int r = 0; r < 3; ++r) { q.ka[r] = load_cbucket(ix.ctable, bk[r]); q.kb[r] = load_cbucket(ix.ctable, bk2[r]); }
  } else {
    const Bucket* __restrict__ B = ix.buckets;
#pragma unroll
    for (int r = 0; r < 3; ++r) { const uint4* kp = reinterpret_cast<const uint4*>(&B[bk[r]]); q.ka[r] = kp[0]; q.kb[r] = kp[1]; }
  }
}

// `lds` (or nullptr): this wave's six 64-lane rows of LDS. When given, the colour descriptors of the strand that has hits are
// fetched straight into them (global_load_lds: no VGPR destination) as soon as the hit masks exist, so that their memory
// round trip runs behind the scalar walk instead of after it. Returns which strand that was: 0 none, 1 read strand, 2 reverse.
template <bool CT>
__device__ __forceinline__ int tri_finish(const DevIndex& ix, const TriProbe<CT>& q, int L, int lane, Strand<3>& F, Strand<3>& R, uint4 (*lds)[64] = nullptr) {
  const int K = ix.K;
  const uint32_t* meta = q.meta; const uint4* ka = q.ka; const uint4* kb = q.kb;
  uint32_t payF[3], payR[3];
  if constexpr (CT) {
#pragma unroll
    for (int r = 0; r < 3; ++r) compact_match(ka[r], kb[r], q.want[r], (meta[r] >> 1) & 1u, meta[r] & 1u, payF[r], payR[r]);
  } else {
  const Bucket* __restrict__ B = ix.buckets;
  const uint64_t* canon = q.canon;
  bool ovf[3]; bool any_ovf = false;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const uint64_t k0 = ((uint64_t)ka[r].y << 32) | ka[r].x, k1 = ((uint64_t)kb[r].y << 32) | kb[r].x;
    const bool valid = meta[r] & 1u, o = meta[r] & 2u;
    const bool m0 = (k0 & KEY_MASK) == canon[r], m1 = k1 == canon[r];
    const bool found = valid && (m0 || m1);
    ovf[r] = valid && !found && (k0 >> 63);
    any_ovf |= ovf[r];
    const uint32_t p0 = m0 ? ka[r].z : kb[r].z, p1 = m0 ? ka[r].w : kb[r].w;
    payF[r] = found ? (o ? p1 : p0) : PAY_ABSENT;
    payR[r] = found ? (o ? p0 : p1) : PAY_ABSENT;
  }
  if (__ballot(any_ovf)) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      uint32_t b = kmer_bucket(canon[r], ix.hash); bool pending = ovf[r];   // (recomputed: only here is it needed after the loads)
      while (pending) {
        b = (b + 1 == ix.hash.n_buckets) ? 0u : b + 1;
        uint32_t p0, p1; bool more;
        if (probe_bucket(B, b, canon[r], p0, p1, more)) { const bool o = meta[r] & 2u; payF[r] = o ? p1 : p0; payR[r] = o ? p0 : p1; pending = false; }
        else pending = more;
      }
    }
  }
  }
  Tri HF, HR;
#pragma unroll
  for (int r = 0; r < 3; ++r) { F.col[r] = payF[r] >> 10; R.col[r] = payR[r] >> 10; HF.m[r] = __ballot(payF[r] != PAY_ABSENT); }
  const bool anyF = (HF.m[0] | HF.m[1] | HF.m[2]) != 0;
  int pre = 0;
  auto fetch_descs = [&](const Tri& H, const uint32_t (&col)[3]) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const uint32_t colour = lane_bit(H.m[r]) ? col[r] : 0u;
      const char* g = reinterpret_cast<const char*>(&ix.desc[colour]);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)&lds[2 * r][0], 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + 16), (__attribute__((address_space(3))) void*)&lds[2 * r + 1][0], 16, 0, 0);
    }
  };
  if (lds && anyF) { pre = 1; fetch_descs(HF, F.col); }
  // The read strand is balloted and walked before the extension masks of the reverse strand exist: eighteen 64-bit masks at
  // once do not fit the scalar register file next to the kernel arguments, and every spilled SGPR costs a v_writelane /
  // v_readlane pair. A read maps to one strand at most: the other one stops at its (empty) hit masks.
  const int kq = K / 3, kr = K - 3 * kq, last = L - K;
  {
    Tri RF, LF, VF;
    if (anyF) {
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const bool next_ok = meta[r] & 4u, prev_ok = meta[r] & 32u;
        const uint32_t nextb = (meta[r] >> 3) & 3u, prevb = (meta[r] >> 6) & 3u;
        const bool hF = payF[r] != PAY_ABSENT;
        RF.m[r] = __ballot(hF && next_ok && ((payF[r] >> nextb) & 1u));
        LF.m[r] = __ballot(hF && prev_ok && ((payF[r] >> (4 + prevb)) & 1u));
      }
    } else { RF.m[0] = RF.m[1] = RF.m[2] = LF.m[0] = LF.m[1] = LF.m[2] = 0; }
    F.cov = anyF ? tri_walk(HF, RF, LF, L, K, kq, kr, VF) : 0;
#pragma unroll
    for (int r = 0; r < 3; ++r) F.V.w[r] = F.cov ? VF.m[r] : 0ull;
  }
  {
    Tri RR, LR, VR;
#pragma unroll
    for (int r = 0; r < 3; ++r) HR.m[r] = __ballot(payR[r] != PAY_ABSENT);      // (after the read strand's walk: fewer masks live at once)
    const bool anyR = (HR.m[0] | HR.m[1] | HR.m[2]) != 0;
    if (lds && !pre && anyR) { pre = 2; fetch_descs(HR, R.col); }
    if (anyR) {
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const bool next_ok = meta[r] & 4u, prev_ok = meta[r] & 32u;
        const uint32_t nextb = (meta[r] >> 3) & 3u, prevb = (meta[r] >> 6) & 3u;
        const bool hR = payR[r] != PAY_ABSENT;
        RR.m[r] = __ballot(hR && prev_ok && ((payR[r] >> (3u - prevb)) & 1u));
        LR.m[r] = __ballot(hR && next_ok && ((payR[r] >> (4 + (3u - nextb))) & 1u));
      }
    } else { RR.m[0] = RR.m[1] = RR.m[2] = LR.m[0] = LR.m[1] = LR.m[2] = 0; }
    R.cov = anyR ? tri_walk_down(HR, RR, LR, L, K, kq, kr, TPos{last / 3, last - 3 * (last / 3)}, VR) : 0;
#pragma unroll
    for (int r = 0; r < 3; ++r) R.V.w[r] = R.cov ? VR.m[r] : 0ull;
  }
  return pre;
}

// Step 5: intern the set held in acc (lane-distributed) and return its dense class id, or NB_NO_CLASS on overflow.
template <int TW>
__device__ __forceinline__ uint32_t intern_class(const DevClasses& ct, int W, int lane, const uint64_t (&acc)[TW]) {
  uint64_t hx = 0;      // set hash: xor of one mixed word per (word index, value); every lane owns distinct word indices
#pragma unroll
  for (int t = 0; t < TW; ++t) if (acc[t]) hx ^= mix64(acc[t] ^ (0x9E3779B97F4A7C15ull * (uint64_t)(64 * t + lane + 1)));
  uint64_t key = wave_xor_u64(hx);
  if (key == 0) key = 1;
  const uint64_t smask = (1ull << ct.cap_log2) - 1;
  uint64_t h = (key * HASH_MULT) >> (64 - ct.cap_log2);
  for (uint64_t probes = 0; probes <= smask; ++probes, h = (h + 1) & smask) {
    uint64_t cur = 0; uint32_t idp = 0;
    if (lane == 0) {   // key and id word fetched together: one round trip on the common (already interned) path
      cur = __hip_atomic_load(&ct.keys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      idp = __hip_atomic_load(&ct.ids[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    cur = bcast_u64(cur); idp = __builtin_amdgcn_readfirstlane(idp);
    if (cur == 0) {
      if (lane == 0) { unsigned long long expected = 0ull;
        __hip_atomic_compare_exchange_strong(&ct.keys[h], &expected, (unsigned long long)key, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        cur = expected; }
      cur = bcast_u64(cur);
      if (cur == 0) {   // this wave owns the slot: allocate an id, publish the row, then the id
        uint32_t id = 0;
        if (lane == 0) id = __hip_atomic_fetch_add(ct.n_classes, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        id = __builtin_amdgcn_readfirstlane(id);
        if (id >= ct.max_classes) {
          if (lane == 0) { atomicOr(ct.err, 1u); __hip_atomic_store(&ct.ids[h], 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
          return NB_NO_CLASS;
        }
        uint32_t wlo = 0xFFFFu, whi = 0u;
#pragma unroll
        for (int t = 0; t < TW; ++t) { const int w = lane + 64 * t;
          if (w < W) __hip_atomic_store(&ct.rows[(size_t)id * W + w], (unsigned long long)acc[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (acc[t]) { wlo = (uint32_t)w < wlo ? (uint32_t)w : wlo; whi = (uint32_t)w > whi ? (uint32_t)w : whi; } }
        wlo = wave_min_u32(wlo); whi = wave_max_u32(whi);
        if (lane == 0) __hip_atomic_store(&ct.spans[id], wlo | (whi << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(&ct.ids[h], id + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return id;
      }
    }
    if (cur == key) {
      for (int spin = 0; idp == 0 && spin < (1 << 22); ++spin) {   // the owner is still publishing: poll
        __builtin_amdgcn_s_sleep(8);
        if (lane == 0) idp = __hip_atomic_load(&ct.ids[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        idp = __builtin_amdgcn_readfirstlane(idp);
      }
      if (idp == 0) { if (lane == 0) atomicOr(ct.err, 2u); return NB_NO_CLASS; }
      if (idp == 0xFFFFFFFFu) return NB_NO_CLASS;
      const uint32_t id = idp - 1u;
      bool diff = false;
#pragma unroll
      for (int t = 0; t < TW; ++t) { const int w = lane + 64 * t;
        if (w < W) diff |= (__hip_atomic_load(&ct.rows[(size_t)id * W + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != acc[t]); }
      if (__ballot(diff) == 0) return id;
      // genuine 64-bit hash collision between two different sets: keep probing
    }
  }
  if (lane == 0) atomicOr(ct.err, 1u);
  return NB_NO_CLASS;
}


// ------------------------------------------------------------------------------------------------------------
// single-end kernel: one read per wave, grid-stride over the batch.
#ifndef NB_SE_WAVES_PER_EU
#define NB_SE_WAVES_PER_EU 1, 6   // measured on MI355X (10M x 150bp, 1k seqs): 6 waves/SIMD 23.4 ms, 7 -> 28.1, 8 -> 25.5:
#endif                             // the kernel is bound by L2-miss request rate, more waves only thrash L2
template <int NR, int TW, bool TRI, bool CT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NB_SE_WAVES_PER_EU))) void nb_align_se_kernel(DevIndex ix, DevClasses ctv, const DevClasses* __restrict__ ctp, BatchArgs a, const uint64_t* __restrict__ reads_ro) {
  __shared__ unsigned int s_reason[NB_NUM_REASONS];
  constexpr bool LDS_DESC = TW == 1 && (TRI || NR == 2);   // (wide class rows go through `packed` anyway: a loss on c5; one round of positions: neutral)
  __shared__ uint4 s_desc[LDS_DESC ? 4 : 1][LDS_DESC ? 2 * NR : 1][LDS_DESC ? 64 : 1];   // per wave: 3 residues x 2 halves of a 32-byte descriptor per lane
  __shared__ unsigned int s_ctag[COUNT_CACHE], s_ccnt[COUNT_CACHE];                        // hot class counts (nb_device.h)
  if (threadIdx.x < NB_NUM_REASONS) s_reason[threadIdx.x] = 0;
  count_cache_init(s_ctag, s_ccnt);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long long nwaves = (long long)gridDim.x * (blockDim.x >> 6);
  const long long r0 = (long long)blockIdx.x * (blockDim.x >> 6) + wave;
  const long long n_items = a.n_dev ? (long long)*a.n_dev : a.n;   // (redo list of the lane-per-read path: the count lives on the device)
  TriRec nxt{}; int Lnxt = 0;
  if constexpr (TRI) {
    if (r0 < n_items) {
      Lnxt = a.lens ? (int)a.lens[r0] : a.max_len; if (Lnxt > a.max_len) Lnxt = a.max_len;
      if (Lnxt >= ix.K) nxt = tri_load_record(reads_ro + (size_t)r0 * a.stride, a.nwb, Lnxt, ix.K, lane);
    }
  }
  for (long long r = r0; r < n_items; r += nwaves) {
    int L;
    TriRec rec{};
    if constexpr (TRI) {
      L = Lnxt; rec = nxt;
    } else {
      L = a.lens ? (int)a.lens[r] : a.max_len;
      if (L > a.max_len) L = a.max_len;   // never index past the packed record
    }
    const int pmin = a.lens ? (int)a.pct_min[L] : a.pct_fixed;
    int reason = NB_TOO_SHORT, score = 0, orient = 0, mates = 0; uint32_t cls = NB_NO_CLASS; int covF = 0, covR = 0;
    auto prefetch_next = [&]() {
      const long long rn = r + nwaves;
      if (rn < n_items) {
        Lnxt = a.lens ? (int)a.lens[rn] : a.max_len; if (Lnxt > a.max_len) Lnxt = a.max_len;
        if (Lnxt >= ix.K) nxt = tri_load_record(reads_ro + (size_t)rn * a.stride, a.nwb, Lnxt, ix.K, lane);
      }
    };
    if (TRI && L < ix.K) prefetch_next();
    if (L >= ix.K) {
      Strand<NR> SF, SR;
      [[maybe_unused]] int pre = 0;                                          // strand whose descriptors wait in LDS (1 F, 2 R)
      if constexpr (TRI) {                                                  // 129..192 positions: three per lane
        TriProbe<CT> q;
        tri_issue<CT>(ix, rec, L, lane, q);
        prefetch_next();                              // this wave's next record, issued BEHIND the bucket loads
        if constexpr (LDS_DESC) pre = tri_finish<CT>(ix, q, L, lane, SF, SR, s_desc[wave]);
        else pre = tri_finish<CT>(ix, q, L, lane, SF, SR);
      } else {
        const uint64_t* __restrict__ rp = reads_ro + (size_t)r * a.stride;   // separate noalias kernel argument: lets the
        if constexpr (LDS_DESC) pre = map_both_strands_fast<NR, CT>(ix, rp, a.nwb, L, lane, SF, SR, s_desc[wave]);   // compiler fetch the record
        else map_both_strands_fast<NR, CT>(ix, rp, a.nwb, L, lane, SF, SR);                                          // with scalar loads
      }
      covF = SF.cov; covR = SR.cov;
      // candidates in preference order: larger coverage first, tie -> F   (SPEC §4 "Chosen orientation")
      const bool first_is_R = covR > covF;
      int reasons[2]; uint64_t acc[TW]; bool called = false;
#pragma unroll
      for (int c = 0; c < 2 && !called; ++c) {
        const bool isR = (c == 0) ? first_is_R : !first_is_R;
        const int cov = isR ? covR : covF;
        int rs = cheap_reason_p(cov, pmin, a);
        if (rs == 0) {
#if defined(NB_ABLATE) && NB_ABLATE >= 2
          acc[0] = (uint64_t)(isR ? SR.col[0] : SF.col[0]); const uint32_t n = 1;   // timing-only build: no intersection
#else
          uint32_t n;
          bool from_lds = false;
          if constexpr (LDS_DESC) from_lds = pre == (isR ? 2 : 1);
          if (from_lds) {                               // descriptors fetched into LDS behind the walk (tri_finish)
            DescRegs d[NR];
            if constexpr (LDS_DESC) {
              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
              for (int rr = 0; rr < NR; ++rr) {
                const uint4 da = s_desc[wave][2 * rr][lane], db = s_desc[wave][2 * rr + 1][lane];
                d[rr].w0 = da.x; d[rr].nw = da.y; d[rr].off = da.z; d[rr].m0 = ((uint64_t)db.y << 32) | db.x; d[rr].m1 = ((uint64_t)db.w << 32) | db.z;
              }
            }
            n = isR ? intersect_descs<NR, TW>(ix, SR, d, lane, acc) : intersect_descs<NR, TW>(ix, SF, d, lane, acc);
          } else n = isR ? intersect_visited<NR, TW>(ix, SR, lane, acc) : intersect_visited<NR, TW>(ix, SF, lane, acc);
#endif
          if (n == 0) rs = NB_EMPTY_EQCLASS;
          else if (a.discard_multiple && n != 1) rs = NB_MULTIPLE_MATCH;
          else if ((isR && a.strand_filter == NB_STRAND_FIVEPRIME) || (!isR && a.strand_filter == NB_STRAND_THREEPRIME)) rs = NB_STRAND;
          else { called = true; orient = isR ? 1 : 0; score = cov; mates = 1; }
        }
        reasons[c] = rs;
      }
      if (called) {
        if (score < a.score_filter) reason = NB_SCORE_FILTER;
        else {
          reason = NB_CALLED;
#if defined(NB_ABLATE) && NB_ABLATE >= 1
          cls = (uint32_t)acc[0] & 1u;   // timing-only build: no interning, no counting
#else
          // Three-per-lane variant: the class-table descriptor is fetched here, once per called read (one scalar load from the
          // constant cache), instead of living in 16 SGPRs for the whole kernel: the masks of the walk need that room, and a
          // spilled SGPR costs a v_writelane / v_readlane pair on the vector pipe. The other variants have the room and keep
          // it in registers (the per-read scalar load costs them 14 %: 134 -> 153 ms on the 91-bp workload).
          DevClasses ct = ctv;
          if constexpr (TRI) { asm volatile("" ::: "memory"); ct = *ctp; }
          cls = intern_class<TW>(ct, ix.W, lane, acc);
          if (cls != NB_NO_CLASS && lane == 0) count_cache_add(s_ctag, s_ccnt, ct.counts, cls, 1u);
#endif
        }
      } else reason = reasons[0] > reasons[1] ? reasons[0] : reasons[1];
    }
    if (lane == 0) {
      atomicAdd(&s_reason[reason], 1u);
      if (a.out_class) a.out_class[r] = (reason == NB_CALLED) ? cls : NB_NO_CLASS;
      if (a.out_info) a.out_info[r] = (uint32_t)(score & 0xFFFF) | ((uint32_t)reason << 16) | ((uint32_t)orient << 24) | ((uint32_t)mates << 25);
      if (a.out_cov) a.out_cov[r] = (uint64_t)(covF & 0xFFFF) | ((uint64_t)(covR & 0xFFFF) << 16);
    }
  }
  __syncthreads();
  if (threadIdx.x < NB_NUM_REASONS && s_reason[threadIdx.x]) atomicAdd(&ctp->reasons[threadIdx.x], (unsigned long long)s_reason[threadIdx.x]);
  count_cache_flush(s_ctag, s_ccnt, ctp->counts);
}

// ------------------------------------------------------------------------------------------------------------
// general kernel: paired templates and/or num_mismatches > 0. Eager evaluation of every alignment of the template
// (SPEC §4), one template per wave. Each sequence is first materialised in LDS in the orientation being mapped
// (forward copy or reverse complement), so the mapper below only ever walks "forwards"; a tolerated mismatch
// (SPEC §3: inside a unitig, budget M) is handled by PATCHING that LDS copy with the reference base and re-running the
// data-parallel lookups — seeding always uses the unpatched hit mask, exactly as the sequential walk seeds from the
// read itself.
constexpr int GEN_NWB = (NB_MAX_POSITIONS + NB_MAX_K - 1 + 31) / 32;   // 11 words of bases
constexpr int GEN_NWN = (NB_MAX_POSITIONS + NB_MAX_K - 1 + 63) / 64;   // 6 words of N mask
constexpr int GEN_WORDS = GEN_NWB + GEN_NWN;

// copy (rev = false) or reverse-complement (rev = true) a packed sequence of length L into LDS, same layout (nwb words
// of bases, then N-mask words)
__device__ __forceinline__ void stage_sequence(uint64_t* dst, const uint64_t* __restrict__ src, int nwb, int L, bool rev, int lane) {
  const int nwn = (nwb + 1) >> 1;
  for (int w = lane; w < nwb + nwn; w += 64) {
    uint64_t v = 0;
    if (!rev) v = src[w];
    else if (w < nwb) {
      for (int t = 0; t < 32; ++t) { const int jp = 32 * w + t; if (jp < L) { const int j = L - 1 - jp;
          const bool isn = (src[nwb + (j >> 6)] >> (j & 63)) & 1;
          const uint64_t c = isn ? 0ull : (3ull - ((src[j >> 5] >> ((j & 31) * 2)) & 3ull));
          v |= c << (2 * t); } }
    } else {
      const int wn = w - nwb;
      for (int t = 0; t < 64; ++t) { const int jp = 64 * wn + t; if (jp < L) { const int j = L - 1 - jp;
          v |= ((src[nwb + (j >> 6)] >> (j & 63)) & 1ull) << t; } }
    }
    dst[w] = v;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

template <int NR> __device__ __forceinline__ bool test_bit(const Mask<NR>& m, int pos) {
  uint64_t w = 0;
#pragma unroll
  for (int wi = 0; wi < NR; ++wi) if (wi == (pos >> 6)) w = m.w[wi];
  return (w >> (pos & 63)) & 1;
}

struct PatchReq { int base; int from_pos; int left; };   // base < 0: none

// SPEC §3 on masks with tolerance decision points. Hseed: hits of the UNPATCHED sequence (seeding). R/LX: extendable
// with the (patched) sequence; RI/LI: k-mer present and r_internal / l_internal.
template <int NR>
__device__ __forceinline__ int walk_general(const Mask<NR>& Hseed, const Mask<NR>& R, const Mask<NR>& LX, const Mask<NR>& RI,
                                            const Mask<NR>& LI, int L, int K, bool can_patch, Mask<NR>& V, PatchReq& pr) {
#pragma unroll
  for (int wi = 0; wi < NR; ++wi) V.w[wi] = 0;
  pr.base = -1; pr.from_pos = 0; pr.left = 0;
  const int last = L - K;
  int p = ffs_stride3<NR>(Hseed, 0);
  if (p < 0) return 0;
  int cov = 0;
  if (p >= L / 5 && p > 0) {
    const int z = flz_upto<NR>(LX, p);          // lowest k-mer position reached going left
    const int t = p - z;
    if (t > 0) { cov += t; set_range<NR>(V, z, p - 1); }
    if (z >= 1 && can_patch && test_bit<NR>(LI, z)) { pr.base = z - 1; pr.from_pos = z; pr.left = 1; return cov; }
  }
  for (;;) {
    const int e = ffz_from<NR>(R, p);
    cov += K + (e - p);
    set_range<NR>(V, p, e);
    const int b = e + K;
    if (b < L && can_patch && test_bit<NR>(RI, e)) { pr.base = b; pr.from_pos = e; pr.left = 0; return cov; }
    if (b > last) break;
    p = ffs_stride3<NR>(Hseed, b);
    if (p < 0) break;
  }
  return cov;
}

// Map the sequence staged in `seq` (LDS). Returns coverage; S.V / S.col describe the visited positions.
template <int NR, bool CT>
__device__ __forceinline__ void map_staged(const DevIndex& ix, uint64_t* seq, int nwb, int L, int M, int lane, Strand<NR>& S) {
  const int K = ix.K, P = L - K + 1;
  Mask<NR> Hseed; int mm = 0;
  for (int pass = 0;; ++pass) {
    Mask<NR> H, R, LX, RI, LI; uint32_t pay[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int i = lane + 64 * r;
      const PosFacts f = position_facts<CT>(ix, seq, nwb, L, i, i < P);
      pay[r] = f.payF; S.col[r] = f.payF >> 10;
      H.w[r] = __ballot(f.hF); R.w[r] = __ballot(f.rF); LX.w[r] = __ballot(f.lF);
      RI.w[r] = __ballot(f.hF && ((f.payF >> 8) & 1u)); LI.w[r] = __ballot(f.hF && ((f.payF >> 9) & 1u));
    }
    if (pass == 0) Hseed = H;
    PatchReq pr;
    S.cov = any_set<NR>(Hseed) ? walk_general<NR>(Hseed, R, LX, RI, LI, L, K, mm < M, S.V, pr) : 0;
    if (!any_set<NR>(Hseed)) { pr.base = -1;
#pragma unroll
      for (int r = 0; r < NR; ++r) S.V.w[r] = 0; }
    if (pr.base < 0) return;
    // fetch the payload of k-mer position from_pos, patch the staged base with its unique extension
    uint32_t pv = 0;
#pragma unroll
    for (int r = 0; r < NR; ++r) if (r == (pr.from_pos >> 6)) pv = pay[r];
    const uint32_t src_pay = (uint32_t)__shfl((int)pv, pr.from_pos & 63, 64);
    const uint32_t ext = pr.left ? ((src_pay >> 4) & 15u) : (src_pay & 15u);
    const uint64_t u = (uint64_t)__builtin_ctz(ext | 16u);
    if (lane == 0) {
      const int q = pr.base;
      seq[q >> 5] = (seq[q >> 5] & ~(3ull << ((q & 31) * 2))) | (u << ((q & 31) * 2));
      seq[nwb + (q >> 6)] &= ~(1ull << (q & 63));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    ++mm;
  }
}

// SPEC §4 checks 1-4 for one mapped strand; on success acc holds eq. Returns the reason (0 = valid).
template <int NR, int TW>
__device__ __forceinline__ int evaluate_alignment(const DevIndex& ix, const Strand<NR>& S, int L, const BatchArgs& a, int lane, uint64_t (&acc)[TW]) {
  int rs = cheap_reason(S.cov, L, a);
  if (rs) return rs;
  const uint32_t n = intersect_visited<NR, TW>(ix, S, lane, acc);
  if (n == 0) return NB_EMPTY_EQCLASS;
  if (a.discard_multiple && n != 1) return NB_MULTIPLE_MATCH;
  return 0;
}

template <int TW> struct OrientRes { bool valid; int reason, score, mates; uint64_t set[TW]; };

// SPEC §4 "Per orientation": combine the two mates' alignments (r1: reason1/cov1/eq1, r2 likewise).
template <int TW>
__device__ __forceinline__ void combine_mates(bool paired, int reason1, int cov1, const uint64_t (&eq1)[TW], int reason2, int cov2,
                                              const uint64_t (&eq2)[TW], const BatchArgs& a, OrientRes<TW>& o) {
  o.valid = false; o.score = 0; o.mates = 0; o.reason = reason1;
#pragma unroll
  for (int t = 0; t < TW; ++t) o.set[t] = 0;
  const bool v1 = reason1 == 0, v2 = paired && reason2 == 0;
  if (!paired) {
    if (v1) { o.valid = true; o.reason = 0; o.score = cov1; o.mates = 1;
#pragma unroll
      for (int t = 0; t < TW; ++t) o.set[t] = eq1[t]; }
    return;
  }
  if (v1 && v2) {
    uint32_t ni = 0;
#pragma unroll
    for (int t = 0; t < TW; ++t) ni += (uint32_t)__popcll(eq1[t] & eq2[t]);
    ni = wave_sum_u32(ni);
    const bool use_inter = (a.intersect_level == 2) || (a.intersect_level == 1 && ni != 0);
    if (a.intersect_level == 2 && ni == 0) { o.reason = NB_EMPTY_EQCLASS; return; }
#pragma unroll
    for (int t = 0; t < TW; ++t) o.set[t] = use_inter ? (eq1[t] & eq2[t]) : (eq1[t] | eq2[t]);
    o.valid = true; o.reason = 0; o.score = cov1 + cov2; o.mates = 2; return;
  }
  if (v1 || v2) {
    if (a.require_valid_pair) { o.reason = NB_INVALID_PAIR; return; }
    o.valid = true; o.reason = 0; o.score = v1 ? cov1 : cov2; o.mates = 1;
#pragma unroll
    for (int t = 0; t < TW; ++t) o.set[t] = v1 ? eq1[t] : eq2[t];
    return;
  }
  o.reason = reason1 > reason2 ? reason1 : reason2;
}

// MODE 0: LDS-staged mapper (any num_mismatches, single-end or paired). MODE 1 / 2: exact mode (num_mismatches == 0) with
// the hot kernel's front ends applied to each mate (1 = one position per lane, 2 = three positions per lane): no staging,
// both strands of a mate from one pass of lookups — the path of paired FASTQ / paired BAM input.
template <int NR, int TW, int MODE, bool CT>
__global__ __launch_bounds__(256) void nb_align_general_kernel(DevIndex ix, DevClasses ct, BatchArgs a) {
  __shared__ unsigned int s_reason[NB_NUM_REASONS];
  __shared__ uint64_t s_seq[4][GEN_WORDS];
  __shared__ unsigned int s_ctag[COUNT_CACHE], s_ccnt[COUNT_CACHE];                        // hot class counts (nb_device.h)
  if (threadIdx.x < NB_NUM_REASONS) s_reason[threadIdx.x] = 0;
  count_cache_init(s_ctag, s_ccnt);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  uint64_t* seq = s_seq[MODE == 0 ? wave : 0];
  const bool paired = a.mates != nullptr;
  const long long nwaves = (long long)gridDim.x * (blockDim.x >> 6);
  const long long n_items = a.n_dev ? (long long)*a.n_dev : a.n;   // (redo list of the paired anchor path: the count lives on the device)
  for (long long r = (long long)blockIdx.x * (blockDim.x >> 6) + wave; r < n_items; r += nwaves) {
    int L1 = a.lens ? (int)a.lens[r] : a.max_len; if (L1 > a.max_len) L1 = a.max_len;
    int L2 = 0;
    if (paired) { L2 = a.mate_lens ? (int)a.mate_lens[r] : a.mate_max_len; if (L2 > a.mate_max_len) L2 = a.mate_max_len; }
    // four alignments: 0 = map(r1), 1 = map(rc(r1)), 2 = map(r2), 3 = map(rc(r2))
    int rs[4], cov[4]; uint64_t eq[4][TW];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      rs[j] = NB_TOO_SHORT; cov[j] = 0;
#pragma unroll
      for (int t = 0; t < TW; ++t) eq[j][t] = 0;
    }
    if constexpr (MODE == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool second = j >= 2;
        if (second && !paired) continue;
        const int L = second ? L2 : L1;
        if (L < ix.K) continue;
        const uint64_t* src = second ? a.mates + (size_t)r * a.mstride : a.reads + (size_t)r * a.stride;
        const int nwb = second ? a.mnwb : a.nwb;
        stage_sequence(seq, src, nwb, L, (j & 1) != 0, lane);
        Strand<NR> S;
        map_staged<NR, CT>(ix, seq, nwb, L, a.num_mismatches, lane, S);
        cov[j] = S.cov;
        rs[j] = evaluate_alignment<NR, TW>(ix, S, L, a, lane, eq[j]);
      }
    } else {
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        if (m && !paired) continue;
        const int L = m ? L2 : L1;
        if (L < ix.K) continue;
        const uint64_t* src = m ? a.mates + (size_t)r * a.mstride : a.reads + (size_t)r * a.stride;
        const int nwb = m ? a.mnwb : a.nwb;
        Strand<NR> SF, SR;
        if constexpr (MODE == 2) {
          const TriRec rec = tri_load_record(src, nwb, L, ix.K, lane);
          TriProbe<CT> q;
          tri_issue<CT>(ix, rec, L, lane, q);
          tri_finish<CT>(ix, q, L, lane, SF, SR);
        } else map_both_strands_fast<NR, CT>(ix, src, nwb, L, lane, SF, SR);
        cov[2 * m] = SF.cov; cov[2 * m + 1] = SR.cov;
        rs[2 * m] = evaluate_alignment<NR, TW>(ix, SF, L, a, lane, eq[2 * m]);
        rs[2 * m + 1] = evaluate_alignment<NR, TW>(ix, SR, L, a, lane, eq[2 * m + 1]);
      }
    }
    OrientRes<TW> F, R;
    combine_mates<TW>(paired, rs[0], cov[0], eq[0], rs[3], cov[3], eq[3], a, F);   // F = {map(r1), map(rc(r2))}
    combine_mates<TW>(paired, rs[1], cov[1], eq[1], rs[2], cov[2], eq[2], a, R);   // R = {map(rc(r1)), map(r2)}
    if (F.valid && a.strand_filter == NB_STRAND_THREEPRIME) { F.valid = false; F.reason = NB_STRAND; }
    if (R.valid && a.strand_filter == NB_STRAND_FIVEPRIME) { R.valid = false; R.reason = NB_STRAND; }
    int reason, score = 0, orient = 0, mates = 0; uint32_t cls = NB_NO_CLASS;
    const bool pickR = R.valid && (!F.valid || R.score > F.score);
    if (F.valid || R.valid) {
      score = pickR ? R.score : F.score; orient = pickR ? 1 : 0; mates = pickR ? R.mates : F.mates;
      if (score < a.score_filter) reason = NB_SCORE_FILTER;
      else {
        reason = NB_CALLED;
        uint64_t acc[TW];
#pragma unroll
        for (int t = 0; t < TW; ++t) acc[t] = pickR ? R.set[t] : F.set[t];
        cls = intern_class<TW>(ct, ix.W, lane, acc);
        if (cls != NB_NO_CLASS && lane == 0) count_cache_add(s_ctag, s_ccnt, ct.counts, cls, 1u);
      }
    } else reason = F.reason > R.reason ? F.reason : R.reason;
    if (lane == 0) {
      atomicAdd(&s_reason[reason], 1u);
      if (a.out_class) a.out_class[r] = (reason == NB_CALLED) ? cls : NB_NO_CLASS;
      if (a.out_info) a.out_info[r] = (uint32_t)(score & 0xFFFF) | ((uint32_t)reason << 16) | ((uint32_t)orient << 24) | ((uint32_t)mates << 25);
      if (a.out_cov) a.out_cov[r] = (uint64_t)(cov[0] & 0xFFFF) | ((uint64_t)(cov[1] & 0xFFFF) << 16) | ((uint64_t)(cov[2] & 0xFFFF) << 32) | ((uint64_t)(cov[3] & 0xFFFF) << 48);
    }
  }
  __syncthreads();
  if (threadIdx.x < NB_NUM_REASONS && s_reason[threadIdx.x]) atomicAdd(&ct.reasons[threadIdx.x], (unsigned long long)s_reason[threadIdx.x]);
  count_cache_flush(s_ctag, s_ccnt, ct.counts);
}

// ------------------------------------------------------------------------------------------------------------
// host side of the C-ABI
thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }

}  // namespace nb

using namespace nb;

static int ensure_pct_table(nb_ctx* c, double score_percent, int max_len, hipStream_t stream) {
  if (c->d_pct && c->pct_value == score_percent && c->pct_len >= max_len + 1) return NB_OK;
  (void)hipFree(c->d_pct); c->d_pct = nullptr;
  std::vector<uint16_t> t((size_t)max_len + 1, 0);
  for (int L = 1; L <= max_len; ++L) {   // smallest cov with (double)cov/(double)L >= score_percent   (SPEC §4.2)
    int cov = 0; while (cov <= 65534 && (double)cov / (double)L < score_percent) ++cov;
    t[(size_t)L] = (uint16_t)cov;
  }
  NB_HIP(hipMalloc((void**)&c->d_pct, t.size() * 2));
  NB_HIP(hipMemcpyAsync(c->d_pct, t.data(), t.size() * 2, hipMemcpyHostToDevice, stream));
  NB_HIP(hipStreamSynchronize(stream));
  c->pct_len = max_len + 1; c->pct_value = score_percent; c->pct_host = t; return NB_OK;
}

// Persistent-style grid: as many 256-thread blocks as are co-resident (occupancy query x CUs), every wave strides over
// the batch. Sizing the grid above residency would run the surplus blocks as a second, partly empty round.
template <class K>
static int resident_blocks(nb_ctx* c, K kernel, long long n_templates) {
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) != hipSuccess || per_cu < 1) per_cu = 4;
  // one block = one wave per SIMD. Beyond 6 waves per SIMD the extra waves only evict each other's table lines from L2
  // (measured on the 91-bp workload: 133 ms at 6, 153 ms at 7; profiles/r01_tuning_notes.md), so the grid stops there.
  int cap = 6;
  if (const char* e = std::getenv("NB_BLOCKS_PER_CU")) { const int v = std::atoi(e); if (v >= 1 && v <= 8) cap = v; }
  if (per_cu > cap) per_cu = cap;
  long long blocks = (long long)c->n_cu * per_cu;
  const long long need = (n_templates + 3) / 4;
  if (blocks > need) blocks = need;
  return (int)(blocks < 1 ? 1 : blocks);
}

template <int NR, int TW, bool TRI, bool CT>
static int launch_se(nb_ctx* c, const DevIndex& ix, const BatchArgs& a, hipStream_t s) {
  static thread_local int cached_blocks = 0; static thread_local int cached_cu = 0;
  if (!cached_blocks || cached_cu != c->n_cu || std::getenv("NB_BLOCKS_PER_CU")) { cached_blocks = resident_blocks(c, nb_align_se_kernel<NR, TW, TRI, CT>, 1ll << 40); cached_cu = c->n_cu; }
  long long blocks = cached_blocks; const long long need = (a.n + 3) / 4;
  if (blocks > need) blocks = need;
  hipLaunchKernelGGL((nb_align_se_kernel<NR, TW, TRI, CT>), dim3((unsigned)blocks), dim3(256), 0, s, ix, c->ct, (const DevClasses*)c->d_ct, a, a.reads);
  NB_HIP(hipGetLastError());
  { char buf[96]; std::snprintf(buf, sizeof buf, "nb_align_se_kernel<%d, %d, %s, %s>", NR, TW, TRI ? "true" : "false", CT ? "true" : "false");
    if (!c->last_kernels.empty()) c->last_kernels += ";"; c->last_kernels += buf; }
  return NB_OK;
}

// three-positions-per-lane front end: 129..192 positions and k <= 29 (window of k+3 bases in one 64-bit word)
static bool use_tri(int positions, int k) { return positions > 128 && positions <= 192 && k <= 29; }

template <int NR, int TW, int MODE, bool CT>
static int launch_general(nb_ctx* c, const DevIndex& ix, const BatchArgs& a, hipStream_t s) {
  const long long blocks = resident_blocks(c, nb_align_general_kernel<NR, TW, MODE, CT>, a.n);
  hipLaunchKernelGGL((nb_align_general_kernel<NR, TW, MODE, CT>), dim3((unsigned)blocks), dim3(256), 0, s, ix, c->ct, a);
  NB_HIP(hipGetLastError());
  { char buf[96]; std::snprintf(buf, sizeof buf, "nb_align_general_kernel<%d, %d, %d, %s>", NR, TW, MODE, CT ? "true" : "false"); c->last_kernels = buf; }
  return NB_OK;
}

template <int TW, bool CT>
static int dispatch_general(nb_ctx* c, int nr, int positions, bool exact, const DevIndex& ix, const BatchArgs& a, hipStream_t s) {
  if (exact && use_tri(positions, ix.K)) return launch_general<3, TW, 2, CT>(c, ix, a, s);
  if (exact) switch (nr) {
    case 1: return launch_general<1, TW, 1, CT>(c, ix, a, s);
    case 2: return launch_general<2, TW, 1, CT>(c, ix, a, s);
    case 3: return launch_general<3, TW, 1, CT>(c, ix, a, s);
    case 4: return launch_general<4, TW, 1, CT>(c, ix, a, s);
    default: return launch_general<5, TW, 1, CT>(c, ix, a, s);
  }
  switch (nr) {
    case 1: return launch_general<1, TW, 0, CT>(c, ix, a, s);
    case 2: return launch_general<2, TW, 0, CT>(c, ix, a, s);
    case 3: return launch_general<3, TW, 0, CT>(c, ix, a, s);
    case 4: return launch_general<4, TW, 0, CT>(c, ix, a, s);
    default: return launch_general<5, TW, 0, CT>(c, ix, a, s);
  }
}

template <int TW, bool CT>
static int dispatch_se(nb_ctx* c, int nr, int positions, const DevIndex& ix, const BatchArgs& a, hipStream_t s) {
  if (use_tri(positions, ix.K)) return launch_se<3, TW, true, CT>(c, ix, a, s);
  switch (nr) {
    case 1: return launch_se<1, TW, false, CT>(c, ix, a, s);
    case 2: return launch_se<2, TW, false, CT>(c, ix, a, s);
    case 3: return launch_se<3, TW, false, CT>(c, ix, a, s);
    case 4: return launch_se<4, TW, false, CT>(c, ix, a, s);
    default: return launch_se<5, TW, false, CT>(c, ix, a, s);
  }
}

extern "C" {

int32_t nb_version(void) { return NB_ABI_VERSION; }
const char* nb_last_error(void) { return g_err.c_str(); }

int32_t nb_ctx_create(int32_t device_id, int64_t max_classes, nb_ctx** out) {
  if (!out || max_classes < 1 || max_classes > (1ll << 28)) { set_error("nb_ctx_create: bad arguments"); return NB_ERR_INVALID; }
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0) { set_error("nb_ctx_create: no HIP device (this library has no CPU fallback)"); return NB_ERR_NOGPU; }
  if (device_id < 0 || device_id >= n) { set_error("nb_ctx_create: device_id out of range"); return NB_ERR_INVALID; }
  NB_HIP(hipSetDevice(device_id));
  nb_ctx* c = new nb_ctx(); c->device = device_id; c->max_classes = max_classes;
  hipDeviceProp_t prop; NB_HIP(hipGetDeviceProperties(&prop, device_id)); c->n_cu = prop.multiProcessorCount;
  c->ct.max_classes = (unsigned int)max_classes;
  unsigned int cl = 10; while ((1ull << cl) < 2ull * (uint64_t)max_classes) ++cl; c->ct.cap_log2 = cl;
  const size_t cap = (size_t)1 << cl;
  NB_HIP(hipMalloc((void**)&c->ct.keys, cap * 8)); NB_HIP(hipMalloc((void**)&c->ct.ids, cap * 4));
  NB_HIP(hipMalloc((void**)&c->ct.counts, (size_t)max_classes * 8)); NB_HIP(hipMalloc((void**)&c->ct.spans, (size_t)max_classes * 4));
  NB_HIP(hipMalloc((void**)&c->ct.n_classes, 4)); NB_HIP(hipMalloc((void**)&c->ct.err, 4));
  NB_HIP(hipMalloc((void**)&c->ct.reasons, NB_NUM_REASONS * 8));
  NB_HIP(hipMemset(c->ct.keys, 0, cap * 8)); NB_HIP(hipMemset(c->ct.ids, 0, cap * 4));
  NB_HIP(hipMemset(c->ct.counts, 0, (size_t)max_classes * 8));
  NB_HIP(hipMemset(c->ct.n_classes, 0, 4)); NB_HIP(hipMemset(c->ct.err, 0, 4)); NB_HIP(hipMemset(c->ct.reasons, 0, NB_NUM_REASONS * 8));
  c->ct.rows = nullptr;  // sized at index upload (needs W)
  // read-backs (class rows, counts, reasons) go through a pinned staging buffer and a stream of their own, both made here: the first
  // pinned allocation / first pageable copy of a process costs tens of milliseconds, which must not land in a caller's timed region
  NB_HIP(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
  c->h_stage_cap = (size_t)32 << 20;
  if (hipHostMalloc(&c->h_stage, c->h_stage_cap, hipHostMallocDefault) != hipSuccess) {   // (no pinned memory to be had: pageable staging, slower copies)
    (void)hipGetLastError();
    c->h_stage = std::malloc(c->h_stage_cap); c->h_stage_pinned = false;
    if (!c->h_stage) { set_error("nb_ctx_create: out of host memory"); return NB_ERR_INVALID; }
  }
  *out = c; return NB_OK;
}

int32_t nb_ctx_destroy(nb_ctx* c) {
  if (!c) return NB_OK;
  (void)hipSetDevice(c->device);
  (void)hipFree(c->d_slots); (void)hipFree(c->d_ctable); (void)hipFree(c->d_desc); (void)hipFree(c->d_packed); (void)hipFree(c->ct.keys); (void)hipFree(c->ct.ids); (void)hipFree(c->ct.rows);
  (void)hipFree(c->ct.counts); (void)hipFree(c->ct.spans); (void)hipFree(c->d_lists); (void)hipFree(c->d_list_counters); (void)hipFree(c->d_scan_counts); (void)hipFree(c->d_retry_list); (void)hipFree(c->d_retry_counts); (void)hipFree(c->d_retry_state); (void)hipFree(c->d_redo_reads); (void)hipFree(c->d_redo_lens); (void)hipFree(c->d_redo_out); (void)hipFree(c->d_redo_mates); (void)hipFree(c->d_redo_mate_lens); (void)hipFree(c->d_mate_res); (void)hipFree(c->d_lane_tail); if (c->h_stage) { if (c->h_stage_pinned) (void)hipHostFree(c->h_stage); else std::free(c->h_stage); } if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
  (void)hipFree(c->d_ph_table); (void)hipFree(c->d_ph_pilots); (void)hipFree(c->d_atable); (void)hipFree(c->d_akeys); (void)hipFree(c->d_pfilter); (void)hipFree(c->d_lfilter); (void)hipFree(c->d_text_rint); (void)hipFree(c->d_text); (void)hipFree(c->d_sparse); (void)hipFree(c->ct.n_classes); (void)hipFree(c->ct.err); (void)hipFree(c->ct.reasons); (void)hipFree(c->d_pct); (void)hipFree(c->d_ct);
  delete c; return NB_OK;
}

int32_t nb_ctx_sync(nb_ctx* c, void* stream) {
  if (!c) { set_error("nb_ctx_sync: null ctx"); return NB_ERR_INVALID; }
  NB_HIP(hipSetDevice(c->device)); NB_HIP(hipStreamSynchronize((hipStream_t)stream)); return NB_OK;
}

int32_t nb_index_build(int32_t n_seqs, const char* const* seqs, const int64_t* seq_lens, int32_t k, nb_index** out) {
  if (!out) { set_error("nb_index_build: null out"); return NB_ERR_INVALID; }
  HostIndex* h = nullptr;
  const int rc = build_host_index(n_seqs, seqs, seq_lens, k, &h);
  if (rc != NB_OK) return rc;
  *out = new nb_index{h}; return NB_OK;
}
int32_t nb_index_destroy(nb_index* ix) { if (ix) { delete ix->h; delete ix; } return NB_OK; }

int32_t nb_index_stats(const nb_index* ix, nb_index_stats_t* s) {
  if (!ix || !s) { set_error("nb_index_stats: null argument"); return NB_ERR_INVALID; }
  const HostIndex& h = *ix->h;
  s->n_seqs = h.n_seqs; s->n_kmers = h.n_kmers; s->n_canonical = h.n_canonical; s->table_slots = h.compact ? (int64_t)h.ctable.size() * 2 : (int64_t)h.buckets.size() * SLOTS_PER_BUCKET;
  s->n_colours = (int64_t)h.eq_sizes.size(); s->words_per_row = h.words;
  s->table_bytes = (int64_t)(h.buckets.size() * sizeof(Bucket) + h.ctable.size() * sizeof(CBucket)); s->compact = h.compact ? 1 : 0; s->words_internal = h.words_dev; s->ph_table_bytes = (int64_t)(h.ph_table.size() * sizeof(PHEntry)); s->ph_pilot_bytes = (int64_t)h.ph_pilots.size();
  s->anchor_bytes = (int64_t)(h.atable.size() * sizeof(AEntry) + h.akeys.size() * 4 + h.pfilter.size() * 4 + h.lfilter.size() * 4 + h.text_rint.size() * 4 + h.text.size() * 4 + h.sparse.size() * 8);
  s->device_bytes = s->table_bytes + (int64_t)(h.desc.size() * sizeof(ColDesc) + h.packed.size() * 8 + h.ph_table.size() * sizeof(PHEntry) + h.ph_pilots.size() + h.atable.size() * sizeof(AEntry) + h.akeys.size() * 4 + h.pfilter.size() * 4 + h.lfilter.size() * 4 + h.text_rint.size() * 4 + h.text.size() * 4 + h.sparse.size() * 8); s->k = h.k; s->uploaded = h.uploaded;
  return NB_OK;
}

int32_t nb_index_query(const nb_index* ix, const char* kmer, uint8_t* lext, uint8_t* rext, uint8_t* r_int, uint8_t* l_int, uint64_t* bits) {
  if (!ix || !kmer) { set_error("nb_index_query: null argument"); return NB_ERR_INVALID; }
  const HostIndex& h = *ix->h; uint64_t x = 0;
  for (int j = 0; j < h.k; ++j) {
    int c; switch (kmer[j]) { case 'A': case 'a': c = 0; break; case 'C': case 'c': c = 1; break; case 'G': case 'g': c = 2; break; case 'T': case 't': c = 3; break; default: return 0; }
    x |= (uint64_t)c << (2 * j);
  }
  const uint64_t y = rc_kmer_host(x, h.k), canon = x < y ? x : y; const int o = (x != canon);
  if (h.compact) {
    const CProbe pr = compact_probe(canon, h.chash);
    const uint32_t wantk[2] = {pr.want | (uint32_t)o, pr.want | (uint32_t)o | 2u}; const uint32_t bs[2] = {pr.b1, pr.b2};
    uint32_t pay = PAY_ABSENT;
    for (int t = 0; t < 2; ++t) for (int j = 0; j < 2; ++j) { const CEntry& e = h.ctable[bs[t]].e[j]; if (e.key == wantk[t] && e.pay != PAY_ABSENT) pay = e.pay; }
    if (h.has_ph) {
      // the perfect-hash table of the lane-per-read kernels must say the same about this k-mer (present / absent on this strand, extension
      // masks, class word): host-side cross-check of the second table, used by the index parity tests
      const PHProbe q = ph_probe(canon, h.ph);
      const PHEntry& e = h.ph_table[ph_slot(q, h.ph_pilots[q.group], h.ph)];
      const uint32_t d = e.key ^ q.want;
      const bool found = d < 4u, both = found && (d & 2u), mine = found && (both || (int)(d & 1u) == o);
      bool ok = mine == (pay != PAY_ABSENT);
      if (ok && mine && !both) {
        const ColDesc& cd = h.desc[pay >> 10];
        ok = (e.pay & 0x3FFu) == (pay & 0x3FFu) && (cd.nw == 1 ? (!(e.pay & (1u << 10)) && (e.pay >> 11) == cd.w0 && e.m == cd.m0) : ((e.pay & (1u << 10)) && (e.pay >> 11) == (pay >> 10)));
      }
      if (!ok) { set_error("nb_index_query: the perfect-hash table disagrees with the compact table"); return NB_ERR_INVALID; }
      if (found && !h.pfilter.empty()) {             // the presence filter of the scan kernel must let every m-mer of a table k-mer pass, in both orientations
        const uint64_t mmask = (1ull << (2 * h.pf_m)) - 1, both[2] = {x, y};
        for (int t = 0; t < 2; ++t) for (int sh = 0; sh < h.pf_s; ++sh) {
            const uint64_t mm = (both[t] >> (2 * sh)) & mmask;
            const uint32_t h0 = pf_hash(mm), m0 = pf_mask(h0), li = lf_index(mm);
            if ((h.pfilter[pf_word(h0, h.pf_wbits)] & m0) != m0 || (!h.lfilter.empty() && !((h.lfilter[li >> 5] >> (li & 31u)) & 1u))) { set_error("nb_index_query: the presence filter rejects an m-mer of a k-mer of the table"); return NB_ERR_INVALID; }
          }
      }
    }
    if (pay == PAY_ABSENT) return 0;
    if (rext) *rext = pay & 15; if (lext) *lext = (pay >> 4) & 15; if (r_int) *r_int = (pay >> 8) & 1; if (l_int) *l_int = (pay >> 9) & 1;
    if (bits) std::memcpy(bits, h.eq_rows.data() + (size_t)(pay >> 10) * h.words, (size_t)h.words * 8);
    return 1;
  }
  uint32_t b = kmer_bucket(canon, h.hash);
  for (;;) {
    const Bucket& bk = h.buckets[b];
    for (int j = 0; j < SLOTS_PER_BUCKET; ++j) if ((bk.slot[j].key & KEY_MASK) == canon) {
        const uint32_t p = bk.slot[j].pay[o]; if (p == PAY_ABSENT) return 0;
        if (rext) *rext = p & 15; if (lext) *lext = (p >> 4) & 15; if (r_int) *r_int = (p >> 8) & 1; if (l_int) *l_int = (p >> 9) & 1;
        if (bits) std::memcpy(bits, h.eq_rows.data() + (size_t)(p >> 10) * h.words, (size_t)h.words * 8);
        return 1;
      }
    if (!(bk.slot[0].key & KEY_OVERFLOW)) return 0;
    b = (b + 1 == h.hash.n_buckets) ? 0u : b + 1;
  }
}

int32_t nb_index_anchor_query(const nb_index* ix, const char* kmer, int32_t run, int64_t* tpos, int32_t* room, uint64_t* colour_and) {
  if (!ix || !kmer || run < 0) { set_error("nb_index_anchor_query: bad arguments"); return NB_ERR_INVALID; }
  const HostIndex& h = *ix->h;
  if (!h.has_anchor) return 0;
  uint64_t x = 0;
  for (int j = 0; j < h.k; ++j) {
    int c; switch (kmer[j]) { case 'A': case 'a': c = 0; break; case 'C': case 'c': c = 1; break; case 'G': case 'g': c = 2; break; case 'T': case 't': c = 3; break; default: return 0; }
    x |= (uint64_t)c << (2 * j);
  }
  const uint64_t y = rc_kmer_host(x, h.k), canon = x < y ? x : y; const uint32_t o = (x != canon) ? 1u : 0u;
  const PHProbe q = ph_probe(canon, h.ph);
  const AEntry& e = h.atable[ph_slot(q, h.ph_pilots[q.group], h.ph)];
  const uint32_t d = e.key ^ q.want;
  if (d >= 4u || (d & 2u) || (d & 1u) != o) return 0;
  const uint32_t rm = e.room & ANCHOR_ROOM_MASK, w0 = e.room >> 16;
  if (tpos) *tpos = (int64_t)e.tpos;
  if (room) *room = (int32_t)rm;
  if (e.room & ANCHOR_WIDE) return 2;
  if (colour_and) {
    const uint32_t n = (uint32_t)std::min<int64_t>(run, (int64_t)rm) + 1u;
    int lv = 0; while ((2u << lv) <= n) ++lv;
    const size_t NW = (size_t)h.anchor_nw;
    const uint64_t* sp = h.sparse.data() + ((size_t)lv * (size_t)h.text_len + e.tpos) * NW;
    std::memset(colour_and, 0, (size_t)h.words * 8);
    for (size_t w = 0; w < NW; ++w) {
      const uint64_t v = sp[w] & sp[(size_t)(n - (1u << lv)) * NW + w];
      for (int b = 0; b < 64; ++b) if ((v >> b) & 1ull) { const int32_t orig = h.inv_perm[((size_t)w0 + w) * 64 + (size_t)b]; if (orig >= 0) colour_and[orig >> 6] |= 1ull << (orig & 63); }
    }
  }
  return 1;
}

int32_t nb_index_upload(nb_ctx* c, nb_index* ix) {
  if (!c || !ix) { set_error("nb_index_upload: null argument"); return NB_ERR_INVALID; }
  NB_HIP(hipSetDevice(c->device));
  HostIndex& h = *ix->h;
  (void)hipFree(c->d_slots); (void)hipFree(c->d_ctable); (void)hipFree(c->d_desc); (void)hipFree(c->d_packed); (void)hipFree(c->ct.rows);
  c->d_slots = nullptr; c->d_ctable = nullptr; c->d_desc = nullptr; c->d_packed = nullptr; c->ct.rows = nullptr;
  c->compact = h.compact; c->chash = h.chash; c->max_desc_nw = h.max_desc_nw;
  (void)hipFree(c->d_ph_table); (void)hipFree(c->d_ph_pilots); c->d_ph_table = nullptr; c->d_ph_pilots = nullptr; c->has_ph = h.has_ph; c->ph = h.ph;
  if (h.has_ph) {
    NB_HIP(hipMalloc((void**)&c->d_ph_table, h.ph_table.size() * sizeof(PHEntry))); NB_HIP(hipMalloc((void**)&c->d_ph_pilots, h.ph_pilots.size()));
    NB_HIP(hipMemcpy(c->d_ph_table, h.ph_table.data(), h.ph_table.size() * sizeof(PHEntry), hipMemcpyHostToDevice));
    NB_HIP(hipMemcpy(c->d_ph_pilots, h.ph_pilots.data(), h.ph_pilots.size(), hipMemcpyHostToDevice));
  }
  (void)hipFree(c->d_atable); (void)hipFree(c->d_text); (void)hipFree(c->d_sparse); (void)hipFree(c->d_akeys); (void)hipFree(c->d_pfilter); (void)hipFree(c->d_lfilter); (void)hipFree(c->d_text_rint); c->d_atable = nullptr; c->d_text = nullptr; c->d_sparse = nullptr; c->d_akeys = nullptr; c->d_pfilter = nullptr; c->d_lfilter = nullptr; c->d_text_rint = nullptr;
  c->has_anchor = h.has_anchor; c->pf_wbits = h.pf_wbits; c->pf_m = h.pf_m; c->pf_s = h.pf_s; c->text_len = h.text_len; c->anchor_nw = h.anchor_nw;
  if (h.has_anchor) {
    // (the sparse table is the one large piece — 8 or 16 bytes x 8 levels per library base: if it does not fit next to the caller's
    //  buffers the index still works, through the kernels that do not need it)
    const bool got = hipMalloc((void**)&c->d_atable, h.atable.size() * sizeof(AEntry)) == hipSuccess && hipMalloc((void**)&c->d_text, h.text.size() * 4) == hipSuccess &&
                     hipMalloc((void**)&c->d_akeys, h.akeys.size() * 4) == hipSuccess && hipMalloc((void**)&c->d_pfilter, h.pfilter.size() * 4) == hipSuccess &&
                     hipMalloc((void**)&c->d_sparse, h.sparse.size() * 8) == hipSuccess;
    if (!got) {
      (void)hipGetLastError();
      (void)hipFree(c->d_atable); (void)hipFree(c->d_text); (void)hipFree(c->d_sparse); (void)hipFree(c->d_akeys); (void)hipFree(c->d_pfilter); (void)hipFree(c->d_lfilter); (void)hipFree(c->d_text_rint); c->d_atable = nullptr; c->d_text = nullptr; c->d_sparse = nullptr; c->d_akeys = nullptr; c->d_pfilter = nullptr; c->d_lfilter = nullptr; c->d_text_rint = nullptr;
      c->has_anchor = false;
    } else {
      NB_HIP(hipMemcpy(c->d_akeys, h.akeys.data(), h.akeys.size() * 4, hipMemcpyHostToDevice));
      NB_HIP(hipMemcpy(c->d_pfilter, h.pfilter.data(), h.pfilter.size() * 4, hipMemcpyHostToDevice));
      NB_HIP(hipMalloc((void**)&c->d_text_rint, h.text_rint.size() * 4)); NB_HIP(hipMemcpy(c->d_text_rint, h.text_rint.data(), h.text_rint.size() * 4, hipMemcpyHostToDevice));
      if (!h.lfilter.empty()) { NB_HIP(hipMalloc((void**)&c->d_lfilter, h.lfilter.size() * 4)); NB_HIP(hipMemcpy(c->d_lfilter, h.lfilter.data(), h.lfilter.size() * 4, hipMemcpyHostToDevice)); }
      NB_HIP(hipMemcpy(c->d_atable, h.atable.data(), h.atable.size() * sizeof(AEntry), hipMemcpyHostToDevice));
      NB_HIP(hipMemcpy(c->d_text, h.text.data(), h.text.size() * 4, hipMemcpyHostToDevice));
      NB_HIP(hipMemcpy(c->d_sparse, h.sparse.data(), h.sparse.size() * 8, hipMemcpyHostToDevice));
    }
  }
  if (h.compact) {
    NB_HIP(hipMalloc((void**)&c->d_ctable, h.ctable.size() * sizeof(CBucket)));
    NB_HIP(hipMemcpy(c->d_ctable, h.ctable.data(), h.ctable.size() * sizeof(CBucket), hipMemcpyHostToDevice));
  } else
  NB_HIP(hipMalloc((void**)&c->d_slots, h.buckets.size() * sizeof(Bucket)));
  NB_HIP(hipMalloc((void**)&c->d_desc, h.desc.size() * sizeof(ColDesc)));
  NB_HIP(hipMalloc((void**)&c->d_packed, h.packed.size() * 8));
  if (!h.compact) NB_HIP(hipMemcpy(c->d_slots, h.buckets.data(), h.buckets.size() * sizeof(Bucket), hipMemcpyHostToDevice));
  NB_HIP(hipMemcpy(c->d_desc, h.desc.data(), h.desc.size() * sizeof(ColDesc), hipMemcpyHostToDevice));
  NB_HIP(hipMemcpy(c->d_packed, h.packed.data(), h.packed.size() * 8, hipMemcpyHostToDevice));
  NB_HIP(hipMalloc((void**)&c->ct.rows, (size_t)c->max_classes * (size_t)h.words_dev * 8));
  if (!c->d_ct) NB_HIP(hipMalloc((void**)&c->d_ct, sizeof(DevClasses)));
  NB_HIP(hipMemcpy(c->d_ct, &c->ct, sizeof(DevClasses), hipMemcpyHostToDevice));
  c->hash = h.hash; c->W = h.words_dev; c->W_ext = h.words; c->inv_perm = h.inv_perm; c->K = h.k; c->n_seqs = h.n_seqs;
  c->index_bytes = (int64_t)(h.buckets.size() * sizeof(Bucket) + h.ctable.size() * sizeof(CBucket) + h.desc.size() * sizeof(ColDesc) + h.packed.size() * 8 + h.ph_table.size() * sizeof(PHEntry) + h.ph_pilots.size() + h.atable.size() * sizeof(AEntry) + h.akeys.size() * 4 + h.pfilter.size() * 4 + h.lfilter.size() * 4 + h.text_rint.size() * 4 + h.text.size() * 4 + h.sparse.size() * 8); h.uploaded = 1;
  return nb_results_reset(c, nullptr);
}

int32_t nb_index_filter_maybe(const nb_index* ix, const char* seq, int32_t len, int32_t* m_out, int32_t* lds_stage_out) {
  if (!ix || !seq) { set_error("nb_index_filter_maybe: null argument"); return NB_ERR_INVALID; }
  const HostIndex& h = *ix->h;
  if (m_out) *m_out = h.pf_m;
  if (lds_stage_out) *lds_stage_out = h.lfilter.empty() ? 0 : 1;
  if (h.pfilter.empty()) return 2;
  if (len == 0) return 1;                            // (parameters only)
  if (len < h.pf_m) { set_error("nb_index_filter_maybe: sequence shorter than the filter's m"); return NB_ERR_INVALID; }
  const uint64_t mmask = (1ull << (2 * h.pf_m)) - 1;
  for (int s0 = 0; s0 + h.pf_m <= len; ++s0) {
    uint64_t x = 0;
    for (int j = 0; j < h.pf_m; ++j) {
      int c; switch (seq[s0 + j]) { case 'A': case 'a': c = 0; break; case 'C': case 'c': c = 1; break; case 'G': case 'g': c = 2; break; case 'T': case 't': c = 3; break; default: return 1; }
      x |= (uint64_t)c << (2 * j);
    }
    x &= mmask;
    const uint32_t hh = pf_hash(x), m = pf_mask(hh), li = lf_index(x);
    if ((h.pfilter[pf_word(hh, h.pf_wbits)] & m) != m) return 0;
    if (!h.lfilter.empty() && !((h.lfilter[li >> 5] >> (li & 31u)) & 1u)) return 0;
  }
  return 1;
}

int64_t nb_packed_stride_words(int32_t max_len) { return (int64_t)((max_len + 31) / 32 + (max_len + 63) / 64); }

int32_t nb_pack_reads(const char* ascii, const int64_t* off, int64_t n, int32_t max_len, uint64_t* out, uint16_t* out_lens) {
  if (!ascii || !off || !out || n < 0 || max_len < 1 || max_len > 65535) { set_error("nb_pack_reads: bad arguments"); return NB_ERR_INVALID; }
  const int nwb = (max_len + 31) / 32; const int64_t stride = nb_packed_stride_words(max_len);
  for (int64_t i = 0; i < n; ++i) {
    const int64_t L = off[i + 1] - off[i];
    if (L < 0 || L > max_len) { set_error("nb_pack_reads: read longer than max_len"); return NB_ERR_INVALID; }
    uint64_t* w = out + i * stride; std::memset(w, 0, (size_t)stride * 8);
    const char* p = ascii + off[i];
    for (int64_t j = 0; j < L; ++j) {
      uint64_t c;
      switch (p[j]) { case 'A': case 'a': c = 0; break; case 'C': case 'c': c = 1; break; case 'G': case 'g': c = 2; break; case 'T': case 't': c = 3; break;
        default: c = 0; w[nwb + (j >> 6)] |= 1ull << (j & 63); }
      w[j >> 5] |= c << ((j & 31) * 2);
    }
    if (out_lens) out_lens[i] = (uint16_t)L;
  }
  return NB_OK;
}

int32_t nb_align_batch(nb_ctx* c, const nb_config* cfg, const uint64_t* d_reads, const uint16_t* d_lens, int32_t max_len,
                       const uint64_t* d_mates, const uint16_t* d_mate_lens, int32_t mate_max_len, int64_t n,
                       uint32_t* d_out_class, uint32_t* d_out_info, uint64_t* d_out_cov, void* stream) {
  if (!c || !cfg || !d_reads || n < 0 || max_len < 1) { set_error("nb_align_batch: bad arguments"); return NB_ERR_INVALID; }
  if (!c->d_slots && !c->d_ctable) { set_error("nb_align_batch: no index uploaded to this ctx"); return NB_ERR_INVALID; }
  if (cfg->k != c->K) { set_error("nb_align_batch: cfg.k differs from the uploaded index's k"); return NB_ERR_INVALID; }
  if (cfg->strand_filter < 0 || cfg->strand_filter > 3 || cfg->intersect_level < 0 || cfg->intersect_level > 2) { set_error("nb_align_batch: bad strand_filter/intersect_level"); return NB_ERR_INVALID; }
  if (cfg->num_mismatches < 0 || cfg->num_mismatches > 64) { set_error("nb_align_batch: num_mismatches must be in 0..64"); return NB_ERR_INVALID; }
  if (d_mates && mate_max_len < 1) { set_error("nb_align_batch: mates given but mate_max_len < 1"); return NB_ERR_INVALID; }
  if (!d_mates) mate_max_len = 0;
  const int lmax = max_len > mate_max_len ? max_len : mate_max_len;
  const int positions = lmax - c->K + 1;
  if (positions > NB_MAX_POSITIONS) { set_error("nb_align_batch: read longer than NB_MAX_POSITIONS + k - 1"); return NB_ERR_UNSUPPORTED; }
  if (c->W > 64 * 16) { set_error("nb_align_batch: the library needs more than 1 024 words per class row (about 65 000 sequences)"); return NB_ERR_UNSUPPORTED; }
  if (n == 0) return NB_OK;
  NB_HIP(hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)stream;
  c->last_stream = s; c->last_stream_set = true;
  int rc = ensure_pct_table(c, cfg->score_percent, lmax, s); if (rc != NB_OK) return rc;
  DevIndex ix{c->d_slots, c->d_ctable, c->d_desc, c->d_packed, c->hash, c->chash, c->W, c->K};
  BatchArgs a{};
  a.reads = d_reads; a.lens = d_lens; a.max_len = max_len; a.stride = (int)nb_packed_stride_words(max_len); a.nwb = (max_len + 31) / 32;
  a.mates = d_mates; a.mate_lens = d_mate_lens; a.mate_max_len = mate_max_len;
  a.mstride = mate_max_len > 0 ? (int)nb_packed_stride_words(mate_max_len) : 0; a.mnwb = (mate_max_len + 31) / 32;
  a.n = n; a.out_class = d_out_class; a.out_info = d_out_info; a.out_cov = d_out_cov; a.pct_min = c->d_pct; a.pct_fixed = c->pct_host[(size_t)max_len];
  a.score_threshold = cfg->score_threshold; a.score_filter = cfg->score_filter; a.num_mismatches = cfg->num_mismatches;
  a.discard_multiple = cfg->discard_multiple_matches; a.intersect_level = cfg->intersect_level;
  a.require_valid_pair = cfg->require_valid_pair; a.strand_filter = cfg->strand_filter;
  const int nr = positions < 1 ? 1 : (positions + 63) / 64;
  c->last_kernels.clear();
  // Lane-per-read path (nb_lane.hip): single-end, exact mode, compact table, every class inside one 64-bit word. Two launches map the
  // batch (forward orientation, then the reads whose better strand is the reverse one); what the online walk cannot finish exactly is
  // gathered and goes through the wave-per-read kernel below.
  const char* lane_env = std::getenv("NB_LANE");
  const char* anchor_env = std::getenv("NB_ANCHOR");
  int lane_wpb = 0;
  // anchor-extend kernel (nb_anchor.hip) when the index carries the anchor structures: far fewer lookups per read; else the lane-per-read
  // kernels of nb_lane.hip (first level of the table must fit LDS)
  const bool use_anchor = c->has_anchor && !(anchor_env && anchor_env[0] == '0') && anchor_fits(c, max_len);
  // (tolerated mismatches, num_mismatches 1 .. 31: the anchor kernels only — they pass over a base while they follow the text; nb_anchor.hip)
  const bool mm_ok = cfg->num_mismatches == 0 || (use_anchor && cfg->num_mismatches <= 31 && !(std::getenv("NB_ANCHOR_MM") && std::getenv("NB_ANCHOR_MM")[0] == '0'));
  if (!d_mates && mm_ok && c->has_ph && positions >= 1 && n < (1ll << 31) && !(lane_env && lane_env[0] == '0') &&
      (use_anchor || lane_lds_bytes(c, max_len, &lane_wpb) > 0)) {
    const int64_t stride = a.stride;
    if (c->lists_cap < n) {
      (void)hipFree(c->d_lists); (void)hipFree(c->d_redo_lens); (void)hipFree(c->d_redo_out); c->d_lists = nullptr; c->d_redo_lens = nullptr; c->d_redo_out = nullptr;
      // [redo list: n][scan / reverse list: n + one batch per wave of slack: the anchor kernels give every workgroup its own segment of it]
      c->scan_list_cap = n + (int64_t)c->n_cu * 4 * 12 * 64;
      NB_HIP(hipMalloc((void**)&c->d_lists, (size_t)(n + c->scan_list_cap) * 4)); NB_HIP(hipMalloc((void**)&c->d_redo_lens, (size_t)n * 2)); NB_HIP(hipMalloc((void**)&c->d_redo_out, (size_t)n * 16));
      c->lists_cap = n;
    }
    if (!c->d_scan_counts) { NB_HIP(hipMalloc((void**)&c->d_scan_counts, (size_t)c->n_cu * 4 * 4)); c->scan_counts_cap = (int64_t)c->n_cu * 4; }
    if (c->redo_words_cap < n * stride) {
      (void)hipFree(c->d_redo_reads); c->d_redo_reads = nullptr;
      NB_HIP(hipMalloc((void**)&c->d_redo_reads, (size_t)(n * stride) * 8)); c->redo_words_cap = n * stride;
    }
    if (!c->d_list_counters) NB_HIP(hipMalloc((void**)&c->d_list_counters, 8));
    LaneLists ll{c->d_lists + c->lists_cap, c->d_lists, c->d_list_counters};
    rc = use_anchor ? launch_anchor(c, nr, ix, a, ll, s) : launch_lane(c, nr, ix, a, ll, s); if (rc != NB_OK) return rc;
    { char buf[240];
      if (use_anchor) {
        char ak[96]; std::snprintf(ak, sizeof ak, "nb_anchor_kernel<%d, %d, %s, false, %s>", nr > 5 ? 5 : nr, c->anchor_nw, anchor_pilots_in_lds(c, max_len) ? "true" : "false", cfg->num_mismatches > 0 ? "true" : "false");
        std::snprintf(buf, sizeof buf, "%s;%s%snb_anchor_scan_kernel<%d, %d, false>", ak, c->anchor_retry_ran ? ak : "", c->anchor_retry_ran ? ";" : "",   // (second launch: the stragglers' retry list)
                      nr > 5 ? 5 : nr, c->pf_s);
      }
      else std::snprintf(buf, sizeof buf, "nb_lane_kernel<%d, 0>;nb_lane_kernel<%d, 1>", nr > 5 ? 5 : nr, nr > 5 ? 5 : nr);
      c->last_kernels = buf; }
    const std::string buf_names = c->last_kernels;
    rc = launch_lane_gather(c, a, ll, c->d_redo_reads, c->d_redo_lens, s); if (rc != NB_OK) return rc;
    BatchArgs b = a;
    b.reads = c->d_redo_reads; b.lens = a.lens ? c->d_redo_lens : nullptr; b.n_dev = c->d_list_counters + 1;
    b.out_class = a.out_class ? c->d_redo_out : nullptr; b.out_info = a.out_info ? c->d_redo_out + c->lists_cap : nullptr;
    b.out_cov = a.out_cov ? reinterpret_cast<uint64_t*>(c->d_redo_out + 2 * c->lists_cap) : nullptr;
    if (cfg->num_mismatches > 0) {                                            // the redo reads of a run with tolerated mismatches: the LDS-staged mapper
      if (c->compact) rc = (c->W <= 64) ? dispatch_general<1, true>(c, nr, positions, false, ix, b, s) : (c->W <= 256 ? dispatch_general<4, true>(c, nr, positions, false, ix, b, s) : dispatch_general<16, true>(c, nr, positions, false, ix, b, s));
      else rc = (c->W <= 64) ? dispatch_general<1, false>(c, nr, positions, false, ix, b, s) : (c->W <= 256 ? dispatch_general<4, false>(c, nr, positions, false, ix, b, s) : dispatch_general<16, false>(c, nr, positions, false, ix, b, s));
      c->last_kernels = std::string(buf_names) + ";" + c->last_kernels;
    } else
    if (c->compact) rc = (c->W <= 64) ? dispatch_se<1, true>(c, nr, positions, ix, b, s) : (c->W <= 256 ? dispatch_se<4, true>(c, nr, positions, ix, b, s) : dispatch_se<16, true>(c, nr, positions, ix, b, s));
    else rc = (c->W <= 64) ? dispatch_se<1, false>(c, nr, positions, ix, b, s) : (c->W <= 256 ? dispatch_se<4, false>(c, nr, positions, ix, b, s) : dispatch_se<16, false>(c, nr, positions, ix, b, s));   // (anchors next to the wide table)
    if (rc != NB_OK) return rc;
    BatchArgs sc = a; sc.n = c->lists_cap;
    return launch_lane_scatter(c, sc, ll, c->d_redo_out, s);
  }
  // Paired templates in exact mode, index with anchor structures: each mate through the anchor kernels (raw alignments), SPEC §4 for the
  // pair in nb_anchor_pair_kernel; pairs it cannot settle are gathered and go through the general kernel below (nb_anchor.hip).
  if (d_mates && (cfg->num_mismatches == 0 || (cfg->num_mismatches <= 31 && !(std::getenv("NB_ANCHOR_MM") && std::getenv("NB_ANCHOR_MM")[0] == '0'))) && c->has_anchor && !(anchor_env && anchor_env[0] == '0') && !(lane_env && lane_env[0] == '0') && n < (1ll << 31) &&
      max_len >= c->K && mate_max_len >= c->K && anchor_fits(c, max_len) && anchor_fits(c, mate_max_len)) {
    if (c->lists_cap < n) {
      (void)hipFree(c->d_lists); (void)hipFree(c->d_redo_lens); (void)hipFree(c->d_redo_out); c->d_lists = nullptr; c->d_redo_lens = nullptr; c->d_redo_out = nullptr;
      c->scan_list_cap = n + (int64_t)c->n_cu * 4 * 12 * 64;
      NB_HIP(hipMalloc((void**)&c->d_lists, (size_t)(n + c->scan_list_cap) * 4)); NB_HIP(hipMalloc((void**)&c->d_redo_lens, (size_t)n * 2)); NB_HIP(hipMalloc((void**)&c->d_redo_out, (size_t)n * 16));
      c->lists_cap = n; c->mate_res_cap = 0;                                   // (d_redo_mate_lens follows lists_cap)
    }
    if (!c->d_scan_counts) { NB_HIP(hipMalloc((void**)&c->d_scan_counts, (size_t)c->n_cu * 4 * 4)); c->scan_counts_cap = (int64_t)c->n_cu * 4; }
    if (c->mate_res_cap < n) {
      (void)hipFree(c->d_mate_res); (void)hipFree(c->d_redo_mate_lens); c->d_mate_res = nullptr; c->d_redo_mate_lens = nullptr;
      NB_HIP(hipMalloc(&c->d_mate_res, (size_t)n * 2 * (size_t)anchor_mate_res_bytes())); NB_HIP(hipMalloc((void**)&c->d_redo_mate_lens, (size_t)c->lists_cap * 2));
      c->mate_res_cap = n;
    }
    if (c->redo_words_cap < n * (int64_t)a.stride) {
      (void)hipFree(c->d_redo_reads); c->d_redo_reads = nullptr;
      NB_HIP(hipMalloc((void**)&c->d_redo_reads, (size_t)(n * a.stride) * 8)); c->redo_words_cap = n * a.stride;
    }
    if (c->redo_mate_words_cap < n * (int64_t)a.mstride) {
      (void)hipFree(c->d_redo_mates); c->d_redo_mates = nullptr;
      NB_HIP(hipMalloc((void**)&c->d_redo_mates, (size_t)(n * a.mstride) * 8)); c->redo_mate_words_cap = n * a.mstride;
    }
    if (!c->d_list_counters) NB_HIP(hipMalloc((void**)&c->d_list_counters, 8));
    LaneLists ll{c->d_lists + c->lists_cap, c->d_lists, c->d_list_counters};
    rc = launch_anchor_pairs(c, ix, a, ll, c->d_mate_res, s); if (rc != NB_OK) return rc;
    c->last_kernels = "nb_anchor_kernel<mates>;nb_anchor_scan_kernel<mates>;nb_anchor_pair_kernel";
    rc = launch_lane_gather(c, a, ll, c->d_redo_reads, c->d_redo_lens, s); if (rc != NB_OK) return rc;
    BatchArgs am = a; am.reads = a.mates; am.lens = a.mate_lens; am.stride = a.mstride;
    rc = launch_lane_gather(c, am, ll, c->d_redo_mates, c->d_redo_mate_lens, s); if (rc != NB_OK) return rc;
    BatchArgs b = a;
    b.reads = c->d_redo_reads; b.lens = a.lens ? c->d_redo_lens : nullptr; b.mates = c->d_redo_mates; b.mate_lens = a.mate_lens ? c->d_redo_mate_lens : nullptr;
    b.n_dev = c->d_list_counters + 1;
    b.out_class = a.out_class ? c->d_redo_out : nullptr; b.out_info = a.out_info ? c->d_redo_out + c->lists_cap : nullptr;
    b.out_cov = a.out_cov ? reinterpret_cast<uint64_t*>(c->d_redo_out + 2 * c->lists_cap) : nullptr;
    const bool exact_redo = cfg->num_mismatches == 0;
    if (c->compact) rc = (c->W <= 64) ? dispatch_general<1, true>(c, nr, positions, exact_redo, ix, b, s) : (c->W <= 256 ? dispatch_general<4, true>(c, nr, positions, exact_redo, ix, b, s) : dispatch_general<16, true>(c, nr, positions, exact_redo, ix, b, s));
    else rc = (c->W <= 64) ? dispatch_general<1, false>(c, nr, positions, exact_redo, ix, b, s) : (c->W <= 256 ? dispatch_general<4, false>(c, nr, positions, exact_redo, ix, b, s) : dispatch_general<16, false>(c, nr, positions, exact_redo, ix, b, s));
    if (rc != NB_OK) return rc;
    c->last_kernels = "nb_anchor_kernel<mates>;nb_anchor_scan_kernel<mates>;nb_anchor_pair_kernel;" + c->last_kernels;
    BatchArgs sc = a; sc.n = c->lists_cap;
    return launch_lane_scatter(c, sc, ll, c->d_redo_out, s);
  }
  if (d_mates || cfg->num_mismatches > 0)   // pairs and tolerated mismatches take the general (eager, LDS-staged) kernel
  {
    const bool exact = cfg->num_mismatches == 0;
    if (c->compact) return (c->W <= 64) ? dispatch_general<1, true>(c, nr, positions, exact, ix, a, s) : (c->W <= 256 ? dispatch_general<4, true>(c, nr, positions, exact, ix, a, s) : dispatch_general<16, true>(c, nr, positions, exact, ix, a, s));
    return (c->W <= 64) ? dispatch_general<1, false>(c, nr, positions, exact, ix, a, s) : (c->W <= 256 ? dispatch_general<4, false>(c, nr, positions, exact, ix, a, s) : dispatch_general<16, false>(c, nr, positions, exact, ix, a, s));
  }
  if (c->compact) return (c->W <= 64) ? dispatch_se<1, true>(c, nr, positions, ix, a, s) : (c->W <= 256 ? dispatch_se<4, true>(c, nr, positions, ix, a, s) : dispatch_se<16, true>(c, nr, positions, ix, a, s));
  return (c->W <= 64) ? dispatch_se<1, false>(c, nr, positions, ix, a, s) : (c->W <= 256 ? dispatch_se<4, false>(c, nr, positions, ix, a, s) : dispatch_se<16, false>(c, nr, positions, ix, a, s));
}

// Read-backs go through a stream of their own (non-blocking: none of the legacy null stream's implicit synchronisation with every other
// stream of the process), after waiting for the stream the last batch was launched on — what the synchronous copies used to imply.
static int readback(nb_ctx* c, void* dst, const void* src, size_t bytes) {
  if (c->last_stream_set) NB_HIP(hipStreamSynchronize(c->last_stream));
  for (size_t done = 0; done < bytes;) {                      // device -> pinned staging -> the caller's (pageable) memory
    const size_t m = std::min(bytes - done, c->h_stage_cap);
    NB_HIP(hipMemcpyAsync(c->h_stage, static_cast<const char*>(src) + done, m, hipMemcpyDeviceToHost, c->copy_stream));
    NB_HIP(hipStreamSynchronize(c->copy_stream));
    std::memcpy(static_cast<char*>(dst) + done, c->h_stage, m);
    done += m;
  }
  return NB_OK;
}

int32_t nb_classes_count(nb_ctx* c, int64_t* n) {
  if (!c || !n) { set_error("nb_classes_count: null argument"); return NB_ERR_INVALID; }
  NB_HIP(hipSetDevice(c->device));
  unsigned int v = 0, e = 0;
  { int r = readback(c, &v, c->ct.n_classes, 4); if (r != NB_OK) return r; r = readback(c, &e, c->ct.err, 4); if (r != NB_OK) return r; }
  if (e & 2u) { set_error("class table: publish wait timed out"); return NB_ERR_HIP; }
  if ((e & 1u) || v > (unsigned int)c->max_classes) { set_error("class table overflow: raise max_classes"); return NB_ERR_OVERFLOW; }
  *n = (int64_t)v; return NB_OK;
}

// rows of the class table are kept in INTERNAL sequence numbering on the device (nb_internal.h, ColDesc); callers get bitsets over the
// ORIGINAL sequence indices, words_per_row = ceil(n_seqs / 64) words per row
static int read_rows_external(nb_ctx* c, int64_t first, int64_t count, uint64_t* rows) {
  // device -> pinned staging buffer of the ctx (whole rows per chunk) -> renumbered into the caller's rows by up to 16 threads
  const bool timing = std::getenv("NB_TIMING") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  double t_copy = 0;
  if (c->last_stream_set) NB_HIP(hipStreamSynchronize(c->last_stream));
  const int64_t per_chunk = std::max<int64_t>(1, (int64_t)(c->h_stage_cap / ((size_t)c->W * 8)));
  const uint64_t* tmp = static_cast<const uint64_t*>(c->h_stage);
  if ((size_t)c->W * 8 > c->h_stage_cap) { set_error("nb_classes_read: a class row is larger than the staging buffer"); return NB_ERR_UNSUPPORTED; }
  for (int64_t base = 0; base < count; base += per_chunk) {
    const int64_t m = std::min(per_chunk, count - base);
    const auto a = std::chrono::steady_clock::now();
    NB_HIP(hipMemcpyAsync(c->h_stage, c->ct.rows + (size_t)(first + base) * (size_t)c->W, (size_t)m * (size_t)c->W * 8, hipMemcpyDeviceToHost, c->copy_stream));
    NB_HIP(hipStreamSynchronize(c->copy_stream));
    t_copy += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count();
    const int T = (int)std::max<int64_t>(1, std::min<int64_t>(16, std::min<int64_t>((int64_t)std::thread::hardware_concurrency(), m * (int64_t)c->W / (1 << 16))));
    auto work = [&](int64_t lo, int64_t hi) {
      std::memset(rows + (size_t)(base + lo) * (size_t)c->W_ext, 0, (size_t)(hi - lo) * (size_t)c->W_ext * 8);
      for (int64_t r = lo; r < hi; ++r) {
        const uint64_t* in = tmp + (size_t)r * (size_t)c->W; uint64_t* out = rows + (size_t)(base + r) * (size_t)c->W_ext;
        for (int w = 0; w < c->W; ++w) for (uint64_t v = in[w]; v; v &= v - 1) {
            const int32_t o = c->inv_perm[(size_t)w * 64 + (size_t)__builtin_ctzll(v)];
            if (o >= 0) out[o >> 6] |= 1ull << (o & 63);
          }
      }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back(work, m * t / T, m * (t + 1) / T);
    work(0, m / T);
    for (auto& x : th) x.join();
  }
  if (timing) std::fprintf(stderr, "[nb_classes_read] %lld rows x %d words: %.2f ms, of which copies %.2f ms\n", (long long)count, c->W,
                           std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), t_copy);
  return NB_OK;
}

static void renumber_rows(const nb_ctx* c, const uint64_t* in_rows, int64_t count, uint64_t* rows) {   // internal -> original numbering, up to 16 threads
  const int T = (int)std::max<int64_t>(1, std::min<int64_t>(16, std::min<int64_t>((int64_t)std::thread::hardware_concurrency(), count * (int64_t)c->W / (1 << 16))));
  auto work = [&](int64_t lo, int64_t hi) {
    std::memset(rows + (size_t)lo * (size_t)c->W_ext, 0, (size_t)(hi - lo) * (size_t)c->W_ext * 8);
    for (int64_t r = lo; r < hi; ++r) {
      const uint64_t* in = in_rows + (size_t)r * (size_t)c->W; uint64_t* out = rows + (size_t)r * (size_t)c->W_ext;
      for (int w = 0; w < c->W; ++w) for (uint64_t v = in[w]; v; v &= v - 1) {
          const int32_t o = c->inv_perm[(size_t)w * 64 + (size_t)__builtin_ctzll(v)];
          if (o >= 0) out[o >> 6] |= 1ull << (o & 63);
        }
    }
  };
  std::vector<std::thread> th;
  for (int t = 1; t < T; ++t) th.emplace_back(work, count * t / T, count * (t + 1) / T);
  work(0, count / T);
  for (auto& x : th) x.join();
}

int32_t nb_classes_copy_device(nb_ctx* c, int64_t n, uint64_t* d_rows, int64_t* d_counts, void* stream) {
  if (!c || n < 0) { set_error("nb_classes_copy_device: bad arguments"); return NB_ERR_INVALID; }
  int64_t have = 0; const int rc = nb_classes_count(c, &have); if (rc != NB_OK) return rc;
  if (n > have) { set_error("nb_classes_copy_device: more rows asked for than there are classes"); return NB_ERR_INVALID; }
  hipStream_t s = (hipStream_t)stream;
  if (n && d_rows) NB_HIP(hipMemcpyAsync(d_rows, c->ct.rows, (size_t)n * (size_t)c->W * 8, hipMemcpyDeviceToDevice, s));
  if (n && d_counts) NB_HIP(hipMemcpyAsync(d_counts, c->ct.counts, (size_t)n * 8, hipMemcpyDeviceToDevice, s));
  return NB_OK;
}

int32_t nb_rows_to_external(nb_ctx* c, const uint64_t* rows_internal, int64_t n, uint64_t* rows_out) {
  if (!c || n < 0 || (n && (!rows_internal || !rows_out)) || c->W <= 0) { set_error("nb_rows_to_external: bad arguments"); return NB_ERR_INVALID; }
  if (n) renumber_rows(c, rows_internal, n, rows_out);
  return NB_OK;
}

int32_t nb_classes_read(nb_ctx* c, int64_t cap, uint64_t* rows, int64_t* counts) {
  int64_t n = 0; const int rc = nb_classes_count(c, &n); if (rc != NB_OK) return rc;
  if (cap < n) { set_error("nb_classes_read: cap smaller than the number of classes"); return NB_ERR_INVALID; }
  if (n == 0) return NB_OK;
  if (rows) { const int r2 = read_rows_external(c, 0, n, rows); if (r2 != NB_OK) return r2; }
  if (counts) { const int r = readback(c, counts, c->ct.counts, (size_t)n * 8); if (r != NB_OK) return r; }
  return NB_OK;
}

int32_t nb_classes_read_range(nb_ctx* c, int64_t first, int64_t count, uint64_t* rows, int64_t* counts) {
  int64_t n = 0; const int rc = nb_classes_count(c, &n); if (rc != NB_OK) return rc;
  if (first < 0 || count < 0 || first + count > n) { set_error("nb_classes_read_range: range outside the known classes"); return NB_ERR_INVALID; }
  if (count == 0) return NB_OK;
  if (rows) { const int r2 = read_rows_external(c, first, count, rows); if (r2 != NB_OK) return r2; }
  if (counts) { const int r = readback(c, counts, c->ct.counts + first, (size_t)count * 8); if (r != NB_OK) return r; }
  return NB_OK;
}

int32_t nb_reasons_read(nb_ctx* c, int64_t* out) {
  if (!c || !out) { set_error("nb_reasons_read: null argument"); return NB_ERR_INVALID; }
  NB_HIP(hipSetDevice(c->device));
  return readback(c, out, c->ct.reasons, NB_NUM_REASONS * 8);
}

int32_t nb_results_reset(nb_ctx* c, void* stream) {
  if (!c) { set_error("nb_results_reset: null ctx"); return NB_ERR_INVALID; }
  NB_HIP(hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)stream; const size_t cap = (size_t)1 << c->ct.cap_log2;
  NB_HIP(hipMemsetAsync(c->ct.keys, 0, cap * 8, s)); NB_HIP(hipMemsetAsync(c->ct.ids, 0, cap * 4, s));
  NB_HIP(hipMemsetAsync(c->ct.counts, 0, (size_t)c->max_classes * 8, s));
  NB_HIP(hipMemsetAsync(c->ct.n_classes, 0, 4, s)); NB_HIP(hipMemsetAsync(c->ct.err, 0, 4, s));
  NB_HIP(hipMemsetAsync(c->ct.reasons, 0, NB_NUM_REASONS * 8, s));
  return NB_OK;
}

int32_t nb_counts_device_ptr(nb_ctx* c, void** d_counts, int64_t* capacity) {
  if (!c || !d_counts || !capacity) { set_error("nb_counts_device_ptr: null argument"); return NB_ERR_INVALID; }
  *d_counts = c->ct.counts; *capacity = c->max_classes; return NB_OK;
}

// ---- device memory, streams and copies for callers without a GPU array library (INTEGRATION.md §B in plain ctypes / C) ----
int32_t nb_device_alloc(nb_ctx* c, int64_t bytes, void** d_out) {
  if (!c || !d_out || bytes < 0) { set_error("nb_device_alloc: bad arguments"); return NB_ERR_INVALID; }
  NB_HIP(hipSetDevice(c->device));
  *d_out = nullptr;
  if (bytes) NB_HIP(hipMalloc(d_out, (size_t)bytes));
  return NB_OK;
}
int32_t nb_device_free(nb_ctx* c, void* d) {
  if (!c) { set_error("nb_device_free: null ctx"); return NB_ERR_INVALID; }
  NB_HIP(hipSetDevice(c->device));
  if (d) NB_HIP(hipFree(d));
  return NB_OK;
}
int32_t nb_host_alloc_pinned(int64_t bytes, void** out) {
  if (!out || bytes < 0) { set_error("nb_host_alloc_pinned: bad arguments"); return NB_ERR_INVALID; }
  *out = nullptr;
  if (bytes) NB_HIP(hipHostMalloc(out, (size_t)bytes, hipHostMallocDefault));
  return NB_OK;
}
int32_t nb_host_free_pinned(void* p) {
  if (p) NB_HIP(hipHostFree(p));
  return NB_OK;
}
int32_t nb_stream_create(nb_ctx* c, void** stream_out) {
  if (!c || !stream_out) { set_error("nb_stream_create: null argument"); return NB_ERR_INVALID; }
  NB_HIP(hipSetDevice(c->device));
  hipStream_t s = nullptr;
  NB_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  *stream_out = (void*)s; return NB_OK;
}
int32_t nb_stream_destroy(nb_ctx* c, void* stream) {
  if (!c) { set_error("nb_stream_destroy: null ctx"); return NB_ERR_INVALID; }
  NB_HIP(hipSetDevice(c->device));
  if (stream) NB_HIP(hipStreamDestroy((hipStream_t)stream));
  return NB_OK;
}
int32_t nb_copy_h2d_async(nb_ctx* c, void* d_dst, const void* src, int64_t bytes, void* stream) {
  if (!c || bytes < 0 || (bytes && (!d_dst || !src))) { set_error("nb_copy_h2d_async: bad arguments"); return NB_ERR_INVALID; }
  NB_HIP(hipSetDevice(c->device));
  if (bytes) NB_HIP(hipMemcpyAsync(d_dst, src, (size_t)bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
  return NB_OK;
}
int32_t nb_copy_d2h_async(nb_ctx* c, void* dst, const void* d_src, int64_t bytes, void* stream) {
  if (!c || bytes < 0 || (bytes && (!dst || !d_src))) { set_error("nb_copy_d2h_async: bad arguments"); return NB_ERR_INVALID; }
  NB_HIP(hipSetDevice(c->device));
  if (bytes) NB_HIP(hipMemcpyAsync(dst, d_src, (size_t)bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
  return NB_OK;
}

const char* nb_last_kernel_names(nb_ctx* c) { return c ? c->last_kernels.c_str() : ""; }

const char* nb_kernel_name(int32_t max_len, int32_t k, int32_t n_seqs, int32_t paired) {
  static thread_local char buf[96];
  const int positions = max_len - k + 1; const int nr = positions < 1 ? 1 : (positions + 63) / 64; const int W = (n_seqs + 63) / 64;
  const char* ct = (k <= COMPACT_MAX_K && !(std::getenv("NB_TABLE") && std::string(std::getenv("NB_TABLE")) == "wide") && n_seqs <= 2000) ? "true" : "false";   // (a guess from the shape: the real choice is by table size, nb_index.cpp; nb_last_kernel_names(ctx) is exact)
  if (paired) std::snprintf(buf, sizeof buf, "nb_align_general_kernel<%d, %d, %d, %s>", use_tri(positions, k) ? 3 : (nr > 5 ? 5 : nr), W <= 64 ? 1 : (W <= 256 ? 4 : 16), use_tri(positions, k) ? 2 : 1, ct);
  else std::snprintf(buf, sizeof buf, "nb_align_se_kernel<%d, %d, %s, %s>", use_tri(positions, k) ? 3 : (nr > 5 ? 5 : nr), W <= 64 ? 1 : (W <= 256 ? 4 : 16), use_tri(positions, k) ? "true" : "false", ct);
  return buf;
}

}  // extern "C"
