// nb_wave.h — wave64 reductions for gfx950 built on DPP row rotations (shared by nb_kernels.hip and tests/native).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace nb {

// Wave-wide reductions with a wave-UNIFORM result: four DPP row rotations reduce inside each row of 16 lanes (every
// lane of a row ends up with the row's value); the rows are then folded with two DPP row broadcasts (row_bcast:15 into
// rows 1 and 3, row_bcast:31 into rows 2 and 3: lane 63 ends up with the total, one v_readlane) — or, for sums, with four
// v_readlane + scalar adds. No LDS traffic (ds_bpermute) and no lgkmcnt waits. Must be called with all 64 lanes active
// (wave-uniform control flow).
template <int CTRL> __device__ __forceinline__ uint32_t dpp_mov(uint32_t v) {
  // old = the source itself: with every lane enabled `old` is never selected, but passing 0 lets hipcc's DPP-combine fold
  // it into v_min_u32_dpp with a wrong identity (observed on ROCm 7.2: wave_min returned 0)
  return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false);
}
// NOTE each DPP move is evaluated ONCE into a temporary before the combining op: writing `cond ? dpp(v) : v` makes hipcc
// re-issue the DPP move inside a divergent branch (partial EXEC), which reads stale lanes.
struct OpMin { __device__ __forceinline__ static uint32_t f(uint32_t a, uint32_t b) { return a < b ? a : b; } };
struct OpMax { __device__ __forceinline__ static uint32_t f(uint32_t a, uint32_t b) { return a > b ? a : b; } };
struct OpAnd { __device__ __forceinline__ static uint32_t f(uint32_t a, uint32_t b) { return a & b; } };
struct OpAdd { __device__ __forceinline__ static uint32_t f(uint32_t a, uint32_t b) { return a + b; } };
template <class OP> __device__ __forceinline__ uint32_t row_reduce(uint32_t v) {
  uint32_t t;
  t = dpp_mov<0x121>(v); v = OP::f(v, t);
  t = dpp_mov<0x122>(v); v = OP::f(v, t);
  t = dpp_mov<0x124>(v); v = OP::f(v, t);
  t = dpp_mov<0x128>(v); v = OP::f(v, t);
  return v;
}
__device__ __forceinline__ uint32_t rows4(uint32_t v, int k) { return (uint32_t)__builtin_amdgcn_readlane((int)v, 16 * k); }
struct OpXor { __device__ __forceinline__ static uint32_t f(uint32_t a, uint32_t b) { return a ^ b; } };
// rows folded by broadcasts; `old` = what a lane of a row that is masked out of a step receives. For the idempotent
// operations that is the lane's own value (op(v, v) = v), for xor it is 0.
template <class OP, bool IDEMPOTENT> __device__ __forceinline__ uint32_t wave_fold(uint32_t v) {
  v = row_reduce<OP>(v);
  uint32_t t;
  t = (uint32_t)__builtin_amdgcn_update_dpp(IDEMPOTENT ? (int)v : 0, (int)v, 0x142, 0xA, 0xF, false); v = OP::f(v, t);   // row_bcast:15 -> rows 1, 3
  t = (uint32_t)__builtin_amdgcn_update_dpp(IDEMPOTENT ? (int)v : 0, (int)v, 0x143, 0xC, 0xF, false); v = OP::f(v, t);   // row_bcast:31 -> rows 2, 3
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) { return wave_fold<OpMin, true>(v); }
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) { return wave_fold<OpMax, true>(v); }
__device__ __forceinline__ uint32_t wave_and_u32(uint32_t v) { return wave_fold<OpAnd, true>(v); }
__device__ __forceinline__ uint32_t wave_xor_u32(uint32_t v) { return wave_fold<OpXor, false>(v); }
__device__ __forceinline__ uint64_t wave_xor_u64(uint64_t v) {
  return ((uint64_t)wave_xor_u32((uint32_t)(v >> 32)) << 32) | wave_xor_u32((uint32_t)v);
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
  v = row_reduce<OpAdd>(v);
  return rows4(v, 0) + rows4(v, 1) + rows4(v, 2) + rows4(v, 3);
}
__device__ __forceinline__ uint64_t wave_and_u64(uint64_t v) {
  return ((uint64_t)wave_and_u32((uint32_t)(v >> 32)) << 32) | wave_and_u32((uint32_t)v);
}
__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {   // mod 2^64: sum the halves separately, carry by hand
  uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
  // low halves: 64 values < 2^32 sum to < 2^38: split into 16-bit limbs so 32-bit lane adds cannot overflow
  const uint32_t s0 = wave_sum_u32(lo & 0xFFFFu), s1 = wave_sum_u32(lo >> 16), s2 = wave_sum_u32(hi & 0xFFFFu), s3 = wave_sum_u32(hi >> 16);
  return (uint64_t)s0 + ((uint64_t)s1 << 16) + ((uint64_t)s2 << 32) + ((uint64_t)s3 << 48);
}
}  // namespace nb
