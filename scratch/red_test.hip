#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <random>
template <int CTRL> __device__ __forceinline__ uint32_t dpp_mov(uint32_t v) {
  // old = the source itself: with every lane enabled `old` is never selected, but passing 0 lets hipcc's DPP-combine fold
  // it into v_min_u32_dpp with a wrong identity (observed on ROCm 7.2: wave_min returned 0)
  return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false);
}
// NOTE each DPP move is evaluated ONCE into a temporary before the combining op: writing `cond ? dpp(v) : v` makes hipcc
// re-issue the DPP move inside a divergent branch (partial EXEC), which reads stale lanes.
struct OpMin { __device__ __forceinline__ static uint32_t f(uint32_t a, uint32_t b) { return a < b ? a : b; } };
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
#define NB_ROW_REDUCE(v, OP) v = row_reduce<OP>(v)
#define NB_OP_MIN OpMin
#define NB_OP_AND OpAnd
#define NB_OP_ADD OpAdd
__device__ __forceinline__ uint32_t rows4(uint32_t v, int k) { return (uint32_t)__builtin_amdgcn_readlane((int)v, 16 * k); }
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
  NB_ROW_REDUCE(v, NB_OP_MIN);
  const uint32_t a = rows4(v, 0), b = rows4(v, 1), c = rows4(v, 2), d = rows4(v, 3);
  const uint32_t ab = a < b ? a : b, cd = c < d ? c : d;
  return ab < cd ? ab : cd;
}
__device__ __forceinline__ uint32_t wave_and_u32(uint32_t v) {
  NB_ROW_REDUCE(v, NB_OP_AND);
  return rows4(v, 0) & rows4(v, 1) & rows4(v, 2) & rows4(v, 3);
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
  NB_ROW_REDUCE(v, NB_OP_ADD);
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
__global__ void k(const uint64_t* in, uint64_t* out) {
  uint64_t v = in[threadIdx.x];
  out[0 * 64 + threadIdx.x] = wave_min_u32((uint32_t)v);
  out[1 * 64 + threadIdx.x] = wave_and_u64(v);
  out[2 * 64 + threadIdx.x] = wave_sum_u32((uint32_t)__popcll(v));
  out[3 * 64 + threadIdx.x] = wave_sum_u64(v);
}
int main() {
  std::mt19937_64 rng(7); int bad = 0;
  uint64_t *di, *dout; hipMalloc(&di, 512); hipMalloc(&dout, 4 * 512);
  for (int it = 0; it < 200; ++it) {
    std::vector<uint64_t> h(64), o(256);
    for (auto& x : h) { x = rng(); if (it % 3 == 0) x |= rng() | rng(); if (it % 5 == 0) x = ~0ull; }
    if (it % 7 == 0) h[rng() % 64] = rng() & 0xFF;
    hipMemcpy(di, h.data(), 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout);
    hipMemcpy(o.data(), dout, 2048, hipMemcpyDeviceToHost);
    uint32_t mn = ~0u; uint64_t an = ~0ull, sm = 0; uint32_t pc = 0;
    for (auto x : h) { mn = std::min(mn, (uint32_t)x); an &= x; sm += x; pc += __builtin_popcountll(x); }
    for (int l = 0; l < 64; ++l) if (o[l] != mn || o[64 + l] != an || o[128 + l] != pc || o[192 + l] != sm) { if (bad < 5) printf("it %d lane %d: min %llx/%x and %llx/%llx pc %llu/%u sum %llx/%llx\n", it, l, (unsigned long long)o[l], mn, (unsigned long long)o[64+l], (unsigned long long)an, (unsigned long long)o[128+l], pc, (unsigned long long)o[192+l], (unsigned long long)sm); ++bad; }
  }
  printf("bad %d\n", bad); return bad != 0;
}
