// Unit test of the DPP wave reductions (nimble_amd/csrc/nb_wave.h) against host arithmetic. Built and run by
// tests/test_native_gpu.py on the GPU box: hipcc --offload-arch=gfx950 -I nimble_amd/csrc ... && ./wave_reduce_test
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>

#include "nb_wave.h"

__global__ void k(const uint64_t* in, uint64_t* out) {
  const uint64_t v = in[threadIdx.x];
  out[0 * 64 + threadIdx.x] = nb::wave_min_u32((uint32_t)v);
  out[1 * 64 + threadIdx.x] = nb::wave_and_u64(v);
  out[2 * 64 + threadIdx.x] = nb::wave_sum_u32((uint32_t)__popcll(v));
  out[3 * 64 + threadIdx.x] = nb::wave_sum_u64(v);
  out[4 * 64 + threadIdx.x] = nb::wave_max_u32((uint32_t)(v >> 7));
  out[5 * 64 + threadIdx.x] = nb::wave_xor_u64(v);
}

int main() {
  std::mt19937_64 rng(7); int bad = 0;
  uint64_t *di, *dout;
  if (hipMalloc(&di, 512) != hipSuccess || hipMalloc(&dout, 6 * 512) != hipSuccess) { std::printf("no device\n"); return 2; }
  for (int it = 0; it < 400; ++it) {
    std::vector<uint64_t> h(64), o(6 * 64);
    for (auto& x : h) { x = rng(); if (it % 3 == 0) x |= rng() | rng(); if (it % 5 == 0) x = ~0ull; }
    if (it % 7 == 0) h[rng() % 64] = rng() & 0xFF;
    hipMemcpy(di, h.data(), 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout);
    hipMemcpy(o.data(), dout, 6 * 512, hipMemcpyDeviceToHost);
    uint32_t mn = ~0u, mx = 0, pc = 0; uint64_t an = ~0ull, sm = 0, xr = 0;
    for (auto x : h) { mn = std::min(mn, (uint32_t)x); mx = std::max(mx, (uint32_t)(x >> 7)); an &= x; sm += x; xr ^= x; pc += (uint32_t)__builtin_popcountll(x); }
    for (int l = 0; l < 64; ++l)
      if (o[l] != mn || o[64 + l] != an || o[128 + l] != pc || o[192 + l] != sm || o[256 + l] != mx || o[320 + l] != xr) ++bad;
  }
  std::printf("bad %d\n", bad);
  return bad != 0;
}
