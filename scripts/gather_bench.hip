// gather_bench.hip — standalone random-gather microbenchmark for MI355X: the measured ceiling that the k-mer table
// lookups of nb_kernels.hip run against (replaces the self-derived "56 G requests/s" of round 1).
//
// Every lane draws pseudo-random slot indices (counter hash, no memory dependence) and reads REQ bytes (16, 32 or 64:
// 1, 2 or 4 global_load_dwordx4 to one aligned slot) from a table of T bytes, UNROLL independent requests in flight per lane,
// at a chosen number of waves per SIMD (grid = CUs x blocks of 256 threads, launch bounds pin the register budget). Output:
// one JSON line per configuration with requests/s, useful GB/s and the byte count to calibrate FETCH_SIZE against
// (rocprofv3 --pmc FETCH_SIZE -- ./gather_bench ...: the program goes directly behind `--`).
//
// build: hipcc -O3 --offload-arch=gfx950 -o gather_bench scripts/gather_bench.hip
// usage: gather_bench [table_MB=49] [req_bytes=32] [blocks_per_cu=5] [requests_per_lane=4096] [repeats=5] [unroll]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {   // lowbias32
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x;
}

// 8-byte requests (one dwordx2 per slot): the access shape of a one-entry-per-lookup table (perfect hash)
template <int UNROLL>
__global__ __launch_bounds__(256) void gather8_kernel(const uint2* __restrict__ table, uint32_t n_slots, int iters, uint32_t seed, uint32_t* __restrict__ sink) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0, ctr = seed + tid * 0x9E3779B9u;
  for (int it = 0; it < iters; ++it) {
    uint2 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) { ctr += 0x6C8E9CF5u; v[u] = table[(uint32_t)(((uint64_t)mix(ctr) * (uint64_t)n_slots) >> 32)]; }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc ^= v[u].x ^ v[u].y;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int REQ, int UNROLL>
__global__ __launch_bounds__(256) void gather_kernel(const uint4* __restrict__ table, uint32_t n_slots, int iters, uint32_t seed, uint32_t* __restrict__ sink) {
  constexpr int Q = REQ / 16;                       // dwordx4 loads per request
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0, ctr = seed + tid * 0x9E3779B9u;
  for (int it = 0; it < iters; ++it) {
    uint4 v[UNROLL][Q];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      ctr += 0x6C8E9CF5u;
      const uint32_t slot = (uint32_t)(((uint64_t)mix(ctr) * (uint64_t)n_slots) >> 32);
      const uint4* p = table + (size_t)slot * Q;
#pragma unroll
      for (int q = 0; q < Q; ++q) v[u][q] = p[q];
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
#pragma unroll
      for (int q = 0; q < Q; ++q) acc ^= v[u][q].x ^ v[u][q].y ^ v[u][q].z ^ v[u][q].w;
  }
  if (acc == 0x12345678u) sink[0] = acc;            // keeps the loads live; practically never true
}

template <int REQ, int UNROLL>
static int run(size_t table_bytes, int blocks_per_cu, int req_per_lane, int repeats) {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const uint32_t n_slots = (uint32_t)(table_bytes / REQ);
  uint4* d_table = nullptr; uint32_t* d_sink = nullptr;
  CHECK(hipMalloc((void**)&d_table, (size_t)n_slots * REQ)); CHECK(hipMalloc((void**)&d_sink, 4));
  { std::vector<uint32_t> h((size_t)n_slots * REQ / 4); uint32_t s = 1; for (auto& x : h) { s = s * 1664525u + 1013904223u; x = s | 1u; }
    CHECK(hipMemcpy(d_table, h.data(), h.size() * 4, hipMemcpyHostToDevice)); }
  const int grid = cus * blocks_per_cu, iters = req_per_lane / UNROLL;
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  auto launch = [&](uint32_t seed) {
    if constexpr (REQ == 8) hipLaunchKernelGGL((gather8_kernel<UNROLL>), dim3(grid), dim3(256), 0, 0, (const uint2*)d_table, n_slots, iters, seed, d_sink);
    else hipLaunchKernelGGL((gather_kernel<REQ, UNROLL>), dim3(grid), dim3(256), 0, 0, d_table, n_slots, iters, seed, d_sink);
  };
  launch(1u);   // warm-up (fills the caches that can hold it)
  CHECK(hipDeviceSynchronize());
  double best = 1e30, sum = 0;
  for (int r = 0; r < repeats; ++r) {
    CHECK(hipEventRecord(a, 0));
    launch(77u + (uint32_t)r);
    CHECK(hipEventRecord(b, 0)); CHECK(hipEventSynchronize(b));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b));
    best = ms < best ? ms : best; sum += ms;
  }
  const double reqs = (double)grid * 256.0 * (double)iters * UNROLL;
  const double avg_ms = sum / repeats;
  std::printf("{\"table_MB\": %.3f, \"req_bytes\": %d, \"blocks_per_cu\": %d, \"waves_per_simd\": %d, \"in_flight_per_lane\": %d, \"requests_per_launch\": %.0f, "
              "\"useful_bytes_per_launch\": %.0f, \"avg_ms\": %.4f, \"best_ms\": %.4f, \"G_requests_per_s\": %.2f, \"useful_GB_per_s\": %.1f, \"lane_loads_G_per_s\": %.2f}\n",
              (double)n_slots * REQ / 1e6, REQ, blocks_per_cu, blocks_per_cu, UNROLL, reqs, reqs * REQ, avg_ms, best, reqs / (avg_ms * 1e-3) / 1e9,
              reqs * REQ / (avg_ms * 1e-3) / 1e9, reqs * (REQ < 16 ? 1 : REQ / 16) / (avg_ms * 1e-3) / 1e9);
  CHECK(hipFree(d_table)); CHECK(hipFree(d_sink));
  return 0;
}

int main(int argc, char** argv) {
  const double table_mb = argc > 1 ? std::atof(argv[1]) : 49.0;
  const int req = argc > 2 ? std::atoi(argv[2]) : 32;
  const int bpc = argc > 3 ? std::atoi(argv[3]) : 5;
  const int rpl = argc > 4 ? std::atoi(argv[4]) : 4096;
  const int rep = argc > 5 ? std::atoi(argv[5]) : 5;
  const size_t bytes = (size_t)(table_mb * 1e6);
  if (bpc < 1 || bpc > 8 || rpl < 8 || bytes < 4096) { std::fprintf(stderr, "bad arguments\n"); return 2; }
  const int unroll = argc > 6 ? std::atoi(argv[6]) : (req == 64 ? 4 : 8);   // independent requests in flight per lane (VGPRs = unroll x req / 4)
  switch (req * 100 + unroll) {
    case 802: return run<8, 2>(bytes, bpc, rpl, rep);
    case 804: return run<8, 4>(bytes, bpc, rpl, rep);
    case 808: return run<8, 8>(bytes, bpc, rpl, rep);
    case 1602: return run<16, 2>(bytes, bpc, rpl, rep);
    case 1604: return run<16, 4>(bytes, bpc, rpl, rep);
    case 1608: return run<16, 8>(bytes, bpc, rpl, rep);
    case 3202: return run<32, 2>(bytes, bpc, rpl, rep);
    case 3204: return run<32, 4>(bytes, bpc, rpl, rep);
    case 3208: return run<32, 8>(bytes, bpc, rpl, rep);
    case 6402: return run<64, 2>(bytes, bpc, rpl, rep);
    case 6404: return run<64, 4>(bytes, bpc, rpl, rep);
    default: std::fprintf(stderr, "req_bytes must be 8, 16, 32 or 64 and unroll 2, 4 or 8 (64 B: 2 or 4)\n"); return 2;
  }
}
