// Host stand-in for <hip/hip_runtime.h>, used ONLY by tests/native/anchor_emul.cpp: it lets g++ compile the device code of
// nimble_amd/csrc/nb_anchor.hip (and the headers it includes) as plain C++ with ONE LANE PER WAVE, so that the walk logic of the
// anchor-extend kernel runs on the CPU under AddressSanitizer / UBSan and is compared with the oracle in the not-gpu suite.
// A wave of one lane: __ballot(p) is p, readlane / shuffles return the lane's own value, atomics are plain memory operations, kernels are
// ordinary functions that the driver calls once per (block, wave) with threadIdx.x = 64 * wave (lane 0). Nothing here is product code.
#pragma once
#include <chrono>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#define __global__
#define __device__
#define __host__
#define __shared__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __HIP_MEMORY_SCOPE_AGENT 0

struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
struct uint4 { uint32_t x, y, z, w; };
extern dim3 threadIdx, blockIdx, blockDim, gridDim;

#define __syncthreads() ((void)0)
#define __ballot(p) ((p) ? 1ull : 0ull)
#define __shfl(v, l) (v)
#define __builtin_amdgcn_readfirstlane(v) (v)
#define __builtin_amdgcn_readlane(v, l) (v)
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) (src)
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_s_sleep(n) ((void)0)
#define __builtin_amdgcn_wave_barrier() ((void)0)
static inline uint32_t __brev(uint32_t x) {
  x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1); x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
  x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4); x = ((x >> 8) & 0x00FF00FFu) | ((x & 0x00FF00FFu) << 8);
  return (x >> 16) | (x << 16);
}
static inline uint64_t __brevll(uint64_t x) { return ((uint64_t)__brev((uint32_t)x) << 32) | __brev((uint32_t)(x >> 32)); }
#define __popcll(x) __builtin_popcountll(x)
static inline long long clock64() { return (long long)std::chrono::steady_clock::now().time_since_epoch().count(); }

template <class T, class V> static inline T atomicAdd(T* p, V v) { const T old = *p; *p = (T)(old + (T)v); return old; }
template <class T, class V, class W> static inline T atomicCAS(T* p, V cmp, W val) { const T old = *p; if (old == (T)cmp) *p = (T)val; return old; }
template <class T, class V> static inline T atomicOr(T* p, V v) { const T old = *p; *p = (T)(old | (T)v); return old; }
#define __hip_atomic_load(p, order, scope) (*(p))
#define __hip_atomic_store(p, v, order, scope) ((void)(*(p) = (v)))
template <class T, class V> static inline T nb_emul_fetch_add(T* p, V v) { const T old = *p; *p = (T)(old + (T)v); return old; }
#define __hip_atomic_fetch_add(p, v, order, scope) nb_emul_fetch_add((p), (v))
template <class T, class V> static inline bool nb_emul_cas(T* p, T* expected, V desired) {
  if (*p == *expected) { *p = (T)desired; return true; }
  *expected = *p; return false;
}
#define __hip_atomic_compare_exchange_strong(p, e, d, o1, o2, scope) nb_emul_cas((p), (e), (d))

// host API, as far as the launch helpers of the included sources mention it (they are compiled, never called)
typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0 };
enum { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2 };
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
static inline const char* hipGetErrorString(hipError_t) { return "emulation"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = std::malloc(n); return hipSuccess; }
static inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { std::memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { std::memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) ((void)0)
