#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
template <int CTRL> __device__ __forceinline__ uint32_t dpp_mov(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false); }
__global__ void k(const uint32_t* in, uint32_t* out) {
  uint32_t v = in[threadIdx.x];
  out[threadIdx.x] = dpp_mov<0x121>(v);
  out[64 + threadIdx.x] = dpp_mov<0x128>(v);
  uint32_t a = v; a &= dpp_mov<0x121>(a); a &= dpp_mov<0x122>(a); a &= dpp_mov<0x124>(a); a &= dpp_mov<0x128>(a);
  out[128 + threadIdx.x] = a;
  out[192 + threadIdx.x] = (uint32_t)__builtin_amdgcn_readlane((int)a, 16);
}
int main() {
  std::vector<uint32_t> h(64), o(256);
  for (int i = 0; i < 64; ++i) h[i] = 0xFFFF0000u | (1u << (i % 16)) | (i << 20 & 0xF00000);
  for (int i = 0; i < 64; ++i) h[i] = ~(1u << (i % 32));
  uint32_t *di, *dout; hipMalloc(&di, 256); hipMalloc(&dout, 1024);
  hipMemcpy(di, h.data(), 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout);
  hipMemcpy(o.data(), dout, 1024, hipMemcpyDeviceToHost);
  for (int i = 0; i < 20; ++i) printf("lane %2d in %08x ror1 %08x ror8 %08x rowand %08x rl16 %08x\n", i, h[i], o[i], o[64 + i], o[128 + i], o[192 + i]);
  return 0;
}
