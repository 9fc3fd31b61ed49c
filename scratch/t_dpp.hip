// probe: wave-wide sum over the lanes of equal (lane & 3) with DPP row rotations + permlane16/32 swaps (no LDS)
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ float wave_sum_mod4(float x) {
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x124, 0xf, 0xf, false));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x128, 0xf, 0xf, false));
  auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, x), false, false);
  x = __builtin_bit_cast(float, r[0]) + __builtin_bit_cast(float, r[1]);
  auto q = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, x), false, false);
  x = __builtin_bit_cast(float, q[0]) + __builtin_bit_cast(float, q[1]);
  return x;
}
__device__ __forceinline__ float wave_sum_mod4_asm(float x) {
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x124, 0xf, 0xf, false));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x128, 0xf, 0xf, false));
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  x = a + b; a = x; b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}
__global__ void k(const float* in, float* o1, float* o2) { o1[threadIdx.x] = wave_sum_mod4(in[threadIdx.x]); o2[threadIdx.x] = wave_sum_mod4_asm(in[threadIdx.x]); }
int main() {
  float h[64], r1[64], r2[64], *d, *e1, *e2;
  for (int i = 0; i < 64; ++i) h[i] = (float)(1 << (i % 4)) * (1 + i / 4) + 0.001f * i;
  hipMalloc(&d, 256); hipMalloc(&e1, 256); hipMalloc(&e2, 256);
  hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, e1, e2);
  hipMemcpy(r1, e1, 256, hipMemcpyDeviceToHost); hipMemcpy(r2, e2, 256, hipMemcpyDeviceToHost);
  int bad1 = 0, bad2 = 0;
  for (int i = 0; i < 64; ++i) {
    double ref = 0; for (int j = i % 4; j < 64; j += 4) ref += h[j];
    if (fabs(r1[i] - ref) > 1e-3 * ref) ++bad1;
    if (fabs(r2[i] - ref) > 1e-3 * ref) ++bad2;
  }
  printf("builtin: %d wrong lanes, asm: %d wrong lanes (lane0 %f / %f)\n", bad1, bad2, r1[0], r2[0]);
  return 0;
}
