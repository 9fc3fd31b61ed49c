// Toolchain + MFMA fragment-layout probe (scratch; not product code).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef short s8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

// D[32x32] = A[32x16] * B[16x32], A row-major [m][k], B given as Bt[n][k] (k contiguous).
__global__ void k_mfma_f16(const _Float16* A, const _Float16* Bt, float* D) {
  int l = threadIdx.x; int r = l & 31, h = l >> 5;
  h8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = A[r * 16 + 8 * h + j]; b[j] = Bt[r * 16 + 8 * h + j]; }
  f16v c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  for (int i = 0; i < 16; ++i) { int row = (i & 3) + 8 * (i >> 2) + 4 * h; D[row * 32 + r] = c[i]; }
}
__global__ void k_mfma_bf16(const uint16_t* A, const uint16_t* Bt, float* D) {
  int l = threadIdx.x; int r = l & 31, h = l >> 5;
  typedef __bf16 b8 __attribute__((ext_vector_type(8)));
  union { s8 s; b8 b; } a, b;
  for (int j = 0; j < 8; ++j) { a.s[j] = A[r * 16 + 8 * h + j]; b.s[j] = Bt[r * 16 + 8 * h + j]; }
  f16v c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.b, b.b, c, 0, 0, 0);
  for (int i = 0; i < 16; ++i) { int row = (i & 3) + 8 * (i >> 2) + 4 * h; D[row * 32 + r] = c[i]; }
}
// f32: D[32x32] = A[32x2]*B[2x32]
__global__ void k_mfma_f32(const float* A, const float* Bt, float* D) {
  int l = threadIdx.x; int r = l & 31, h = l >> 5;
  f16v c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * 2 + h], Bt[r * 2 + h], c, 0, 0, 0);
  for (int i = 0; i < 16; ++i) { int row = (i & 3) + 8 * (i >> 2) + 4 * h; D[row * 32 + r] = c[i]; }
}
// 16x16x32 f16
__global__ void k_mfma_f16_16(const _Float16* A, const _Float16* Bt, float* D) {
  int l = threadIdx.x; int r = l & 15, q = l >> 4;
  h8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = A[r * 32 + 8 * q + j]; b[j] = Bt[r * 32 + 8 * q + j]; }
  f4v c = {0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  for (int i = 0; i < 4; ++i) { int row = q * 4 + i; D[row * 16 + r] = c[i]; }
}
// transposed LDS read probe: LDS holds tile [16 rows(k)][64 cols] of u16 with value = row*256+col.
// each lane supplies address of row (4*blk + q), columns 4p..4p+3 within its 16-lane group's 16-col window
__global__ void k_trread(uint32_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t t[16 * 64];
  int l = threadIdx.x;
  for (int i = l; i < 16 * 64; i += 64) t[i] = (uint16_t)((i / 64) * 256 + (i % 64));
  __syncthreads();
  int g = l >> 4, i16 = l & 15, q = i16 >> 2, p = i16 & 3;
  // group g reads rows 0..3 (q), columns 16g + 4p .. +3
  uint32_t addr = (uint32_t)(uintptr_t)(&t[q * 64 + 16 * g + 4 * p]);
  uint64_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  out[l * 2] = (uint32_t)v; out[l * 2 + 1] = (uint32_t)(v >> 32);
}
__global__ void k_axpy(float* y, const float* x, float a, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) y[i] += a * x[i];
}
extern "C" {
int probe_axpy(float* y, const float* x, float a, int n, void* stream) {
  hipLaunchKernelGGL(k_axpy, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, y, x, a, n);
  return (int)hipGetLastError();
}
int probe_mfma(int which, const void* A, const void* Bt, float* D, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (which == 0) hipLaunchKernelGGL(k_mfma_f16, dim3(1), dim3(64), 0, s, (const _Float16*)A, (const _Float16*)Bt, D);
  if (which == 1) hipLaunchKernelGGL(k_mfma_bf16, dim3(1), dim3(64), 0, s, (const uint16_t*)A, (const uint16_t*)Bt, D);
  if (which == 2) hipLaunchKernelGGL(k_mfma_f32, dim3(1), dim3(64), 0, s, (const float*)A, (const float*)Bt, D);
  if (which == 3) hipLaunchKernelGGL(k_mfma_f16_16, dim3(1), dim3(64), 0, s, (const _Float16*)A, (const _Float16*)Bt, D);
  return (int)hipGetLastError();
}
int probe_trread(uint32_t* out, void* stream) {
  hipLaunchKernelGGL(k_trread, dim3(1), dim3(64), 0, (hipStream_t)stream, out);
  return (int)hipGetLastError();
}
}
