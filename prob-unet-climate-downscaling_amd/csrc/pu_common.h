// Internal shared definitions for libprobunet (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

namespace pu {

// ------------------------------------------------------------------ element types
typedef _Float16 f16;
struct bf16 { uint16_t v; };   // storage-only; converted explicitly

typedef float  f32x4  __attribute__((ext_vector_type(4)));
typedef float  f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short  s16x4 __attribute__((ext_vector_type(4)));
typedef short  s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float bf16_to_f(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even through the hardware converter (v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ uint16_t f_to_bf16(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
__device__ __forceinline__ uint32_t f2_to_bf16x2(float lo, float hi) {
  bf16x2 v; v[0] = (__bf16)lo; v[1] = (__bf16)hi;
  return __builtin_bit_cast(uint32_t, v);
}

template <typename T> struct ET;                       // element traits
template <> struct ET<float> {
  static constexpr int VEC = 4;                        // elements per 16-byte vector
  static constexpr int DT = 0;
  __device__ static __forceinline__ float ld(const float* p) { return *p; }
  __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct ET<f16> {
  static constexpr int VEC = 8;
  static constexpr int DT = 1;
  __device__ static __forceinline__ float ld(const f16* p) { return (float)*p; }
  __device__ static __forceinline__ void st(f16* p, float v) { *p = (f16)v; }
};
template <> struct ET<bf16> {
  static constexpr int VEC = 8;
  static constexpr int DT = 2;
  __device__ static __forceinline__ float ld(const bf16* p) { return bf16_to_f(p->v); }
  __device__ static __forceinline__ void st(bf16* p, float v) { p->v = f_to_bf16(v); }
};

// 16-byte vector of T as raw bits
// a NATIVE 128-bit vector (round 3; was `struct alignas(16) V16 { uint32_t w[4]; }`): arrays of the struct were split into scalars by SROA and re-joined
// with v_mov copies behind an s_waitcnt vmcnt right after the loads that filled them - loads meant to be in flight together were serialised
typedef uint32_t V16 __attribute__((ext_vector_type(4)));

template <typename T> __device__ __forceinline__ void unpack(const V16& v, float* out);   // VEC floats
template <> __device__ __forceinline__ void unpack<float>(const V16& v, float* o) {
  o[0] = __uint_as_float(v[0]); o[1] = __uint_as_float(v[1]); o[2] = __uint_as_float(v[2]); o[3] = __uint_as_float(v[3]);
}
template <> __device__ __forceinline__ void unpack<f16>(const V16& v, float* o) {
  const f16* h = reinterpret_cast<const f16*>(&v);
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = (float)h[i];
}
template <> __device__ __forceinline__ void unpack<bf16>(const V16& v, float* o) {
#pragma unroll
  for (int i = 0; i < 4; ++i) { o[2 * i] = __uint_as_float(v[i] << 16); o[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u); }
}
template <typename T> __device__ __forceinline__ V16 pack(const float* in);
template <> __device__ __forceinline__ V16 pack<float>(const float* in) {
  V16 v; v[0] = __float_as_uint(in[0]); v[1] = __float_as_uint(in[1]); v[2] = __float_as_uint(in[2]); v[3] = __float_as_uint(in[3]); return v;
}
template <> __device__ __forceinline__ V16 pack<f16>(const float* in) {
  V16 v; f16* h = reinterpret_cast<f16*>(&v);
#pragma unroll
  for (int i = 0; i < 8; ++i) h[i] = (f16)in[i];
  return v;
}
template <> __device__ __forceinline__ V16 pack<bf16>(const float* in) {
  V16 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = f2_to_bf16x2(in[2 * i], in[2 * i + 1]);
  return v;
}
__device__ __forceinline__ V16 zero16() { V16 v; v[0] = v[1] = v[2] = v[3] = 0; return v; }

// ------------------------------------------------------------------ tensor view (NHWC, strided pixel rows)
struct TV {
  void* p = nullptr;     // element (b,y,x,c) at p[((b*H + y)*W + x)*ld + c]
  int B = 0, H = 0, W = 0, C = 0, ld = 0;
  __host__ __device__ size_t pixels() const { return (size_t)B * H * W; }
};

// SiLU and its derivative.  16-bit engines: hardware exp2 / reciprocal (v_exp_f32, v_rcp_f32, ~1 ulp).
// fp32 parity engine (PRECISE): libm expf and an IEEE division, so that the path stays at the fp32 noise floor.
template <bool PRECISE> __device__ __forceinline__ float sigmoid_f(float v) {
  if (PRECISE) return 1.0f / (1.0f + expf(-v));
  return __builtin_amdgcn_rcpf(1.0f + __expf(-v));
}
template <bool PRECISE> __device__ __forceinline__ float silu_f(float v) { return v * sigmoid_f<PRECISE>(v); }
template <bool PRECISE> __device__ __forceinline__ float dsilu_f(float v) { const float s = sigmoid_f<PRECISE>(v); return s * (1.0f + v * (1.0f - s)); }

// Counter-hash RNG for dropout, regenerated (never stored) in backward.  One 32-bit hash serves TWO consecutive elements
// (16 bits each): keep iff u16 < thr16 = round(keep * 65536).  key = drop_key(seed, stream) is computed once per kernel.
__device__ __forceinline__ uint32_t hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint32_t drop_key(uint64_t seed, uint32_t stream) {
  return hash32((uint32_t)seed ^ hash32((uint32_t)(seed >> 32) + 0x9E3779B9u * (stream + 1u)));
}
__device__ __forceinline__ uint32_t drop_thr16(float keep) { return (uint32_t)(keep * 65536.0f + 0.5f); }
// 16-bit uniforms of the element pair containing element index `idx` (idx even): low half -> idx, high half -> idx + 1
__device__ __forceinline__ uint32_t drop_pair(uint32_t key, uint64_t idx) {
  return hash32((uint32_t)(idx >> 1) * 0x9E3779B1u + key + (uint32_t)(idx >> 33));
}

// hipFuncSetAttribute is per device: each launcher remembers which devices it has configured
struct AttrOnce {
  bool done[32] = {};
  bool& cur() { int d = 0; (void)hipGetDevice(&d); return done[d & 31]; }
};
// RAII: make `dev` current for the duration of an entry point and restore the caller's device afterwards
struct DeviceGuard {
  int prev = -1; bool switched = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
  }
  ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};

inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
inline int rup(int a, int b) { return (a + b - 1) / b * b; }

}  // namespace pu
