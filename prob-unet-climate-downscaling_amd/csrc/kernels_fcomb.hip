// Fused Fcomb (prob_unet.py:87-138): z-as-bias 1x1 -> ReLU -> 1x1 -> ReLU -> 1x1, forward and backward.
//   * Fcomb.tile + cat never materialise: layer 0 = W0[:, :F] . feat + (W0[:, F:] . z + b0), a per-(member, sample) bias.
//   * the W0[:, :F] . feat product is shared by all M ensemble members of a pixel.
//   * one lane per pixel, fp32 VALU math; weights are wave-uniform (scalar loads, SGPR operands).
//   * HBM-bound: reads F planes once per pixel for all M members, writes M*Cout planes (coalesced NCHW fp32).
#include "pu_kernels.h"

namespace pu {

template <typename T> __device__ __forceinline__ V16 ldv16(const T* p) { return *reinterpret_cast<const V16*>(p); }

// zb[m][b][o] = b0[o] + sum_l W0[o][F+l] z[m][b][l]
__global__ void fcomb_zbias_kernel(const float* __restrict__ z, const float* __restrict__ w0, const float* __restrict__ b0,
                                   int F, int L, int MB, float* __restrict__ zb) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= MB * F) return;
  const int o = i % F, mb = i / F;
  float acc = b0[o];
  for (int l = 0; l < L; ++l) acc += w0[(long)o * (F + L) + F + l] * z[(long)mb * L + l];
  zb[i] = acc;
}

template <typename T, int F>
__device__ __forceinline__ void load_feat(const T* p, float* f) {
  constexpr int VEC = ET<T>::VEC;
#pragma unroll
  for (int v = 0; v < F / VEC; ++v) unpack<T>(ldv16<T>(p + v * VEC), f + v * VEC);
}

template <typename T, int F>
__global__ __launch_bounds__(256) void fcomb_fwd_kernel(FcombArgs a, const float* __restrict__ zb) {
  const long HW = (long)a.feat.H * a.feat.W;
  const int b = blockIdx.y;
  const T* fp = reinterpret_cast<const T*>(a.feat.p) + (a.bcast ? 0 : (long)b * HW * a.feat.ld);
  const int WS = F + a.L;
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += (long)gridDim.x * blockDim.x) {
    float f[F], pre[F];
    load_feat<T, F>(fp + p * a.feat.ld, f);
#pragma unroll
    for (int o = 0; o < F; ++o) {
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < F; ++c) acc += a.w0[o * WS + c] * f[c];
      pre[o] = acc;
    }
    for (int m = 0; m < a.M; ++m) {
      const float* zbm = zb + ((long)m * a.B + b) * F;
      float h0[F], h1[F];
#pragma unroll
      for (int o = 0; o < F; ++o) h0[o] = fmaxf(pre[o] + zbm[o], 0.f);
#pragma unroll
      for (int o = 0; o < F; ++o) {
        float acc = a.b1[o];
#pragma unroll
        for (int c = 0; c < F; ++c) acc += a.w1[o * F + c] * h0[c];
        h1[o] = fmaxf(acc, 0.f);
      }
      for (int co = 0; co < a.Cout; ++co) {
        float acc = a.b2[co];
#pragma unroll
        for (int c = 0; c < F; ++c) acc += a.w2[co * F + c] * h1[c];
        a.out[(((long)b * a.M + m) * a.Cout + co) * HW + p] = acc;
      }
    }
  }
}

// backward: per pixel recompute h0/h1, propagate dout -> dh1 -> dh0 -> dfeat; weight gradients are accumulated per block
// through LDS outer products (thread (o, 4 c's) sums over the block's 256 pixels), flushed once per block with atomics.
template <typename T, int F>
__global__ __launch_bounds__(256) void fcomb_bwd_kernel(FcombBwdArgs a, const float* __restrict__ zb, float* __restrict__ dzb) {
  constexpr int LDF = F + 1;
  extern __shared__ __attribute__((aligned(16))) float fc_smem[];
  float* sA = fc_smem;                 // "left" vectors  [pix][F]  (dout padded / dh1 / dh0)
  float* sB = fc_smem + 256 * LDF;     // "right" vectors [pix][F]  (h1 / h0 / feat)
  const FcombArgs& f = a.f;
  const long HW = (long)f.feat.H * f.feat.W;
  const int b = blockIdx.y, tid = threadIdx.x;
  const T* fp = reinterpret_cast<const T*>(f.feat.p) + (f.bcast ? 0 : (long)b * HW * f.feat.ld);
  const int WS = F + f.L;
  constexpr int NE = F * F / 4;                 // (o, c-quad) entries of an FxF matrix
  const int eo = tid / (F / 4), ec = (tid % (F / 4)) * 4;
  float aw0[4] = {0, 0, 0, 0}, aw1[4] = {0, 0, 0, 0}, aw2[4] = {0, 0, 0, 0};
  float ab1 = 0.f, ab2 = 0.f;
  const int nchunks = (int)((HW + 255) / 256);
  for (int ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
    const long p = (long)ch * 256 + tid;
    const bool valid = p < HW;
    float ft[F], pre[F], dfe[F];
    if (valid) load_feat<T, F>(fp + p * f.feat.ld, ft);
    else {
#pragma unroll
      for (int c = 0; c < F; ++c) ft[c] = 0.f;
    }
#pragma unroll
    for (int o = 0; o < F; ++o) {
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < F; ++c) acc += f.w0[o * WS + c] * ft[c];
      pre[o] = acc; dfe[o] = 0.f;
    }
    for (int m = 0; m < f.M; ++m) {
      const float* zbm = zb + ((long)m * f.B + b) * F;
      float h0[F], h1[F], dh1[F], dh0[F];
#pragma unroll
      for (int o = 0; o < F; ++o) h0[o] = fmaxf(pre[o] + zbm[o], 0.f);
#pragma unroll
      for (int o = 0; o < F; ++o) {
        float acc = f.b1[o];
#pragma unroll
        for (int c = 0; c < F; ++c) acc += f.w1[o * F + c] * h0[c];
        h1[o] = fmaxf(acc, 0.f);
      }
#pragma unroll
      for (int c = 0; c < F; ++c) dh1[c] = 0.f;
      __syncthreads();                                     // previous phase-B readers done
      for (int co = 0; co < f.Cout; ++co) {
        const float g = valid ? a.dout[(((long)b * f.M + m) * f.Cout + co) * HW + p] : 0.f;
        if (co < F) sA[tid * LDF + co] = g;
#pragma unroll
        for (int c = 0; c < F; ++c) dh1[c] += f.w2[co * F + c] * g;
      }
#pragma unroll
      for (int c = 0; c < F; ++c) { dh1[c] = (h1[c] > 0.f && valid) ? dh1[c] : 0.f; sB[tid * LDF + c] = valid ? h1[c] : 0.f; }
      __syncthreads();
      // dW2[co][c] += sum_pix dout[co] h1[c];  db2[co] += sum_pix dout[co]
      if (tid < NE && eo < f.Cout) {
        for (int q = 0; q < 256; ++q) {
          const float l = sA[q * LDF + eo];
#pragma unroll
          for (int e = 0; e < 4; ++e) aw2[e] += l * sB[q * LDF + ec + e];
        }
      }
      if (tid >= 128 && tid - 128 < f.Cout && tid - 128 < F) { for (int q = 0; q < 256; ++q) ab2 += sA[q * LDF + tid - 128]; }
      __syncthreads();
#pragma unroll
      for (int c = 0; c < F; ++c) { sA[tid * LDF + c] = dh1[c]; sB[tid * LDF + c] = valid ? h0[c] : 0.f; }
#pragma unroll
      for (int c = 0; c < F; ++c) {
        float acc = 0.f;
#pragma unroll
        for (int o = 0; o < F; ++o) acc += f.w1[o * F + c] * dh1[o];
        dh0[c] = (h0[c] > 0.f && valid) ? acc : 0.f;
      }
      __syncthreads();
      if (tid < NE) {
        for (int q = 0; q < 256; ++q) {
          const float l = sA[q * LDF + eo];
#pragma unroll
          for (int e = 0; e < 4; ++e) aw1[e] += l * sB[q * LDF + ec + e];
        }
      }
      if (tid >= 128 && tid - 128 < F) { for (int q = 0; q < 256; ++q) ab1 += sA[q * LDF + tid - 128]; }
      __syncthreads();
#pragma unroll
      for (int c = 0; c < F; ++c) { sA[tid * LDF + c] = dh0[c]; sB[tid * LDF + c] = ft[c]; }
#pragma unroll
      for (int c = 0; c < F; ++c) {
        float acc = 0.f;
#pragma unroll
        for (int o = 0; o < F; ++o) acc += f.w0[o * WS + c] * dh0[o];
        dfe[c] += acc;
      }
      __syncthreads();
      if (tid < NE) {
        for (int q = 0; q < 256; ++q) {
          const float l = sA[q * LDF + eo];
#pragma unroll
          for (int e = 0; e < 4; ++e) aw0[e] += l * sB[q * LDF + ec + e];
        }
      }
      if (tid >= 128 && tid - 128 < F) {                   // d(zb)[m][b][o] = sum_pix dh0[o]
        float s = 0.f;
        for (int q = 0; q < 256; ++q) s += sA[q * LDF + tid - 128];
        atomicAdd(dzb + ((long)m * f.B + b) * F + (tid - 128), s);
      }
    }
    if (a.dfeat.p && valid) {
      constexpr int VEC = ET<T>::VEC;
      T* dp = reinterpret_cast<T*>(a.dfeat.p) + ((long)b * HW + p) * a.dfeat.ld;
#pragma unroll
      for (int v = 0; v < F / VEC; ++v) {
        float o[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) o[e] = dfe[v * VEC + e];
        if (a.dfeat_accumulate) {
          float old[VEC]; unpack<T>(ldv16<T>(dp + v * VEC), old);
#pragma unroll
          for (int e = 0; e < VEC; ++e) o[e] += old[e];
        }
        *reinterpret_cast<V16*>(dp + v * VEC) = pack<T>(o);
      }
    }
  }
  // flush
  if (tid < NE) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      atomicAdd(a.dw0 + (long)eo * WS + ec + e, aw0[e]);
      atomicAdd(a.dw1 + (long)eo * F + ec + e, aw1[e]);
      if (eo < f.Cout) atomicAdd(a.dw2 + (long)eo * F + ec + e, aw2[e]);
    }
  }
  if (tid >= 128 && tid - 128 < F) {
    atomicAdd(a.db1 + (tid - 128), ab1);
    if (tid - 128 < f.Cout) atomicAdd(a.db2 + (tid - 128), ab2);
  }
}

// from dzb[m][b][o]: db0[o] += sum; dW0[o][F+l] += sum_{m,b} dzb z[m][b][l]; dz[m][b][l] = sum_o W0[o][F+l] dzb[m][b][o]
__global__ void fcomb_bwd_z_kernel(const float* __restrict__ dzb, const float* __restrict__ z, const float* __restrict__ w0, int F, int L, int MB,
                                   float* dz, float* dw0, float* db0) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int WS = F + L;
  if (i < F * L) {
    const int o = i / L, l = i % L;
    float acc = 0.f;
    for (int mb = 0; mb < MB; ++mb) acc += dzb[(long)mb * F + o] * z[(long)mb * L + l];
    dw0[(long)o * WS + F + l] += acc;
  } else if (i < F * L + F) {
    const int o = i - F * L;
    float acc = 0.f;
    for (int mb = 0; mb < MB; ++mb) acc += dzb[(long)mb * F + o];
    db0[o] += acc;
  }
  if (dz) {
    for (int j = i; j < MB * L; j += gridDim.x * blockDim.x) {
      const int mb = j / L, l = j % L;
      float acc = 0.f;
      for (int o = 0; o < F; ++o) acc += w0[(long)o * WS + F + l] * dzb[(long)mb * F + o];
      dz[j] = acc;
    }
  }
}

// scratch: zb and dzb live in a small workspace the engine provides through the stream-ordered pointers below.
static float* g_fc_ws = nullptr; static long g_fc_ws_n = 0;
static hipError_t fc_ws(long n, float** out) {
  if (n > g_fc_ws_n) {
    if (g_fc_ws) { hipError_t e0 = hipFree(g_fc_ws); if (e0 != hipSuccess) return e0; }
    hipError_t e = hipMalloc(&g_fc_ws, n * sizeof(float)); if (e != hipSuccess) return e;
    g_fc_ws_n = n;
  }
  *out = g_fc_ws; return hipSuccess;
}

template <typename T>
hipError_t launch_fcomb_fwd(const FcombArgs& a, hipStream_t s) {
  float* ws; hipError_t e = fc_ws(2L * a.M * a.B * a.F + 16, &ws); if (e != hipSuccess) return e;
  float* zb = ws;
  const int MB = a.M * a.B;
  hipLaunchKernelGGL(fcomb_zbias_kernel, dim3(cdiv((long)MB * a.F, 256)), dim3(256), 0, s, a.z, a.w0, a.b0, a.F, a.L, MB, zb);
  const long HW = (long)a.feat.H * a.feat.W;
  dim3 grid((unsigned)min((long)1024, (HW + 255) / 256), a.B);
  if (a.F == 32) hipLaunchKernelGGL((fcomb_fwd_kernel<T, 32>), grid, dim3(256), 0, s, a, zb);
  else if (a.F == 16) hipLaunchKernelGGL((fcomb_fwd_kernel<T, 16>), grid, dim3(256), 0, s, a, zb);
  else if (a.F == 8) hipLaunchKernelGGL((fcomb_fwd_kernel<T, 8>), grid, dim3(256), 0, s, a, zb);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

template <typename T>
hipError_t launch_fcomb_bwd(const FcombBwdArgs& a, hipStream_t s) {
  const FcombArgs& f = a.f;
  const int MB = f.M * f.B;
  float* ws; hipError_t e = fc_ws(2L * MB * f.F + 16, &ws); if (e != hipSuccess) return e;
  float* zb = ws; float* dzb = ws + (long)MB * f.F;
  hipLaunchKernelGGL(fcomb_zbias_kernel, dim3(cdiv((long)MB * f.F, 256)), dim3(256), 0, s, f.z, f.w0, f.b0, f.F, f.L, MB, zb);
  e = hipMemsetAsync(dzb, 0, (size_t)MB * f.F * sizeof(float), s); if (e != hipSuccess) return e;
  const long HW = (long)f.feat.H * f.feat.W;
  const long nchunks = (HW + 255) / 256;
  dim3 grid((unsigned)min((long)16, nchunks), f.B);
  if (f.Cout > f.F) return hipErrorInvalidValue;
  const size_t lds = (size_t)2 * 256 * (f.F + 1) * sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(fcomb_bwd_kernel<T, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 256 * 33 * 4);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  if (f.F == 32) hipLaunchKernelGGL((fcomb_bwd_kernel<T, 32>), grid, dim3(256), lds, s, a, zb, dzb);
  else if (f.F == 16) hipLaunchKernelGGL((fcomb_bwd_kernel<T, 16>), grid, dim3(256), lds, s, a, zb, dzb);
  else if (f.F == 8) hipLaunchKernelGGL((fcomb_bwd_kernel<T, 8>), grid, dim3(256), lds, s, a, zb, dzb);
  else return hipErrorInvalidValue;
  hipLaunchKernelGGL(fcomb_bwd_z_kernel, dim3(cdiv((long)f.F * f.L + f.F, 256)), dim3(256), 0, s, dzb, f.z, f.w0, f.F, f.L, MB, a.dz, a.dw0, a.db0);
  return hipGetLastError();
}

#define PU_INST_FC(T)                                                          \
  template hipError_t launch_fcomb_fwd<T>(const FcombArgs&, hipStream_t);      \
  template hipError_t launch_fcomb_bwd<T>(const FcombBwdArgs&, hipStream_t);
PU_INST_FC(float)
PU_INST_FC(f16)
PU_INST_FC(bf16)

}  // namespace pu
