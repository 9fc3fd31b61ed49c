// Implicit-GEMM convolution kernels for gfx950 (MFMA), NHWC activations.
//
//   conv_igemm : y[b,p,co] = sum_{tap,ci} in[b,p+tap,ci] * w[co,tap,ci]  (+bias, +residual, ReLU, accumulate)
//                used for forward (networks.py:89, prob_unet.py:41,45) and, with the (flipped, transposed) packed
//                weights, for the data gradient (aten conv backward-input).
//   conv_wgrad : dw[co,ci,tap] += sum_{b,p} dy[b,p,co] * in[b,p+tap,ci]   (aten conv backward-weight), split over
//                pixel tiles, fp32 atomics into the flat gradient buffer.
//
// GEMM orientation: D[row = cout][col = pixel] = W[cout][k] * X[k][pixel], k = (tap, cin-chunk).
//   v_mfma_f32_32x32x16_{f16,bf16}: lane l holds A[row l&31][k 8(l>>5)..+7], B[k 8(l>>5)..+7][col l&31]
//   v_mfma_f32_32x32x2_f32        : lane l holds A[row l&31][k l>>5],        B[k l>>5][col l&31]
//   D: col = l&31, row = (r&3) + 8(r>>2) + 4(l>>5)            (verified on hardware, tools/mfma_probe.hip)
// LDS tiles are [pixel][channel] / [cout][channel] with the channel (K) axis contiguous, so one ds_read_b128 yields
// a whole 16-bit fragment; the 3x3 halo tile is staged once per 32-channel chunk and reused by all 9 taps.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "pu_kernels.h"

namespace pu {

// ------------------------------------------------------------------ profiling hooks
namespace {
struct ProfRec { char name[128]; double flops, bytes; hipEvent_t e0, e1; };
bool g_prof_on = false;
std::vector<ProfRec> g_prof_log;
}  // namespace
void prof_enable(bool on) { g_prof_on = on; }
bool prof_enabled() { return g_prof_on; }
void prof_record(const char* name, double flops, double bytes, hipStream_t s, bool begin) {
  if (begin) {
    ProfRec r; snprintf(r.name, sizeof r.name, "%s", name); r.flops = flops; r.bytes = bytes;
    if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return;
    (void)hipEventRecord(r.e0, s);
    g_prof_log.push_back(r);
  } else if (!g_prof_log.empty()) {
    (void)hipEventRecord(g_prof_log.back().e1, s);
  }
}
int prof_collect(ProfEntry* out, int max_entries) {
  int n = 0;
  for (auto& r : g_prof_log) {
    float ms = 0.f;
    if (hipEventSynchronize(r.e1) == hipSuccess) (void)hipEventElapsedTime(&ms, r.e0, r.e1);
    (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1);
    int i = 0;
    for (; i < n; ++i) if (!strcmp(out[i].name, r.name)) break;
    if (i == n) { if (n >= max_entries) continue; memset(&out[n], 0, sizeof(ProfEntry)); snprintf(out[n].name, sizeof out[n].name, "%s", r.name); ++n; }
    out[i].launches += 1; out[i].ms += ms; out[i].flops += r.flops; out[i].bytes += r.bytes;
  }
  g_prof_log.clear();
  return n;
}

// ------------------------------------------------------------------ MFMA traits
template <typename T> struct MM;
template <> struct MM<f16> {
  static constexpr int KSTEP = 16;
  static constexpr int PAD = 8;                      // LDS row pad (elements): 80-byte rows -> conflict-free b128
  typedef f16x8 Frag;
  __device__ static __forceinline__ Frag ld(const f16* row, int l) { return *reinterpret_cast<const f16x8*>(row + 8 * (l >> 5)); }
  __device__ static __forceinline__ f32x16 mfma(Frag a, Frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
  __device__ static __forceinline__ f32x4 mfma16(Frag a, Frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
template <> struct MM<bf16> {
  static constexpr int KSTEP = 16;
  static constexpr int PAD = 8;
  typedef bf16x8 Frag;
  __device__ static __forceinline__ Frag ld(const bf16* row, int l) { return *reinterpret_cast<const bf16x8*>(reinterpret_cast<const uint16_t*>(row) + 8 * (l >> 5)); }
  __device__ static __forceinline__ f32x16 mfma(Frag a, Frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
  __device__ static __forceinline__ f32x4 mfma16(Frag a, Frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct MM<float> {
  static constexpr int KSTEP = 2;
  static constexpr int PAD = 1;                      // 33-dword rows -> conflict-free ds_read_b32
  typedef float Frag;
  __device__ static __forceinline__ Frag ld(const float* row, int l) { return row[l >> 5]; }
  __device__ static __forceinline__ f32x16 mfma(Frag a, Frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
};

template <typename T> __device__ __forceinline__ void lds_store_vec(T* dst, const V16& v) { *reinterpret_cast<V16*>(dst) = v; }
template <> __device__ __forceinline__ void lds_store_vec<float>(float* dst, const V16& v) {   // rows are 33 dwords: not 16B aligned
  dst[0] = __uint_as_float(v[0]); dst[1] = __uint_as_float(v[1]); dst[2] = __uint_as_float(v[2]); dst[3] = __uint_as_float(v[3]);
}

template <typename T> __device__ __forceinline__ void load4(const T* p, float* o);
template <> __device__ __forceinline__ void load4<float>(const float* p, float* o) { f32x4 v = *reinterpret_cast<const f32x4*>(p); o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3]; }
template <> __device__ __forceinline__ void load4<f16>(const f16* p, float* o) { f16x4 v = *reinterpret_cast<const f16x4*>(p); o[0] = (float)v[0]; o[1] = (float)v[1]; o[2] = (float)v[2]; o[3] = (float)v[3]; }
template <> __device__ __forceinline__ void load4<bf16>(const bf16* p, float* o) {
  uint2 v = *reinterpret_cast<const uint2*>(p);
  o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u); o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
}
template <typename T> __device__ __forceinline__ void store4(T* p, const float* o);
template <> __device__ __forceinline__ void store4<float>(float* p, const float* o) { f32x4 v = {o[0], o[1], o[2], o[3]}; *reinterpret_cast<f32x4*>(p) = v; }
template <> __device__ __forceinline__ void store4<f16>(f16* p, const float* o) { f16x4 v = {(f16)o[0], (f16)o[1], (f16)o[2], (f16)o[3]}; *reinterpret_cast<f16x4*>(p) = v; }
template <> __device__ __forceinline__ void store4<bf16>(bf16* p, const float* o) {
  uint2 v; v.x = f2_to_bf16x2(o[0], o[1]); v.y = f2_to_bf16x2(o[2], o[3]);
  *reinterpret_cast<uint2*>(p) = v;
}

constexpr int KC = 32;     // channels staged per K chunk

// 16 zero bytes in device memory: out-of-image / beyond-Cin staging vectors are LOADED from here instead of being loaded from a
// dummy address and zeroed by a select - a select next to the load makes the wave wait for the load at once (see conv3_kernel)
__device__ __attribute__((aligned(16))) unsigned int g_zero16[4] = {0u, 0u, 0u, 0u};

// ------------------------------------------------------------------ forward / dgrad implicit GEMM
// Block = 64*WM*WN threads; pixel tile TH x TW (BM = TH*TW pixels, BM/WM per wave in 32-pixel MFMA columns);
// cout tile BN = 32*NTN*WN.  K loop over 32-channel chunks: the global loads of chunk c+1 are issued into registers
// before the MFMAs of chunk c (register prefetch), the 9 taps x K-steps of a chunk are fully unrolled so that the LDS
// fragment reads of the next MFMA group overlap the current one.
template <typename T, int KS, int TH, int TW, int WM, int WN, int NTN>
__global__ __launch_bounds__(64 * WM * WN) void conv_igemm_kernel(ConvArgs a) {
  typedef MM<T> M;
  constexpr int NT = 64 * WM * WN;
  constexpr int BM = TH * TW;
  constexpr int NTM = BM / (32 * WM);
  constexpr int BN = 32 * NTN * WN;
  constexpr int TAPS = KS * KS;
  constexpr int PADP = KS / 2;
  constexpr int IH = TH + 2 * PADP, IW = TW + 2 * PADP;
  constexpr int KCP = KC + M::PAD;
  constexpr int VEC = ET<T>::VEC;
  constexpr int CV = KC / VEC;
  constexpr int NVI_TOT = IH * IW * CV, NVI = (NVI_TOT + NT - 1) / NT;
  constexpr int NVW_TOT = BN * TAPS * CV, NVW = (NVW_TOT + NT - 1) / NT;
  constexpr int KSTEPS = KC / M::KSTEP;
  static_assert(BM % (32 * WM) == 0, "pixel tile");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* sIn = reinterpret_cast<T*>(smem_raw);                 // [IH*IW][KCP]
  T* sW = sIn + IH * IW * KCP;                             // [TAPS][BN][KCP]

  const int tid = threadIdx.x;
  const int l = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave % WM, wn = wave / WM;

  const int tiles_x = a.W / TW, tiles_y = a.H / TH;
  int pt = blockIdx.x;
  const int tx0 = (pt % tiles_x) * TW; pt /= tiles_x;
  const int ty0 = (pt % tiles_y) * TH; pt /= tiles_y;
  const int b = pt;
  const int n0 = blockIdx.y * BN;

  const T* in = reinterpret_cast<const T*>(a.in);
  const T* wpk = reinterpret_cast<const T*>(a.wpk);

  f32x16 acc[NTN][NTM];
#pragma unroll
  for (int i = 0; i < NTN; ++i)
#pragma unroll
    for (int j = 0; j < NTM; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- per-thread staging plan (chunk independent): global element offsets (-1 = zero fill) and LDS offsets
  long gi[NVI]; int li[NVI]; int ci_[NVI];
#pragma unroll
  for (int k = 0; k < NVI; ++k) {
    const int i = tid + k * NT;
    gi[k] = -1; li[k] = -1; ci_[k] = 0;
    if (i < NVI_TOT) {
      const int pix = i / CV, cv = i - pix * CV;
      const int hy = pix / IW, hx = pix - hy * IW;
      const int gy = ty0 + hy - PADP, gx = tx0 + hx - PADP;
      li[k] = pix * KCP + cv * VEC; ci_[k] = cv * VEC;
      if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) gi[k] = ((long)(b * a.H + gy) * a.W + gx) * a.in_ld + cv * VEC;
    }
  }
  long gw[NVW]; int lw[NVW];
#pragma unroll
  for (int k = 0; k < NVW; ++k) {
    const int i = tid + k * NT;
    gw[k] = -1; lw[k] = -1;
    if (i < NVW_TOT) {
      const int cv = i % CV;
      const int t = (i / CV) % TAPS;
      const int n = i / (CV * TAPS);
      lw[k] = (t * BN + n) * KCP + cv * VEC;
      if (n0 + n < a.cout_pk) gw[k] = ((long)(n0 + n) * TAPS + t) * a.cin_pk + cv * VEC;
    }
  }
  V16 ri[NVI], rw[NVW];
  auto gload = [&](int c0) {
#pragma unroll
    for (int k = 0; k < NVI; ++k) {
      ri[k] = zero16();
      if (gi[k] >= 0 && c0 + ci_[k] < a.Cin) ri[k] = *reinterpret_cast<const V16*>(in + gi[k] + c0);
    }
#pragma unroll
    for (int k = 0; k < NVW; ++k) {
      rw[k] = zero16();
      if (gw[k] >= 0) rw[k] = *reinterpret_cast<const V16*>(wpk + gw[k] + c0);
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int k = 0; k < NVI; ++k) if (li[k] >= 0) lds_store_vec<T>(sIn + li[k], ri[k]);
#pragma unroll
    for (int k = 0; k < NVW; ++k) if (lw[k] >= 0) lds_store_vec<T>(sW + lw[k], rw[k]);
  };

  // per-lane LDS bases of the B (pixel) fragments, tap (0,0)
  int pbase[NTM];
#pragma unroll
  for (int j = 0; j < NTM; ++j) {
    const int m = (wm * NTM + j) * 32 + (l & 31);
    pbase[j] = ((m / TW) * IW + (m % TW)) * KCP;
  }
  const int wrow = (wn * NTN * 32 + (l & 31)) * KCP;

  gload(0);
  for (int c0 = 0; c0 < a.cin_pk; c0 += KC) {
    __syncthreads();                               // every wave finished reading the previous chunk
    lstore();
    __syncthreads();
    if (c0 + KC < a.cin_pk) gload(c0 + KC);        // in flight during the MFMAs below
    if (c0 < a.Cin) {
      // fp32 engine: two-level summation (each 32-channel chunk into a fresh accumulator, then one add into the total) keeps
      // the rounding error of the exact-fp32 MFMA chain from growing with the full K = 9 Cin
      f32x16 part[NTN][NTM];
      if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int i = 0; i < NTN; ++i)
#pragma unroll
          for (int j = 0; j < NTM; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) part[i][j][r] = 0.f;
      }
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        const int toff = ((t / KS) * IW + (t % KS)) * KCP;
        const T* wt = sW + t * BN * KCP + wrow;
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) {
          typename M::Frag fa[NTN], fb[NTM];
#pragma unroll
          for (int i = 0; i < NTN; ++i) fa[i] = M::ld(wt + i * 32 * KCP + kk * M::KSTEP, l);
#pragma unroll
          for (int j = 0; j < NTM; ++j) fb[j] = M::ld(sIn + pbase[j] + toff + kk * M::KSTEP, l);
#pragma unroll
          for (int i = 0; i < NTN; ++i)
#pragma unroll
            for (int j = 0; j < NTM; ++j) {
              if constexpr (sizeof(T) == 4) part[i][j] = M::mfma(fa[i], fb[j], part[i][j]);
              else acc[i][j] = M::mfma(fa[i], fb[j], acc[i][j]);
            }
        }
      }
      if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int i = 0; i < NTN; ++i)
#pragma unroll
          for (int j = 0; j < NTM; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] += part[i][j][r];
      }
    }
  }

  // ---- epilogue: lane owns pixel (col) l&31 of each pixel sub-tile and 4 consecutive couts per register quad
  T* out = reinterpret_cast<T*>(a.out);
  const T* res = reinterpret_cast<const T*>(a.res);
#pragma unroll
  for (int j = 0; j < NTM; ++j) {
    const int m = (wm * NTM + j) * 32 + (l & 31);
    const int gy = ty0 + m / TW, gx = tx0 + m % TW;
    const size_t pix = (size_t)(b * a.H + gy) * a.W + gx;
#pragma unroll
    for (int i = 0; i < NTN; ++i) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int co = n0 + (wn * NTN + i) * 32 + 8 * q + 4 * (l >> 5);
        if (co < a.Cout) {
          float v[4] = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
          if (a.bias) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += a.bias[co + e];
          }
          if (res) {
            float r[4]; load4<T>(res + pix * a.res_ld + co, r);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += r[e];
          }
          if (a.accumulate) {
            float r[4]; load4<T>(out + pix * a.out_ld + co, r);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += r[e];
          }
          if (a.relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
          }
          store4<T>(out + pix * a.out_ld + co, v);
        }
      }
    }
  }
}

// Sum over the 16 lanes of equal (lane & 3), delivered to every lane, without LDS traffic: two DPP row rotations inside each
// 16-lane row, then v_permlane16_swap / v_permlane32_swap across the rows.  (The ds_bpermute butterfly this replaces cost 64 LDS
// crossbar round trips per wave and tile.)  Inline asm: __builtin_amdgcn_permlane16/32_swap miscompiles on ROCm 7.2 when both
// results are used (the sum came out as vdst + vdst; scratch/t_dpp.hip is the probe), and the swap needs two wait states after a
// VALU write of its operands (s_nop 1).
__device__ __forceinline__ float wave_sum_mod4(float x) {
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x124, 0xf, 0xf, false));   // row_ror:4
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x128, 0xf, 0xf, false));   // row_ror:8
  float p = x, q = x;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(p), "+v"(q));
  x = p + q; p = x; q = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(p), "+v"(q));
  return p + q;
}
// Fused GroupNorm statistics: lane l owns the 8-channel chunk (l & 3) of every pixel it stores; after the reduction over the
// 16 lanes of equal chunk, lanes 0..3 hold the wave's (sum, sum of squares) per channel and write one 64-byte row each.
__device__ __forceinline__ void wave_stat_store(float* s1, float* s2, int l, float* dst, bool ok) {
#pragma unroll
  for (int e = 0; e < 8; ++e) { s1[e] = wave_sum_mod4(s1[e]); s2[e] = wave_sum_mod4(s2[e]); }
  if (l < 4 && ok) {
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      float4 o; o.x = s1[e]; o.y = s2[e]; o.z = s1[e + 1]; o.w = s2[e + 1];
      reinterpret_cast<float4*>(dst)[e >> 1] = o;
    }
  }
}

// host side: report the slot count of the chosen tiling, or switch the statistics off when the buffer is too small
static inline void stat_resolve(ConvArgs& a, int slots) {
  if (!a.stat_out) return;
  if (slots > a.stat_cap) { a.stat_out = nullptr; slots = 0; }
  if (a.stat_slots) *a.stat_slots = slots;
}


// host side: the data-gradient epilogue that writes dv + pass-1 rows (GNBwdFuse) exists in the conv3 / conv3p kernels only and excludes
// every other epilogue option; report the slot count, or switch the request off (the kernel then stores the plain gradient)
static inline void gnb_resolve(ConvArgs& a, int slots) {
  if (a.gnb.slots) *a.gnb.slots = 0;
  if (!a.gnb.x) return;
  if (slots > a.gnb.cap || a.bias || a.res || a.relu_mask || a.relu || a.accumulate || a.stat_out || !a.gnb.part || !a.gnb.coef || a.gnb.C != a.Cout) { a.gnb.x = nullptr; return; }
  if (a.gnb.slots) *a.gnb.slots = slots;
}

// ---- second half of the conv3 / conv3p epilogue: the wave's output tile sits in its private LDS staging tile [pixel][32 couts]
// (ERS-element rows); each lane takes (pixel, 8-cout chunk) vectors: residual / accumulate / ReLU in the 16-byte domain, 64-byte
// coalesced runs per pixel, plus either the GroupNorm statistics of the stored values (forward producers) or, for a data gradient that
// feeds a GroupNorm backward (a.gnb), the pre-activation gradient dv and its pass-1 rows (see GNBwdFuse).
// Addressing: pixbase = index of the tile's first pixel ((b * H + ty0) * W + tx0), lpix[it] = the lane's pixel offset inside the tile
// for vector `it` ((m / TW) * W + m % TW, tile-invariant: the persistent kernel computes it once), everything in 32-bit element
// offsets (pu_create rejects plans whose tensors reach 2^32 elements) - the 64-bit multiplies of the first form were a measurable
// part of an instruction-bound kernel (conv3p issues ~600 non-MFMA instructions per 36 MFMAs).
template <int NTM, int TW>
__device__ __forceinline__ void conv3_epilogue_lanes(int l, int wm, int W, unsigned* lpix) {
#pragma unroll
  for (int it = 0; it < NTM * 2; ++it) {
    const int m = wm * NTM * 32 + ((it * 64 + l) >> 2);
    lpix[it] = (unsigned)((m / TW) * W + (m % TW));
  }
}
// PRE: the residual / accumulate operands of all NIT vectors are loaded before the first use (conv3_kernel: the accumulators are staged, their
// registers free - one memory latency instead of NIT dependent ones; conv3p keeps the per-vector form, its epilogue runs inside the tile loop
// next to the resident weight fragments and 32 more registers would halve its occupancy)
template <typename T, int NTM, int TW, int TH, int WM, bool GNB, bool PRE = false>
__device__ __forceinline__ void conv3_epilogue_store(const ConvArgs& a, const T* stage, int l, int wm, int ct, int b, unsigned pixbase,
                                                     int tile_xy, int tiles_per_img, const unsigned* lpix) {
  constexpr int ERS = 40;
  constexpr int NIT = NTM * 2;
  T* out = reinterpret_cast<T*>(a.out);
  const T* res = reinterpret_cast<const T*>(a.res);
  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
  const int ch = l & 3;
  const int co = ct * 32 + ch * 8;
  const int slots = tiles_per_img * WM;
  const int slot = tile_xy * WM + wm;
  const bool okc = co < a.Cout;
  if constexpr (GNB) {
    // ---- data gradient -> dv of the preceding GroupNorm + SiLU (+ dropout), pass-1 rows
    const T* gx = reinterpret_cast<const T*>(a.gnb.x);
    V16 xr[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it)                             // every x vector in flight before the first use
      xr[it] = *reinterpret_cast<const V16*>(gx + (size_t)((pixbase + lpix[it]) * (unsigned)a.gnb.x_ld + (okc ? co : 0)));
    float cA[8], cB[8], cM[8];
    const float4* cf = reinterpret_cast<const float4*>(a.gnb.coef) + ((size_t)b * a.Cout + (okc ? co : 0));
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float4 q = cf[e]; cA[e] = q.x; cB[e] = q.y; cM[e] = q.z; }
    const float dp = a.gnb.drop.drop_p, keep = 1.f - dp, inv_keep = dp > 0.f ? 1.f / keep : 1.f;
    const uint32_t dkey = drop_key(a.gnb.drop.seed, a.gnb.drop.stream), dthr = drop_thr16(keep);
    const uint64_t HW = (uint64_t)a.H * a.W;
    // keep decisions: saved by the forward (one byte per vector; the usual case), else re-derived (hash stream / injected mask).
    // All loaded / computed before the arithmetic so that the loop below is one straight-line block per vector.
    uint32_t kbs[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) kbs[it] = 0xffu;
    if (dp > 0.f && okc) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const uint64_t base = ((uint64_t)a.gnb.drop.b0 * HW + (uint64_t)(pixbase + lpix[it])) * (uint64_t)a.Cout + (uint64_t)co;
        kbs[it] = a.gnb.drop.bits ? (uint32_t)a.gnb.drop.bits[base >> 3] : drop_keep_bits8(a.gnb.drop.mask, dkey, dthr, base);
      }
    }
    if (okc) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int pw = (it * 64 + l) >> 2;
        float v[8], xv[8];
        unpack<T>(*reinterpret_cast<const V16*>(stage + pw * ERS + ch * 8), v);
        unpack<T>(xr[it], xv);
        const uint32_t kb = kbs[it];
        float d[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          d[e] = xv[e] - cM[e];
          const float dh = ((kb >> e) & 1u) ? v[e] * inv_keep : 0.f;
          v[e] = dh * dsilu_f<false>(cA[e] * d[e] + cB[e]);
        }
        const V16 pk = pack<T>(v);
        *reinterpret_cast<V16*>(out + (size_t)((pixbase + lpix[it]) * (unsigned)a.out_ld + co)) = pk;
        float w[8]; unpack<T>(pk, w);                            // sums of the values as stored (what pass 2 reads)
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1[e] += w[e]; s2[e] += w[e] * d[e]; }
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) s2[e] *= cf[e].w;                // S2 = rstd * sum dv (x - mean): rstd is constant per (sample, channel)
    wave_stat_store(s1, s2, l, a.gnb.part + (((size_t)b * slots + slot) * a.Cout + co) * 2, okc);
    return;
  }
  const T* mk = reinterpret_cast<const T*>(a.relu_mask);
  if (mk && !res && !a.accumulate && !a.relu && !a.stat_out) {
    // ---- data gradient of a conv -> ReLU chain: the ReLU backward of the layer below (its saved output is the mask), every mask vector in
    //      flight before the first use
    if (okc) {
      V16 mr[NIT];
#pragma unroll
      for (int it = 0; it < NIT; ++it) mr[it] = *reinterpret_cast<const V16*>(mk + (size_t)((pixbase + lpix[it]) * (unsigned)a.relu_mask_ld + co));
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int pw = (it * 64 + l) >> 2;
        float v[8], m[8];
        unpack<T>(*reinterpret_cast<const V16*>(stage + pw * ERS + ch * 8), v);
        unpack<T>(mr[it], m);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = m[e] > 0.f ? v[e] : 0.f;
        *reinterpret_cast<V16*>(out + (size_t)((pixbase + lpix[it]) * (unsigned)a.out_ld + co)) = pack<T>(v);
      }
    }
    return;
  }
  const bool mod = res != nullptr || a.accumulate || a.relu || mk != nullptr;      // (wave-uniform)
  if (okc) {
    typedef int ep_i32x4 __attribute__((ext_vector_type(4)));      // whole 128-bit values (a {w[4]} struct is split into scalars and re-joined
    ep_i32x4 pre_r[PRE ? NIT : 1], pre_a[PRE ? NIT : 1];           // with copies behind an s_waitcnt right after the loads)
    if constexpr (PRE) {
      if (res) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) pre_r[it] = *reinterpret_cast<const ep_i32x4*>(res + (size_t)((pixbase + lpix[it]) * (unsigned)a.res_ld + co));
      }
      if (a.accumulate) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) pre_a[it] = *reinterpret_cast<const ep_i32x4*>(out + (size_t)((pixbase + lpix[it]) * (unsigned)a.out_ld + co));
      }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int pw = (it * 64 + l) >> 2;                          // (pixel of the wave, 8-cout chunk ch)
      const unsigned pix = pixbase + lpix[it];
      T* op = out + (size_t)(pix * (unsigned)a.out_ld + co);
      const V16 raw = *reinterpret_cast<const V16*>(stage + pw * ERS + ch * 8);
      float v[8];
      unpack<T>(raw, v);
      V16 pk = raw;
      if (mod) {
        if (res) {
          float r[8];
          if constexpr (PRE) unpack<T>(__builtin_bit_cast(V16, pre_r[it]), r);
          else unpack<T>(*reinterpret_cast<const V16*>(res + (size_t)(pix * (unsigned)a.res_ld + co)), r);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += r[e];
        }
        if (a.accumulate) {
          float r[8];
          if constexpr (PRE) unpack<T>(__builtin_bit_cast(V16, pre_a[it]), r);
          else unpack<T>(*reinterpret_cast<const V16*>(op), r);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += r[e];
        }
        if (a.relu) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
        }
        if (mk) {
          float m[8]; unpack<T>(*reinterpret_cast<const V16*>(mk + (size_t)(pix * (unsigned)a.relu_mask_ld + co)), m);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = m[e] > 0.f ? v[e] : 0.f;
        }
        pk = pack<T>(v);
        if (a.stat_out) unpack<T>(pk, v);                         // statistics of the values as stored (rounded to T)
      }
      *reinterpret_cast<V16*>(op) = pk;
      if (a.stat_out) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1[e] += v[e]; s2[e] += v[e] * v[e]; }
      }
    }
  }
  if (a.stat_out) wave_stat_store(s1, s2, l, a.stat_out + (((size_t)b * slots + slot) * a.Cout + co) * 2, okc);
}

// ------------------------------------------------------------------ conv3: cout-split waves, weights straight to registers
// 16-bit only.  Block = 4 waves; wave (wm, wn) owns NTM 32-pixel columns x ONE 32-cout row tile, so its A (weight)
// fragments are private: they are read from the fragment-major packed buffer with one fully coalesced 1-KiB load per
// (tap, k-step) directly into VGPRs (no LDS, no duplicate staging), and the registers of chunk c are refilled with
// chunk c+1 right after their last MFMA (prefetch distance = one whole chunk).  Only the 3x3 input halo tile goes
// through LDS: double-buffered, register-prefetched, one barrier per 32-channel chunk.
// MS = MFMA shape: 32 -> v_mfma_f32_32x32x16 (A fragment = 32 couts x 16 k), 16 -> v_mfma_f32_16x16x32 (16 couts x 32 k; same FLOPs per
// cycle, but the chip holds a higher clock on it under load, MI355X_MICROARCH.md "DVFS give-back" item 7).  The MS = 16 form reads
// [pixel][32 ch] LDS rows of 96 bytes (conflict-free for the lane -> (pixel = l & 15, 8 channels at 8 (l >> 4)) fragment) and
// weights packed as [cout tile][chunk][tap][cout half][lane][8] (PackDesc mode bit 2).
// GNB: the epilogue of a data gradient that feeds a GroupNorm backward (GNBwdFuse) - its own instantiation, so that the register
// allocation of the plain kernel is untouched
template <typename T, int KS, int TH, int TW, int WM, int WN, int MS = 32, bool GNB = false>
__global__ __launch_bounds__(64 * WM * WN, (MS == 16 && TH * TW / WM <= 128) ? 2 : 1) void conv3_kernel(ConvArgs a) {
  typedef MM<T> M;
  constexpr int NT = 64 * WM * WN;
  constexpr int BM = TH * TW;
  constexpr int NTM = BM / (32 * WM);
  constexpr int TAPS = KS * KS, PADP = KS / 2;
  constexpr int IH = TH + 2 * PADP, IW = TW + 2 * PADP;
  constexpr int KCP = KC + (MS == 16 ? 16 : M::PAD);
  constexpr int CV = KC / 8;
  constexpr int NVI_TOT = IH * IW * CV, NVI = (NVI_TOT + NT - 1) / NT;
  constexpr int BUF = IH * IW * KCP;
  static_assert(sizeof(T) == 2, "conv3 layout");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* sIn = reinterpret_cast<T*>(smem_raw);                 // [2][IH*IW][KCP]

  const int tid = threadIdx.x, l = tid & 63, wave = tid >> 6;
  const int wm = wave % WM, wn = wave / WM;
  const int tiles_x = a.W / TW, tiles_y = a.H / TH;
  int pt = blockIdx.x;
  if ((gridDim.x & 7) == 0) pt = (pt & 7) * (gridDim.x >> 3) + (pt >> 3);   // XCD-aware: each XCD (L2) gets a contiguous band of
                                                                             // pixel tiles, so halo rows are re-read from its own L2
  const int tx0 = (pt % tiles_x) * TW; pt /= tiles_x;
  const int ty0 = (pt % tiles_y) * TH; pt /= tiles_y;
  const int b = pt;
  const int ct = blockIdx.y * WN + wn;                     // 32-cout tile of this wave
  const bool have_w = ct * 32 < a.cout_pk;
  const int nch = a.cin_pk / KC;
  const T* in = reinterpret_cast<const T*>(a.in);
  const T* wfrag = reinterpret_cast<const T*>(a.wpk) + ((size_t)ct * nch) * TAPS * 2 * 512 + l * 8;

  constexpr int NPG = 2 * NTM;                             // MS = 16: 16-pixel groups per wave
  f32x16 acc[MS == 32 ? NTM : 1];
  f32x4 acc4[MS == 16 ? 2 : 1][MS == 16 ? NPG : 1];        // [cout half][pixel group]
  if constexpr (MS == 32) {
#pragma unroll
    for (int j = 0; j < NTM; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  } else {
    // the accumulators start from the bias (fetched here, under the latency of the first halo loads) instead of adding it in the epilogue
    // behind eight dependent loads: D of v_mfma_f32_16x16x32 holds rows (couts) 4 (l >> 4) + r of column (pixel) l & 15
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f32x4 b4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = ct * 32 + h * 16 + 4 * (l >> 4) + r;
        b4[r] = (a.bias && co < a.Cout) ? a.bias[co] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < NPG; ++j) acc4[h][j] = b4;
    }
  }

  int gi[NVI], li[NVI];
#pragma unroll
  for (int k = 0; k < NVI; ++k) {
    const int i = tid + k * NT;
    gi[k] = -1; li[k] = -1;
    if (i < NVI_TOT) {
      const int pix = i / CV, cv = i - pix * CV;
      const int hy = pix / IW, hx = pix - hy * IW;
      const int gy = ty0 + hy - PADP, gx = tx0 + hx - PADP;
      li[k] = pix * KCP + cv * 8;
      if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) gi[k] = (int)(((long)(b * a.H + gy) * a.W + gx) * a.in_ld + cv * 8);
    }
  }
  V16 ri[NVI];
  // branch-free staging loads: out-of-image / beyond-Cin vectors are loaded from a 16-byte zero page (g_zero16), so the whole chunk
  // body stays one basic block and NOTHING post-processes a loaded vector.  (Round 1 loaded a dummy address and zeroed it with a
  // select next to the load: every chunk then started with s_waitcnt vmcnt(0) - the halo loads AND the 18 weight-fragment loads in
  // flight drained before the first MFMA of the chunk.)
  const ptrdiff_t zoff = reinterpret_cast<const T*>(g_zero16) - in;
  auto gload = [&](int c0) {
#pragma unroll
    for (int k = 0; k < NVI; ++k) {
      const bool ok = gi[k] >= 0 && c0 + (li[k] % KCP) < a.Cin;
      const V16 v = *reinterpret_cast<const V16*>(in + (ok ? (ptrdiff_t)((size_t)(unsigned)gi[k] + c0) : zoff));
      ri[k] = v;
    }
  };
  auto lstore = [&](int buf, int) {
#pragma unroll
    for (int k = 0; k < NVI; ++k) if (li[k] >= 0) *reinterpret_cast<V16*>(sIn + buf * BUF + li[k]) = ri[k];
  };
  typename M::Frag fa[TAPS][2];
  const T* wbase = have_w ? wfrag : reinterpret_cast<const T*>(a.wpk) + l * 8;     // idle waves read tile 0 (never stored)
  auto wload = [&](int c, int t, int kk) -> typename M::Frag {
    union { V16 v; typename M::Frag f; } u;
    u.v = *reinterpret_cast<const V16*>(wbase + ((size_t)(c * TAPS + t) * 2 + kk) * 512);
    return u.f;
  };
  int pbase[MS == 32 ? NTM : NPG];
  if constexpr (MS == 32) {
#pragma unroll
    for (int j = 0; j < NTM; ++j) {
      const int m = (wm * NTM + j) * 32 + (l & 31);
      pbase[j] = ((m / TW) * IW + (m % TW)) * KCP + 8 * (l >> 5);
    }
  } else {
#pragma unroll
    for (int j = 0; j < NPG; ++j) {
      const int m = wm * NTM * 32 + j * 16 + (l & 15);
      pbase[j] = ((m / TW) * IW + (m % TW)) * KCP + 8 * (l >> 4);
    }
  }

  gload(0);
  // the prologue issues the weight-fragment loads in EXACTLY the order the loop re-issues them (sched_barrier pins it): the
  // compiler merges the vmcnt distances of the loop's two entry paths conservatively, and a reordered prologue turned one of the
  // loop's waits into a near-drain of the load queue (vmcnt(9) where 21 loads may stay in flight) on every chunk
#pragma unroll
  for (int t = 0; t < TAPS; ++t) {
    fa[t][0] = wload(0, t, 0); __builtin_amdgcn_sched_barrier(0);
    fa[t][1] = wload(0, t, 1); __builtin_amdgcn_sched_barrier(0);
  }
  lstore(0, 0);
  __syncthreads();
  int cur = 0;
  for (int c = 0; c < nch; ++c) {
    const bool more = c + 1 < nch;
    const int cn = more ? c + 1 : c;               // the last chunk harmlessly re-reads its own fragments
    gload(cn * KC);                                // unconditional (the last chunk re-reads itself): the chunk body stays ONE basic
                                                   // block, so the compiler's s_waitcnt vmcnt counts are exact across the back-edge
    const T* sb = sIn + cur * BUF;
    // software-pipelined over the 2*TAPS (tap, k-step) groups: the NTM LDS fragment reads of group g+1 are issued between
    // the MFMAs of group g (one ds_read_b128 per MFMA slot), so no MFMA waits on a read issued right before it
    if constexpr (MS == 16) {
      // groups = (tap, half of the wave's 16-pixel groups): NTM fragment reads of group g+1 under the 2 * NTM MFMAs (cout halves
      // x pixel groups, 16 cycles each) of group g; the two weight fragments of a tap are refilled after its second group
      typename M::Frag fb[2][NTM];
#pragma unroll
      for (int j = 0; j < NTM; ++j) fb[0][j] = *reinterpret_cast<const typename M::Frag*>(sb + pbase[j]);
#pragma unroll
      for (int g = 0; g < 2 * TAPS; ++g) {
        const int t = g >> 1, ph = g & 1;
        if (g + 1 < 2 * TAPS) {
          const int t1 = (g + 1) >> 1, ph1 = (g + 1) & 1;
          const int toff1 = ((t1 / KS) * IW + (t1 % KS)) * KCP;
#pragma unroll
          for (int j = 0; j < NTM; ++j) fb[(g + 1) & 1][j] = *reinterpret_cast<const typename M::Frag*>(sb + pbase[ph1 * NTM + j] + toff1);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int j = 0; j < NTM; ++j) acc4[h][ph * NTM + j] = M::mfma16(fa[t][h], fb[g & 1][j], acc4[h][ph * NTM + j]);
        if (ph == 1) { fa[t][0] = wload(cn, t, 0); fa[t][1] = wload(cn, t, 1); }
        // issue order inside the group: one LDS read of group g+1 behind each of the first NTM MFMAs (a wave issues in order - a block
        // of reads between two blocks of MFMAs is ~8 idle pipe cycles per read whenever the SIMD's other wave is not multiplying),
        // the rest of the MFMAs, then the weight refills
        if (g + 1 < 2 * TAPS) {
#pragma unroll
          for (int j = 0; j < NTM; ++j) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
          __builtin_amdgcn_sched_group_barrier(0x008, NTM, 0);
        } else {
          __builtin_amdgcn_sched_group_barrier(0x008, 2 * NTM, 0);
        }
        if (ph == 1) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
    typename M::Frag fb[2][NTM];
#pragma unroll
    for (int j = 0; j < NTM; ++j) fb[0][j] = *reinterpret_cast<const typename M::Frag*>(sb + pbase[j]);
#pragma unroll
    for (int g = 0; g < 2 * TAPS; ++g) {
      const int t = g >> 1, kk = g & 1;
      if (g + 1 < 2 * TAPS) {
        const int t1 = (g + 1) >> 1, kk1 = (g + 1) & 1;
        const int toff1 = ((t1 / KS) * IW + (t1 % KS)) * KCP + kk1 * 16;
#pragma unroll
        for (int j = 0; j < NTM; ++j) fb[(g + 1) & 1][j] = *reinterpret_cast<const typename M::Frag*>(sb + pbase[j] + toff1);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NTM; ++j) acc[j] = M::mfma(fa[t][kk], fb[g & 1][j], acc[j]);
      fa[t][kk] = wload(cn, t, kk);
      __builtin_amdgcn_sched_barrier(0);           // keep [reads of g+1 | MFMAs of g | weight refill] as issued: the reads land
                                                   // under the 8 MFMAs instead of being sunk next to their consumers
    }
    }
    if (more) lstore(cur ^ 1, (c + 1) * KC);
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue: D[row = cout][col = pixel] -> (+bias) -> wave-private LDS tile [pixel][32 couts] (80-byte rows) ->
  //      16-byte rows per lane: residual / accumulate / ReLU in the 16-byte domain, fully coalesced 64-byte runs per pixel
  constexpr int ERS = 40;                                   // staged row stride (elements)
  T* stage = reinterpret_cast<T*>(smem_raw) + wave * (NTM * 32 * ERS);
  if constexpr (MS == 16) {
    // D of v_mfma_f32_16x16x32: lane holds rows (couts) 4 (l >> 4) + r, r = 0..3, of column (pixel) l & 15
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int j = 0; j < NPG; ++j) {
        float v[4] = {acc4[h][j][0], acc4[h][j][1], acc4[h][j][2], acc4[h][j][3]};
        store4<T>(stage + (j * 16 + (l & 15)) * ERS + h * 16 + 4 * (l >> 4), v);
      }
  } else {
    float bq[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
      bq[r] = (a.bias && co < a.Cout) ? a.bias[co] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < NTM; ++j) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v[4] = {acc[j][4 * q] + bq[4 * q], acc[j][4 * q + 1] + bq[4 * q + 1], acc[j][4 * q + 2] + bq[4 * q + 2], acc[j][4 * q + 3] + bq[4 * q + 3]};
        store4<T>(stage + (j * 32 + (l & 31)) * ERS + 8 * q + 4 * (l >> 5), v);
      }
    }
  }
  unsigned lpix[NTM * 2];
  conv3_epilogue_lanes<NTM, TW>(l, wm, a.W, lpix);
  conv3_epilogue_store<T, NTM, TW, TH, WM, GNB, true>(a, stage, l, wm, ct, b, (unsigned)((b * a.H + ty0) * a.W + tx0), (ty0 / TH) * tiles_x + tx0 / TW,
                                                      tiles_x * tiles_y, lpix);
}

// ------------------------------------------------------------------ conv3p: persistent variant for K <= 64 input channels
// Layers with one or two 32-channel chunks (levels 0-1 forward, every data gradient whose K = Cout <= 64) are HBM-bound and
// too short per tile to hide their own load -> MFMA -> store latency at one block per CU.  Here a block keeps ALL its weight
// fragments in registers (NCH x taps x 2 fragments, loaded once), walks over many pixel tiles, and software-pipelines them:
// the halo tile of tile t+1 is in flight (global -> registers) during the MFMAs of tile t and lands in the other LDS buffer
// afterwards; the epilogue stores of tile t drain under the MFMAs of tile t+1.  One barrier per tile.
// PF = tiles whose halo loads are kept in flight per block (register sets): 1, or 2 (the 32 -> 32 configuration, whose two resident blocks per
// CU otherwise keep only ~43 KB of loads in flight per CU - below what hides the loaded HBM latency at 6 TB/s)
template <typename T, int KS, int TH, int TW, int WM, int WN, int NCH, bool GNB = false, int PF = 1>
__global__ __launch_bounds__(64 * WM * WN, ((GNB || PF == 2) && NCH == 1 && WN == 1) ? 2 : 1) void conv3p_kernel(ConvArgs a) {
  typedef MM<T> M;
  constexpr int NT = 64 * WM * WN;
  constexpr int BM = TH * TW;
  constexpr int NTM = BM / (32 * WM);
  constexpr int TAPS = KS * KS, PADP = KS / 2;
  constexpr int IH = TH + 2 * PADP, IW = TW + 2 * PADP;
  constexpr int KCP = KC + M::PAD;
  constexpr int CV = KC / 8;
  constexpr int NVI_TOT = IH * IW * CV, NVI = (NVI_TOT + NT - 1) / NT;
  constexpr int CHB = IH * IW * KCP;                         // elements of one chunk's halo tile
  constexpr int BUF = NCH * CHB;
  constexpr int ERS = 40;
  static_assert(sizeof(T) == 2, "conv3p layout");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* sIn = reinterpret_cast<T*>(smem_raw);                   // [2][NCH][IH*IW][KCP]
  const int tid = threadIdx.x, l = tid & 63, wave = tid >> 6;
  const int wm = wave % WM, wn = wave / WM;
  T* stage = sIn + 2 * BUF + wave * (NTM * 32 * ERS);        // wave-private epilogue staging
  const int tiles_x = a.W / TW, tiles_y = a.H / TH;
  const int ntiles = tiles_x * tiles_y * a.B;
  const int ct = blockIdx.y * WN + wn;
  const bool have_w = ct * 32 < a.cout_pk;
  const T* in = reinterpret_cast<const T*>(a.in);
  const T* wbase = reinterpret_cast<const T*>(a.wpk) + (have_w ? ((size_t)ct * NCH) * TAPS * 2 * 512 : (size_t)0) + l * 8;

  typename M::Frag fa[NCH][TAPS][2];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        union { V16 v; typename M::Frag f; } u;
        u.v = *reinterpret_cast<const V16*>(wbase + ((size_t)(c * TAPS + t) * 2 + kk) * 512);
        fa[c][t][kk] = u.f;
      }
  float bq[GNB ? 1 : 16];                                     // (a data gradient has no bias)
  if constexpr (!GNB) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
      bq[r] = (a.bias && co < a.Cout) ? a.bias[co] : 0.f;
    }
  }
  int pbase[NTM];
#pragma unroll
  for (int j = 0; j < NTM; ++j) {
    const int m = (wm * NTM + j) * 32 + (l & 31);
    pbase[j] = ((m / TW) * IW + (m % TW)) * KCP + 8 * (l >> 5);
  }

  V16 ri[NCH][NVI], rj[PF == 2 ? NCH : 1][PF == 2 ? NVI : 1];
  const ptrdiff_t zoff = reinterpret_cast<const T*>(g_zero16) - in;
  // tile-invariant staging plan of this thread: element offset of halo vector k relative to the tile's first pixel, and which
  // image edges it lies beyond when the tile touches them (bit 0 top, 1 bottom, 2 left, 3 right; bit 4 = slot unused)
  int hoff[NVI]; unsigned hedge = 0;
#pragma unroll
  for (int k = 0; k < NVI; ++k) {
    const int i = tid + k * NT;
    const int ii = i < NVI_TOT ? i : 0;
    const int pix = ii / CV, cv = ii - pix * CV;
    const int hy = pix / IW, hx = pix - hy * IW;
    hoff[k] = ((hy - PADP) * a.W + (hx - PADP)) * a.in_ld + cv * 8;
    unsigned e = (hy < PADP ? 1u : 0u) | (hy >= TH + PADP ? 2u : 0u) | (hx < PADP ? 4u : 0u) | (hx >= TW + PADP ? 8u : 0u) | (i < NVI_TOT ? 0u : 16u);
    hedge |= e << (5 * k);
  }
  static_assert(NVI * 5 <= 32, "edge bits of the staging plan");
  bool chan_ok[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) chan_ok[c] = c * KC + ((tid % CV) * 8) < a.Cin;      // (NT is a multiple of CV: the channel group of a thread is fixed)
  auto gload_into = [&](int tile, auto& ri) {
    int pt = tile;
    const int tx0 = (pt % tiles_x) * TW; pt /= tiles_x;
    const int ty0 = (pt % tiles_y) * TH; pt /= tiles_y;
    const int b = pt;
    const unsigned ibase = (unsigned)((b * a.H + ty0) * a.W + tx0) * (unsigned)a.in_ld;
    const unsigned omask = (ty0 == 0 ? 1u : 0u) | (ty0 + TH >= a.H ? 2u : 0u) | (tx0 == 0 ? 4u : 0u) | (tx0 + TW >= a.W ? 8u : 0u) | 16u;
#pragma unroll
    for (int k = 0; k < NVI; ++k) {
      const bool inimg = ((hedge >> (5 * k)) & omask) == 0;
      const unsigned off = ibase + (unsigned)hoff[k];
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        // (element offset of the zero page relative to `in`: one offset select per vector, no post-processing: the load stays in flight)
        const V16 v = *reinterpret_cast<const V16*>(in + ((inimg && chan_ok[c]) ? (ptrdiff_t)(off + (unsigned)(c * KC)) : zoff));
        ri[c][k] = v;
      }
    }
  };
  auto lstore_from = [&](int buf, const auto& ri) {
#pragma unroll
    for (int k = 0; k < NVI; ++k) {
      const int i = tid + k * NT;
      if (i < NVI_TOT) {
        const int pix = i / CV, cv = i - pix * CV;
#pragma unroll
        for (int c = 0; c < NCH; ++c) *reinterpret_cast<V16*>(sIn + buf * BUF + c * CHB + pix * KCP + cv * 8) = ri[c][k];
      }
    }
  };
  auto gload = [&](int tile) { gload_into(tile, ri); };
  auto lstore = [&](int buf) { lstore_from(buf, ri); };

  // XCD-aware walk: block b handles tiles b, b + G, ...; neighbouring blocks work on neighbouring tiles at the same time
  const int tiles_img = tiles_x * tiles_y;
  unsigned lpix[NTM * 2];
  conv3_epilogue_lanes<NTM, TW>(l, wm, a.W, lpix);
  // the MFMAs + epilogue of one tile out of LDS buffer `cur`
  auto compute = [&](int tile, int cur) {
    f32x16 acc[NTM];
#pragma unroll
    for (int j = 0; j < NTM; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = GNB ? 0.f : bq[GNB ? 0 : r];
    // software-pipelined over the NCH * TAPS * 2 (chunk, tap, k-step) groups: the LDS fragment reads of group g + 2 are issued before
    // the MFMAs of group g (sched_barrier keeps the blocks apart).  Left to the compiler the loop came out as read -> lgkmcnt(0) ->
    // MFMA, 36 times per tile: every MFMA waited out a full LDS round trip, which - not HBM - bounded this kernel.
    {
      constexpr int NG = NCH * TAPS * 2, LA = 2;               // groups, look-ahead
      typename M::Frag fb[LA + 1][NTM];
      const T* sb0 = sIn + cur * BUF;
      auto rd = [&](int g, int slot) {
        const int c = g / (TAPS * 2), t = (g >> 1) % TAPS, kk = g & 1;
        const int off = c * CHB + ((t / KS) * IW + (t % KS)) * KCP + kk * 16;
#pragma unroll
        for (int j = 0; j < NTM; ++j) fb[slot][j] = *reinterpret_cast<const typename M::Frag*>(sb0 + pbase[j] + off);
      };
#pragma unroll
      for (int g = 0; g < LA && g < NG; ++g) rd(g, g);
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        if (g + LA < NG) rd(g + LA, (g + LA) % (LA + 1));
        __builtin_amdgcn_sched_barrier(0);
        const int c = g / (TAPS * 2), t = (g >> 1) % TAPS, kk = g & 1;
#pragma unroll
        for (int j = 0; j < NTM; ++j) acc[j] = M::mfma(fa[c][t][kk], fb[g % (LA + 1)][j], acc[j]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // ---- epilogue of this tile (wave-private staging, no block barrier needed)
    const int txy = tile % tiles_img, b = tile / tiles_img;
    const int tx0 = (txy % tiles_x) * TW, ty0 = (txy / tiles_x) * TH;
#pragma unroll
    for (int j = 0; j < NTM; ++j) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v[4] = {acc[j][4 * q], acc[j][4 * q + 1], acc[j][4 * q + 2], acc[j][4 * q + 3]};
        store4<T>(stage + (j * 32 + (l & 31)) * ERS + 8 * q + 4 * (l >> 5), v);
      }
    }
    conv3_epilogue_store<T, NTM, TW, TH, WM, GNB>(a, stage, l, wm, ct, b, (unsigned)((b * a.H + ty0) * a.W + tx0), txy, tiles_img, lpix);
  };
  const int G = gridDim.x;
  int tile = blockIdx.x;
  int cur = 0;
  if constexpr (PF == 2) {
    // two register sets: while tile t is multiplied, the halo tiles of t + G and t + 2 G (or t + 2 G and t + 3 G) are in flight
    if (tile < ntiles) { gload_into(tile, ri); lstore_from(0, ri); }
    if (tile + G < ntiles) gload_into(tile + G, ri);
    if (tile + 2 * G < ntiles) gload_into(tile + 2 * G, rj);
    __syncthreads();
    for (; tile < ntiles; tile += 2 * G) {
      compute(tile, cur);
      if (tile + G < ntiles) lstore_from(cur ^ 1, ri);
      if (tile + 3 * G < ntiles) gload_into(tile + 3 * G, ri);
      __syncthreads();
      cur ^= 1;
      if (tile + G >= ntiles) break;
      compute(tile + G, cur);
      if (tile + 2 * G < ntiles) lstore_from(cur ^ 1, rj);
      if (tile + 4 * G < ntiles) gload_into(tile + 4 * G, rj);
      __syncthreads();
      cur ^= 1;
    }
  } else {
    if (tile < ntiles) { gload(tile); lstore(0); }
    __syncthreads();
    for (; tile < ntiles; tile += G) {
      const int nxt = tile + G;
      if (nxt < ntiles) gload(nxt);
      compute(tile, cur);
      if (nxt < ntiles) lstore(cur ^ 1);
      __syncthreads();
      cur ^= 1;
    }
  }
}

template <typename T, int KS, int TH, int TW, int WM, int WN, int NCH>
static hipError_t launch_conv3p_cfg(const ConvArgs& a0, hipStream_t s) {
  ConvArgs a = a0;
  stat_resolve(a, (a.W / TW) * (a.H / TH) * WM);
  if (KS != 3 || WN != 1) a.gnb.x = nullptr;                  // the dv epilogue: 3x3 kernels, one cout tile per block (<= 32 planes) - a wave of
                                                              // the cout-split forms walks 8-16 vectors per tile, whose prefetch registers
                                                              // cost more than the fused pass saves (254 vs 80 us measured at 64 planes)
  gnb_resolve(a, (a.W / TW) * (a.H / TH) * WM);
  constexpr int TAPS = KS * KS, PADP = KS / 2;
  constexpr int IH = TH + 2 * PADP, IW = TW + 2 * PADP, KCP = KC + MM<T>::PAD, BN = 32 * WN;
  constexpr size_t lds = ((size_t)2 * NCH * IH * IW * KCP + (size_t)(TH * TW / (32 * WM)) * 32 * 40 * WM * WN) * sizeof(T);
  static_assert(lds <= 160 * 1024, "conv3p LDS");
  void (*kern)(ConvArgs) = conv3p_kernel<T, KS, TH, TW, WM, WN, NCH>;
  int kidx = 0;
  if constexpr (KS == 3 && WN == 1) { if (a.gnb.x) { kern = conv3p_kernel<T, KS, TH, TW, WM, WN, NCH, true>; kidx = 1; } }
  if constexpr (KS == 3 && WN == 1 && NCH == 1) {
    static const int pf = getenv("PU_CONV3P_PF") ? atoi(getenv("PU_CONV3P_PF")) : 1;
    if (pf == 2 && !a.gnb.x) { kern = conv3p_kernel<T, KS, TH, TW, WM, WN, NCH, false, 2>; kidx = 2; }
  }
  static AttrOnce attr_once[3];
  if (!attr_once[kidx].cur()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_once[kidx].cur() = true;
  }
  const int ntiles = (a.W / TW) * (a.H / TH) * a.B;
  const int gy = cdiv(a.Cout, BN);
  const int per_cu = lds <= 76 * 1024 ? 2 : 1;                // resident blocks per CU by LDS
  int gx = cdiv(256 * per_cu, gy);
  if (gx > ntiles) gx = ntiles;
  char tag[128];
  const bool prof = prof_enabled();
  if (prof) {
    snprintf(tag, sizeof tag, "conv3p_kernel<%s,%d,%d,%d,%d,%d,%d%s>", ET<T>::DT == 1 ? "f16" : "bf16", KS, TH, TW, WM, WN, NCH, a.gnb.x ? ",1" : "");
    const double px = (double)a.B * a.H * a.W;
    prof_record(tag, 2.0 * px * a.Cout * (double)a.Cin * TAPS, px * (a.Cin + a.Cout) * sizeof(T) + (double)a.Cout * a.Cin * TAPS * sizeof(T), s, true);
  }
  hipLaunchKernelGGL(kern, dim3(gx, gy), dim3(64 * WM * WN), lds, s, a);
  if (prof) prof_record(tag, 0, 0, s, false);
  return hipGetLastError();
}
// persistent variant when the whole K fits one or two chunks; returns hipErrorNotSupported when no configuration applies
template <typename T, int KS, int TH, int TW>
static hipError_t launch_conv3p(const ConvArgs& a, hipStream_t s) {
  const int nch = a.cin_pk / KC;
  if (a.Cout > 64) {
    if (nch == 1) return launch_conv3p_cfg<T, KS, TH, TW, 1, 4, 1>(a, s);
    return hipErrorNotSupported;                              // 2 chunks + 80 KB staging exceed the LDS
  }
  if (a.Cout > 32) {
    // 64 -> 64 class, PU_C64_MODE=3: ONE 8-wave block per CU (4 pixel quarters x 2 cout tiles), every weight fragment of the layer
    // resident in registers (144 per lane), 2 waves per SIMD: no weight traffic from L2 after the prologue
    static const int c64_mode = getenv("PU_C64_MODE") ? atoi(getenv("PU_C64_MODE")) : 1;
    if (nch == 2 && c64_mode == 3) return launch_conv3p_cfg<T, KS, TH, TW, 4, 2, 2>(a, s);
    return nch == 1 ? launch_conv3p_cfg<T, KS, TH, TW, 2, 2, 1>(a, s) : launch_conv3p_cfg<T, KS, TH, TW, 2, 2, 2>(a, s);
  }
  return nch == 1 ? launch_conv3p_cfg<T, KS, TH, TW, 4, 1, 1>(a, s) : launch_conv3p_cfg<T, KS, TH, TW, 4, 1, 2>(a, s);
}

template <typename T, int KS, int TH, int TW, int WM, int WN, int MS = 32>
static hipError_t launch_conv3_cfg(const ConvArgs& a0, hipStream_t s) {
  if constexpr (MS == 32) {
    if (a0.mfma16) return launch_conv3_cfg<T, KS, TH, TW, WM, WN, 16>(a0, s);      // weights of this convolution are packed for 16x16x32
  }
  ConvArgs a = a0;
  stat_resolve(a, (a.W / TW) * (a.H / TH) * WM);
  if (KS != 3 || WN != 1) a.gnb.x = nullptr;                  // (see launch_conv3p_cfg)
  gnb_resolve(a, (a.W / TW) * (a.H / TH) * WM);
  constexpr int TAPS = KS * KS, PADP = KS / 2;
  constexpr int IH = TH + 2 * PADP, IW = TW + 2 * PADP, KCP = KC + (MS == 16 ? 16 : MM<T>::PAD), BN = 32 * WN;
  (void)TAPS;
  constexpr size_t lds_in = (size_t)2 * IH * IW * KCP * sizeof(T);
  constexpr size_t lds_ep = (size_t)(TH * TW / (32 * WM)) * 32 * 40 * WM * WN * sizeof(T);      // epilogue staging, wave-private
  constexpr size_t lds = lds_in > lds_ep ? lds_in : lds_ep;
  void (*kern)(ConvArgs) = conv3_kernel<T, KS, TH, TW, WM, WN, MS>;
  if constexpr (KS == 3 && WN == 1) { if (a.gnb.x) kern = conv3_kernel<T, KS, TH, TW, WM, WN, MS, true>; }
  static AttrOnce attr_once[2];
  if (!attr_once[a.gnb.x ? 1 : 0].cur()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_once[a.gnb.x ? 1 : 0].cur() = true;
  }
  dim3 grid((unsigned)((a.W / TW) * (a.H / TH) * a.B), (unsigned)cdiv(a.Cout, BN));
  char tag[128];
  const bool prof = prof_enabled();
  if (prof) {
    snprintf(tag, sizeof tag, "conv3_kernel<%s,%d,%d,%d,%d,%d,%d%s>", ET<T>::DT == 1 ? "f16" : "bf16", KS, TH, TW, WM, WN, MS, a.gnb.x ? ",1" : "");
    const double px = (double)a.B * a.H * a.W;
    prof_record(tag, 2.0 * px * a.Cout * (double)a.Cin * TAPS, px * (a.Cin + a.Cout) * sizeof(T) + (double)a.Cout * a.Cin * TAPS * sizeof(T), s, true);
  }
  hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), lds, s, a);
  if (prof) prof_record(tag, 0, 0, s, false);
  return hipGetLastError();
}
template <typename T, int KS>
static hipError_t launch_conv3(const ConvArgs& a, hipStream_t s) {
  if constexpr (sizeof(T) == 2) {
    const bool persistent = conv3_goes_persistent(a.cin_pk, a.Cout, a.H, a.W);
    if (a.mfma16 && persistent) return hipErrorInvalidValue;      // 16x16x32 fragments are only read by conv3_kernel
    if (persistent) {                             // short-K layers: persistent, weight-resident variant
      hipError_t e = hipErrorNotSupported;
      if (a.W % 32 == 0 && a.H % 8 == 0) e = launch_conv3p<T, KS, 8, 32>(a, s);
      else if (a.W % 16 == 0 && a.H % 16 == 0) e = launch_conv3p<T, KS, 16, 16>(a, s);
      return e == hipErrorNotSupported ? hipErrorInvalidValue : e;
    }
    // 64 -> 64-class layers: the non-persistent kernel keeps several blocks per CU resident and beats the 148 KB-LDS persistent
    // configuration by ~7 %; on 256-pixel tiles (one weight-fragment load per 4 MFMAs instead of per 2) another +0.8 % per step
    if (a.cin_pk == 2 * KC && a.Cout > 32 && a.Cout <= 64 && a.W % 32 == 0) {
      static const int c64_mode = getenv("PU_C64_MODE") ? atoi(getenv("PU_C64_MODE")) : 1;
      if (a.H % 8 == 0 && c64_mode == 1) return launch_conv3_cfg<T, KS, 8, 32, 2, 2>(a, s);
      if (a.H % 4 == 0 && c64_mode == 0) return launch_conv3_cfg<T, KS, 4, 32, 2, 2>(a, s);
    }
    // 128-pixel tiles for the wide-cout configuration: ~200 registers -> two co-resident blocks per CU hide each other's
    // prologue / epilogue / barrier stalls (256 -> 256 @32x32: 39.5 -> 35.8 us, 128 -> 128 @64x64: 48 -> 39 us)
    if (a.W % 32 == 0 && a.H % 4 == 0 && a.Cout > 64) return launch_conv3_cfg<T, KS, 4, 32, 1, 4>(a, s);
    if (a.W % 32 == 0 && a.H % 8 == 0) {
      if (a.Cout > 64) return launch_conv3_cfg<T, KS, 8, 32, 1, 4>(a, s);
      if (a.Cout > 32) return launch_conv3_cfg<T, KS, 8, 32, 2, 2>(a, s);
      return launch_conv3_cfg<T, KS, 8, 32, 4, 1>(a, s);
    }
    if (a.W % 16 == 0 && a.H % 8 == 0 && a.Cout > 64 &&
        (long)(a.W / 16) * (a.H / 16) * a.B * cdiv(a.Cout, 128) < 200)      // grid would leave CUs idle: 128-pixel tiles
      return launch_conv3_cfg<T, KS, 8, 16, 1, 4>(a, s);
    if (a.W % 16 == 0 && a.H % 16 == 0) {
      if (a.Cout > 64) return launch_conv3_cfg<T, KS, 16, 16, 1, 4>(a, s);
      if (a.Cout > 32) return launch_conv3_cfg<T, KS, 16, 16, 2, 2>(a, s);
      return launch_conv3_cfg<T, KS, 16, 16, 4, 1>(a, s);
    }
  }
  return hipErrorInvalidValue;
}

template <typename T, int KS, int TH, int TW, int WM, int WN, int NTN>
static hipError_t launch_cfg(const ConvArgs& a, hipStream_t s) {
  constexpr int TAPS = KS * KS, PADP = KS / 2;
  constexpr int IH = TH + 2 * PADP, IW = TW + 2 * PADP, KCP = KC + MM<T>::PAD, BN = 32 * NTN * WN;
  constexpr size_t lds = (size_t)(IH * IW + TAPS * BN) * KCP * sizeof(T);
  auto kern = conv_igemm_kernel<T, KS, TH, TW, WM, WN, NTN>;
  static AttrOnce attr_once;
  if (!attr_once.cur()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_once.cur() = true;
  }
  dim3 grid((unsigned)((a.W / TW) * (a.H / TH) * a.B), (unsigned)cdiv(a.Cout, BN));
  char tag[128];
  const bool prof = prof_enabled();
  if (prof) {
    snprintf(tag, sizeof tag, "conv_igemm_kernel<%s,%d,%d,%d,%d,%d,%d>", sizeof(T) == 4 ? "f32" : (ET<T>::DT == 1 ? "f16" : "bf16"), KS, TH, TW, WM, WN, NTN);
    const double px = (double)a.B * a.H * a.W;
    prof_record(tag, 2.0 * px * a.Cout * (double)a.Cin * TAPS, px * (a.Cin + a.Cout) * sizeof(T) + (double)a.Cout * a.Cin * TAPS * sizeof(T), s, true);
  }
  hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), lds, s, a);
  if (prof) prof_record(tag, 0, 0, s, false);
  return hipGetLastError();
}

template <typename T, int KS>
static hipError_t launch_ks(const ConvArgs& a, hipStream_t s) {
  const bool narrow = a.Cout <= 32;
  const bool wide = sizeof(T) == 2 && a.Cout >= 128;
  if (a.W % 32 == 0 && a.H % 8 == 0) {
    if constexpr (sizeof(T) == 2) { if (wide) return launch_cfg<T, KS, 8, 32, 4, 1, 4>(a, s); }
    return narrow ? launch_cfg<T, KS, 8, 32, 4, 1, 1>(a, s) : launch_cfg<T, KS, 8, 32, 4, 1, 2>(a, s);
  }
  if (a.W % 16 == 0 && a.H % 16 == 0) {
    if constexpr (sizeof(T) == 2) { if (wide) return launch_cfg<T, KS, 16, 16, 4, 1, 4>(a, s); }
    return narrow ? launch_cfg<T, KS, 16, 16, 4, 1, 1>(a, s) : launch_cfg<T, KS, 16, 16, 4, 1, 2>(a, s);
  }
  if (a.W % 8 == 0 && a.H % 8 == 0)
    return launch_cfg<T, KS, 8, 8, 1, 2, 1>(a, s);
  return hipErrorInvalidValue;
}

template <typename T>
hipError_t launch_conv(const ConvArgs& a, hipStream_t s) {
  if (a.stat_slots) *a.stat_slots = 0;             // only the conv3 / conv3p launchers produce fused statistics
  if (a.gnb.slots) *a.gnb.slots = 0;               // ... and the dv epilogue of the GroupNorm backward
  if (a.frag_layout) {
    if (!conv_uses_frag_layout((int)sizeof(T), a.H, a.W)) return hipErrorInvalidValue;
    if (a.taps == 9) return launch_conv3<T, 3>(a, s);
    if (a.taps == 1) return launch_conv3<T, 1>(a, s);
    return hipErrorInvalidValue;
  }
  if (a.relu_mask) return hipErrorInvalidValue;     // only the conv3 / conv3p epilogue applies a ReLU mask (conv_uses_frag_layout())
  if (a.taps == 9) return launch_ks<T, 3>(a, s);
  if (a.taps == 1) return launch_ks<T, 1>(a, s);
  return hipErrorInvalidValue;
}
template hipError_t launch_conv<float>(const ConvArgs&, hipStream_t);
template hipError_t launch_conv<f16>(const ConvArgs&, hipStream_t);
template hipError_t launch_conv<bf16>(const ConvArgs&, hipStream_t);

// ------------------------------------------------------------------ weight gradient
// D[row = cout][col = cin] per tap, K = pixels. Operands are read from [pixel][channel] LDS tiles:
//   16-bit: ds_read_b64_tr_b16 (hardware transposed read: lane i of a 16-lane group receives column i of a
//           4-row x 16-column block; verified in tools/mfma_probe.hip), two reads per 8-element fragment.
//   fp32  : plain ds_read_b32 (the 32x32x2 fragment is one element per lane).
template <typename T> struct WG;
template <> struct WG<float> {
  static constexpr int DPAD = 1, APAD = 1;
};
template <> struct WG<f16> { static constexpr int DPAD = 32, APAD = 0; };   // row strides 192 B / 64 B: == 64 (mod 256)
template <> struct WG<bf16> { static constexpr int DPAD = 32, APAD = 0; };

__device__ __forceinline__ uint64_t ds_read_tr16(const void* lds_ptr) {
  uint64_t v;
  const uint32_t addr = (uint32_t)(uintptr_t)lds_ptr;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  return v;
}

constexpr int WG_BCO = 64, WG_BCI = 32;

template <typename T, int KS, int TH, int TW>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs a) {
  typedef MM<T> M;
  constexpr int TAPS = KS * KS, PADP = KS / 2;
  constexpr int BM = TH * TW;
  constexpr int IH = TH + 2 * PADP, IW = TW + 2 * PADP;
  constexpr int DS = WG_BCO + WG<T>::DPAD;     // dy tile row stride (elements)
  constexpr int AS = WG_BCI + WG<T>::APAD;     // input tile row stride
  constexpr int VEC = ET<T>::VEC;
  constexpr int NTILE = 2 * TAPS;              // (cout sub-tile, tap) MFMA tiles per block
  constexpr int NJ = (NTILE + 3) / 4;          // per wave

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* sDy = reinterpret_cast<T*>(smem_raw);     // [BM][DS]
  T* sA = sDy + BM * DS;                       // [IH*IW][AS]

  const int tid = threadIdx.x, l = tid & 63, wave = tid >> 6;
  const int co0 = blockIdx.y * WG_BCO, ci0 = blockIdx.z * WG_BCI;
  const int tiles_x = a.W / TW, tiles_y = a.H / TH;
  const int ntiles = tiles_x * tiles_y * a.B;
  const T* dy = reinterpret_cast<const T*>(a.dy);
  const T* in = reinterpret_cast<const T*>(a.in);

  f32x16 acc[NJ], part[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[j][r] = 0.f; part[j][r] = 0.f; }

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int pt = tile;
    const int tx0 = (pt % tiles_x) * TW; pt /= tiles_x;
    const int ty0 = (pt % tiles_y) * TH; pt /= tiles_y;
    const int b = pt;
    __syncthreads();
    for (int i = tid; i < BM * (WG_BCO / VEC); i += 256) {
      const int pix = i / (WG_BCO / VEC), cv = i % (WG_BCO / VEC);
      const int gy = ty0 + pix / TW, gx = tx0 + pix % TW;
      const int co = co0 + cv * VEC;
      V16 v = zero16();
      if (co < a.Cout) v = *reinterpret_cast<const V16*>(dy + ((size_t)(b * a.H + gy) * a.W + gx) * a.dy_ld + co);
      lds_store_vec<T>(sDy + pix * DS + cv * VEC, v);
    }
    for (int i = tid; i < IH * IW * (WG_BCI / VEC); i += 256) {
      const int pix = i / (WG_BCI / VEC), cv = i % (WG_BCI / VEC);
      const int hy = pix / IW, hx = pix % IW;
      const int gy = ty0 + hy - PADP, gx = tx0 + hx - PADP;
      const int ci = ci0 + cv * VEC;
      V16 v = zero16();
      if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W && ci < a.Cin)
        v = *reinterpret_cast<const V16*>(in + ((size_t)(b * a.H + gy) * a.W + gx) * a.in_ld + ci);
      lds_store_vec<T>(sA + pix * AS + cv * VEC, v);
    }
    __syncthreads();

    for (int kk = 0; kk < BM / M::KSTEP; ++kk) {
      if constexpr (sizeof(T) == 2) {
        // lane roles for the transposed reads
        const int g = l >> 4, h = l >> 5, q = (l & 15) >> 2, p = l & 3;
        const int cb = 16 * (g & 1) + 4 * p;
        const int m0 = kk * 16 + 8 * h + q, m1 = m0 + 4;           // pixel rows supplied by this lane (two reads)
        union { uint64_t u[2]; typename M::Frag f; } fa[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          fa[ct].u[0] = ds_read_tr16(sDy + m0 * DS + ct * 32 + cb);
          fa[ct].u[1] = ds_read_tr16(sDy + m1 * DS + ct * 32 + cb);
        }
        const int h0 = (m0 / TW) * IW + (m0 % TW), h1 = (m1 / TW) * IW + (m1 % TW);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int tl = wave + 4 * j;
          if (tl < NTILE) {
            const int ct = tl & 1, t = tl >> 1;
            const int toff = (t / KS) * IW + (t % KS);
            union { uint64_t u[2]; typename M::Frag f; } fb;
            fb.u[0] = ds_read_tr16(sA + (h0 + toff) * AS + cb);
            fb.u[1] = ds_read_tr16(sA + (h1 + toff) * AS + cb);
            acc[j] = M::mfma(ct ? fa[1].f : fa[0].f, fb.f, acc[j]);
          }
        }
      } else {
        const int m = kk * 2 + (l >> 5);
        const int hh = (m / TW) * IW + (m % TW);
        const float a0 = reinterpret_cast<const float*>(sDy)[m * DS + (l & 31)];
        const float a1 = reinterpret_cast<const float*>(sDy)[m * DS + 32 + (l & 31)];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int tl = wave + 4 * j;
          if (tl < NTILE) {
            const int ct = tl & 1, t = tl >> 1;
            const int toff = (t / KS) * IW + (t % KS);
            const float bv = reinterpret_cast<const float*>(sA)[(hh + toff) * AS + (l & 31)];
            part[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ct ? a1 : a0, bv, part[j], 0, 0, 0);
          }
        }
      }
    }
    if constexpr (sizeof(T) == 4) {           // two-level summation: one pixel tile per partial accumulator
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[j][r] += part[j][r]; part[j][r] = 0.f; }
    }
  }
  // ---- flush: atomics into the fp32 gradient in the reference layout [Cout][Cin][KS][KS]
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int tl = wave + 4 * j;
    if (tl < NTILE) {
      const int ct = tl & 1, t = tl >> 1;
      const int ci = ci0 + (l & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        if (co < a.Cout && ci < a.Cin) atomicAdd(a.dw + ((size_t)co * a.Cin + ci) * TAPS + t, acc[j][r] * a.inv_scale);
      }
    }
  }
}

template <typename T, int KS, int TH, int TW>
static hipError_t launch_wg(const WgradArgs& a, hipStream_t s) {
  constexpr int PADP = KS / 2, BM = TH * TW, IH = TH + 2 * PADP, IW = TW + 2 * PADP;
  constexpr size_t lds = ((size_t)BM * (WG_BCO + WG<T>::DPAD) + (size_t)IH * IW * (WG_BCI + WG<T>::APAD)) * sizeof(T);
  auto kern = conv_wgrad_kernel<T, KS, TH, TW>;
  static AttrOnce attr_once;
  if (!attr_once.cur()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_once.cur() = true;
  }
  const int ntiles = (a.W / TW) * (a.H / TH) * a.B;
  const int gy = cdiv(a.Cout, WG_BCO), gz = cdiv(a.Cin, WG_BCI);
  int split = cdiv(1024, gy * gz);                       // ~4 blocks per CU
  if (split > ntiles) split = ntiles;
  if (split < 1) split = 1;
  char tag[128];
  const bool prof = prof_enabled();
  if (prof) {
    snprintf(tag, sizeof tag, "conv_wgrad_kernel<%s,%d,%d,%d>", sizeof(T) == 4 ? "f32" : (ET<T>::DT == 1 ? "f16" : "bf16"), KS, TH, TW);
    const double px = (double)a.B * a.H * a.W;
    prof_record(tag, 2.0 * px * a.Cout * (double)a.Cin * a.taps, px * (a.Cin + a.Cout) * sizeof(T) + (double)a.Cout * a.Cin * a.taps * 4, s, true);
  }
  hipLaunchKernelGGL(kern, dim3(split, gy, gz), dim3(256), lds, s, a);
  if (prof) prof_record(tag, 0, 0, s, false);
  return hipGetLastError();
}

template <typename T, int KS>
static hipError_t launch_wg_ks(const WgradArgs& a, hipStream_t s) {
  if (a.W % 32 == 0 && a.H % 4 == 0) return launch_wg<T, KS, 4, 32>(a, s);
  if (a.W % 16 == 0 && a.H % 8 == 0) return launch_wg<T, KS, 8, 16>(a, s);
  if (a.W % 8 == 0 && a.H % 8 == 0) return launch_wg<T, KS, 8, 8>(a, s);
  return hipErrorInvalidValue;
}

template <typename T>
hipError_t launch_wgrad(const WgradArgs& a, hipStream_t s) {
  if constexpr (sizeof(T) == 2) {
    return launch_wgrad16<T>(a, s);                 // kernels_wgrad.hip
  } else {
    if (a.taps == 9) return launch_wg_ks<T, 3>(a, s);
    if (a.taps == 1) return launch_wg_ks<T, 1>(a, s);
    return hipErrorInvalidValue;
  }
}
template hipError_t launch_wgrad<float>(const WgradArgs&, hipStream_t);
template hipError_t launch_wgrad<f16>(const WgradArgs&, hipStream_t);
template hipError_t launch_wgrad<bf16>(const WgradArgs&, hipStream_t);

// ------------------------------------------------------------------ weight packing (fp32 master -> T, K-contiguous)
// mode 0 (forward):  dst[(co*taps + t)*kpk + ci]          = w[co][ci][t]
// mode 1 (dgrad)  :  dst[(ci*taps + t)*kpk + co]          = w[co][ci][taps-1-t]     (roles swapped, taps flipped)
template <typename T>
__global__ void pack_weights_kernel(const float* __restrict__ params, T* __restrict__ packed, const PackDesc* __restrict__ descs) {
  const PackDesc d = descs[blockIdx.y];
  if (sizeof(T) == 2 && d.mode >= 2) return;               // fragment-major descriptors: pack_frag_kernel
  const long total = (long)d.rows_pk * d.taps * d.k_pk;
  const float* w = params + d.src_off;
  T* dst = packed + d.dst_off;
  const bool frag = d.mode >= 2, dgrad = (d.mode & 1) != 0;
  const int nch = d.k_pk / 32;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int k, t, r;
    if (!frag) {
      k = (int)(i % d.k_pk); t = (int)((i / d.k_pk) % d.taps); r = (int)(i / ((long)d.k_pk * d.taps));
    } else {
      const int e = (int)(i & 7), lane = (int)((i >> 3) & 63);
      long q = i >> 9;
      const int kk = (int)(q & 1); q >>= 1;
      t = (int)(q % d.taps); q /= d.taps;
      const int c = (int)(q % nch); const int ct = (int)(q / nch);
      r = ct * 32 + (lane & 31); k = c * 32 + kk * 16 + 8 * (lane >> 5) + e;
    }
    float v = 0.f;
    if (!dgrad) {
      if (r < d.Cout && k < d.Cin) v = w[((size_t)r * d.Cin + k) * d.taps + t];
    } else {
      if (r < d.Cin && k < d.Cout) v = w[((size_t)k * d.Cin + r) * d.taps + (d.taps - 1 - t)];
    }
    ET<T>::st(dst + i, v);
  }
}

// Fragment-major descriptors (modes 2 / 3) through LDS: a block takes one (row tile, 32-channel chunk) = 32 x 32 x taps weights.
// Their fp32 sources are 32 contiguous runs of 32 * taps floats (forward: one run per cout over its cin range; data gradient:
// one run per cout over the cin ROW range), read coalesced into LDS rows of 32 * taps + 1 floats (conflict-free re-read), then
// written as 16-byte fragment vectors.  The element-wise kernel above gathers with a 36-byte stride and runs at a third of this.
template <typename T>
__global__ __launch_bounds__(256) void pack_frag_kernel(const float* __restrict__ params, T* __restrict__ packed, const PackDesc* __restrict__ descs) {
  static_assert(sizeof(T) == 2, "fragment layouts exist for the 16-bit engines only");
  __shared__ float sm[32][32 * 9 + 1];
  const PackDesc d = descs[blockIdx.y];
  if (d.mode < 2) return;
  const bool dgrad = (d.mode & 1) != 0, l16 = (d.mode & 4) != 0;     // l16: fragments of v_mfma_f32_16x16x32 ([cout half][lane][8])
  const int taps = d.taps, run = 32 * taps, nch = d.k_pk / 32, nct = d.rows_pk / 32;
  const float* w = params + d.src_off;
  const int tid = threadIdx.x;
  for (int tile = blockIdx.x; tile < nct * nch; tile += gridDim.x) {
    const int ct = tile / nch, c = tile - ct * nch;
    __syncthreads();                                       // previous tile's reads are done
    // run q (0..31) = cout index within the tile's cout range; elements = (cin offset within the tile's cin range) * taps + tap
    const int cbase = (dgrad ? ct : c) * 32, rbase = (dgrad ? c : ct) * 32;
    const bool fast = (d.src_off & 3) == 0 && (d.Cin & 3) == 0 && cbase + 32 <= d.Cin && rbase + 32 <= d.Cout &&
                      (reinterpret_cast<uintptr_t>(params) & 15) == 0;           // whole 32 x 32 tile, 16-byte aligned rows
    if (fast) {
      // 16-byte loads: a row is run / 4 float4 (the 4-byte form issued 36 scalar loads per thread and ran at 1.7 TB/s of traffic)
      const int run4 = run >> 2;                             // 72 (3x3) or 8 (1x1) vectors per row
      constexpr int U4 = 9;
      f32x4 tv[U4];
#pragma unroll
      for (int u = 0; u < U4; ++u) {
        const int i = u * 256 + tid;
        const int q = i / run4, o4 = i - q * run4;
        tv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (i < 32 * run4) tv[u] = *reinterpret_cast<const f32x4*>(w + ((size_t)(rbase + q) * d.Cin + cbase) * taps + 4 * o4);
      }
#pragma unroll
      for (int u = 0; u < U4; ++u) {
        const int i = u * 256 + tid;
        if (i < 32 * run4) {
          const int q = i / run4, o4 = i - q * run4;
#pragma unroll
          for (int e = 0; e < 4; ++e) sm[q][4 * o4 + e] = tv[u][e];
        }
      }
    } else {
    constexpr int U = 9;                                   // nine loads in flight per thread before the first LDS store
    for (int j0 = 0; j0 < 32 * run; j0 += 256 * U) {
      float tmp[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = j0 + u * 256 + tid;
        const int q = i / run, o = i - q * run;
        const int co = (dgrad ? c : ct) * 32 + q, ci = (dgrad ? ct : c) * 32 + o / taps;
        tmp[u] = (i < 32 * run && co < d.Cout && ci < d.Cin) ? w[((size_t)co * d.Cin + (dgrad ? ct : c) * 32) * taps + o] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = j0 + u * 256 + tid;
        if (i < 32 * run) { const int q = i / run; sm[q][i - q * run] = tmp[u]; }
      }
    }
    }
    __syncthreads();
    T* dst = packed + d.dst_off + (size_t)tile * taps * 2 * 512;
    for (int v = tid; v < taps * 2 * 64; v += 256) {       // one 8-element fragment vector per iteration
      const int lane = v & 63, kk = (v >> 6) & 1, t = v >> 7;
      const int r = l16 ? kk * 16 + (lane & 15) : lane & 31, k0 = l16 ? 8 * (lane >> 4) : kk * 16 + 8 * (lane >> 5);
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = dgrad ? sm[k0 + e][r * taps + (taps - 1 - t)] : sm[r][(k0 + e) * taps + t];
      *reinterpret_cast<V16*>(dst + (size_t)v * 8) = pack<T>(o);
    }
  }
}

template <typename T>
hipError_t launch_pack(const float* params, void* packed, const PackDesc* descs_dev, int ndesc, hipStream_t s) {
  if (ndesc == 0) return hipSuccess;
  // row-major descriptors (and every fp32 one) element-wise; fragment-major ones through LDS (each kernel skips the other kind)
  hipLaunchKernelGGL(pack_weights_kernel<T>, dim3(288, ndesc), dim3(256), 0, s, params, reinterpret_cast<T*>(packed), descs_dev);
  if constexpr (sizeof(T) == 2)
    hipLaunchKernelGGL(pack_frag_kernel<T>, dim3(64, ndesc), dim3(256), 0, s, params, reinterpret_cast<T*>(packed), descs_dev);
  return hipGetLastError();
}
template hipError_t launch_pack<float>(const float*, void*, const PackDesc*, int, hipStream_t);
template hipError_t launch_pack<f16>(const float*, void*, const PackDesc*, int, hipStream_t);
template hipError_t launch_pack<bf16>(const float*, void*, const PackDesc*, int, hipStream_t);

}  // namespace pu
