// Kernel launch interface shared by the engine and the kernel translation units.
#pragma once
#include "pu_common.h"
#include <cstdlib>

namespace pu {

// ---------------------------------------------------------------- GroupNorm work fused into the convolutions (16-bit engines)
// Where the dropout decisions of a GroupNorm site come from: the counter-hash stream (seed, stream) or an injected keep mask.
struct DropSrc {
  float drop_p;                 // 0 = no dropout at this site
  uint64_t seed; uint32_t stream;
  const uint8_t* mask;          // optional injected keep mask, dense NHWC uint8 [B,H,W,C] (null: hash stream)
  int b0;                       // batch offset added to the sample index of the dropout counter
  const uint8_t* bits;          // optional: the keep decisions the forward saved (GNArgs::keep_bits), one byte per 8 channels
};
// keep decisions (bit e = keep) of the 8 consecutive channels starting at dense NHWC index `base` of the site
__device__ __forceinline__ uint32_t drop_keep_bits8(const uint8_t* mask, uint32_t dkey, uint32_t dthr, uint64_t base) {
  uint32_t bits = 0;
  if (mask) {
    const uint2 m = *reinterpret_cast<const uint2*>(mask + base);
#pragma unroll
    for (int e = 0; e < 4; ++e) { bits |= ((m.x >> (8 * e)) & 1u) << e; bits |= ((m.y >> (8 * e)) & 1u) << (e + 4); }
    return bits;
  }
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    const uint32_t r = drop_pair(dkey, base + e);
    bits |= ((r & 0xffffu) < dthr ? 1u : 0u) << e;
    bits |= ((r >> 16) < dthr ? 1u : 0u) << (e + 1);
  }
  return bits;
}
// Data-gradient epilogue of the convolution that FOLLOWS a GroupNorm(+scale/shift)+SiLU(+dropout) in the forward: instead of
// dh = d(loss)/d(GroupNorm output) the kernel stores  dv = dh * keep/(1-p) * silu'(A (x - mean) + Bp)  (the gradient at the
// pre-activation, rounded to T) and writes per-wave rows of  S1 = sum dv,  S2 = sum dv * (x - mean) * rstd  - pass 1 of the GroupNorm
// backward (src/networks.py:170-177 differentiated) without its own read of (x, dh) and with the exp / rcp / mask hash evaluated once
// per element instead of once per pass.  Rows: part[((b * slots + slot) * Cout + c) * 2 + {0, 1}], written, never accumulated.
struct GNBwdFuse {
  const void* x; int x_ld;      // GroupNorm input, same geometry as the data gradient being produced; null = off
  const float* coef;            // [B][C][4] = (A, Bp, group mean, group rstd), the forward's coefficients
  int C;                        // channels of the GroupNorm (must equal the stored planes of the data gradient)
  DropSrc drop;
  float* part; int cap;         // row buffer and its capacity in slots per image
  int* slots;                   // host pointer: slot count per image of the chosen tiling (0 = not produced: plain dh was stored)
};

// ---------------------------------------------------------------- convolution
struct ConvArgs {
  const void* in; int in_ld; int Cin;            // NHWC input view, logical channels
  const void* wpk; int cin_pk; int cout_pk; int taps;   // packed weights [cout_pk][taps][cin_pk]
  const float* bias;                             // fp32 [Cout] or null
  const void* res; int res_ld;                   // residual (same shape as out) or null
  const void* relu_mask; int relu_mask_ld;       // optional (conv3 / conv3p only): out = mask > 0 ? value : 0 - the ReLU backward of the layer
                                                 // whose output `relu_mask` is, folded into the data gradient that produces its gradient
  void* out; int out_ld; int Cout;
  int B, H, W;
  int relu; int accumulate;
  int frag_layout;                               // 1: wpk is fragment-major (conv3 kernel), see conv_uses_frag_layout()
  int mfma16;                                    // 1: wpk holds v_mfma_f32_16x16x32 fragments (conv_uses_mfma16(); PackDesc mode bit 2)
  // Fused GroupNorm statistics of the OUTPUT (optional): each wave adds the (sum, sum of squares) of the values it stores
  // into its own row  stat_out[((b * slots + slot) * Cout + c) * 2 + {0,1}]  (written, never accumulated -> deterministic).
  // The launcher reports the slot count per image of the tiling it chose through *stat_slots (host pointer; 0 = this
  // kernel configuration does not produce statistics, the consumer then falls back to its own statistics pass).
  float* stat_out; int stat_cap; int* stat_slots;
  GNBwdFuse gnb;                                 // optional (data gradient only: no bias / residual / ReLU / accumulate)
};
// 16-bit convolutions on >= 16-pixel-wide levels run the cout-split kernel whose weights are packed fragment-major:
//   [cout tile of 32][32-channel chunk][tap][k-step][lane 0..63][8]  =  W[32 ct + (lane & 31)][tap][32 c + 16 kk + 8 (lane >> 5) + e]
inline bool conv_uses_frag_layout(int elem_size, int H, int W) {
  return elem_size == 2 && ((W % 32 == 0 && H % 8 == 0) || (W % 16 == 0 && H % 16 == 0));
}
// A split-K weight gradient whose slabs still have to be summed (fixed order) and added into the fp32 gradient.
// Routing of a fragment-major convolution (shared by the launcher and by the weight packer, which must agree on the fragment shape):
// true -> the persistent, weight-resident conv3p_kernel (K <= 64 input channels; 32x32x16 fragments), false -> conv3_kernel.
inline bool conv3_goes_persistent(int cin_pk, int Cout, int H, int W) {
  static const int c64_mode = getenv("PU_C64_MODE") ? atoi(getenv("PU_C64_MODE")) : 1;   // 64 -> 64 class: 1 = conv3 on 256-pixel tiles (default),
                                                                                        // 0 = conv3 on 128-pixel tiles, 2 = conv3p
  static const bool no_conv3p = getenv("PU_NO_CONV3P") != nullptr;                       // diagnostic switch
  if (cin_pk > 64 || no_conv3p) return false;
  if (cin_pk == 64 && Cout > 32 && Cout <= 64 && W % 32 == 0 && ((H % 8 == 0 && c64_mode == 1) || (H % 4 == 0 && c64_mode == 0))) return false;
  if (!((W % 32 == 0 && H % 8 == 0) || (W % 16 == 0 && H % 16 == 0))) return false;
  if (Cout > 64 && cin_pk == 64) return false;          // two chunks of weights + 80 KB of staging exceed the LDS
  return true;
}
// conv3_kernel runs on v_mfma_f32_16x16x32 fragments (weights packed [cout tile][chunk][tap][cout half][lane][8], PackDesc mode bit 2):
// same FLOPs per cycle as 32x32x16, but the chip holds a higher clock on that shape under load (+10...13 % on the MFMA-bound
// convolutions, tools/conv_microbench.py A/B).  PU_MFMA16=0 switches back.
inline bool conv_uses_mfma16(int elem_size, int taps, int cin_pk, int Cout, int H, int W) {
  static const int mode = getenv("PU_MFMA16") ? atoi(getenv("PU_MFMA16")) : 1;
  (void)taps;
  // mode 2: only the cout-split instances (more than 64 output channels).  The <..,2,2> / <..,4,1> instances need
  // __launch_bounds__(256, 2) in this form to keep their second resident block (270 -> 242 registers; without it 64 -> 64 @128x128 fell
  // from 608 to 462 TFLOP/s)
  if (mode == 2 && Cout <= 64) return false;
  return mode != 0 && conv_uses_frag_layout(elem_size, H, W) && !conv3_goes_persistent(cin_pk, Cout, H, W);
}
struct WgradReduce { const float* slab; int split, taps, cout_pad, cin_pad, Cout, Cin; float* dw; float inv_scale; float* db0; float* db1; const float* inv_dev; int overwrite; };
struct WgradArgs {
  const void* dy; int dy_ld; int Cout;
  const void* in; int in_ld; int Cin;
  float* dw;                                     // fp32 [Cout][Cin][ks][ks], ADDED into
  int B, H, W; int taps;
  float* slab; long slab_floats;                 // split-K partial slabs (16-bit path; required there)
  float inv_scale;                               // parameter gradients are multiplied by this (loss-scale removal)
  const float* inv_scale_dev;                    // optional device float multiplied on top (sub-graph scale chosen on the device)
  float* dbias0; float* dbias1;                  // optional (16-bit path): bias gradient(s) = column sums of dy, ADDED into
  int overwrite;                                 // 16-bit path: dw / dbias are WRITTEN instead of added into (the fused ELBO backward, where
                                                 // every convolution parameter receives exactly one contribution: no memset of the buffer)
};
struct PackDesc { long src_off; long dst_off; int Cout, Cin, taps, rows_pk, k_pk, mode; };   // mode: 0 fwd, 1 dgrad; +2 = fragment-major

template <typename T> hipError_t launch_conv(const ConvArgs&, hipStream_t);
template <typename T> hipError_t launch_wgrad(const WgradArgs&, hipStream_t);
template <typename T> hipError_t launch_wgrad16(const WgradArgs&, hipStream_t);   // kernels_wgrad.hip (f16 / bf16): main + reduce on one stream
// The two halves separately: the split-K main kernel fills `slab` and describes the pending fixed-order reduction.
template <typename T> hipError_t launch_wgrad16_main(const WgradArgs&, hipStream_t, WgradReduce* out);
hipError_t launch_wgrad16_reduce(const WgradReduce&, hipStream_t);
template <typename T> hipError_t launch_pack(const float* params, void* packed, const PackDesc* descs_dev, int ndesc, hipStream_t);

// ---------------------------------------------------------------- layout / elementwise
// NCHW fp32 (two optional sources concatenated along C: src0 has c0 planes, src1 has c1 planes) -> NHWC T with
// C_alloc >= c0+c1 channels (extra channels zero). bstride0: batch stride of src0 in elements (0 = broadcast).
template <typename T> hipError_t launch_nchw_to_nhwc(const float* src0, long bstride0, int c0, const float* src1, int c1,
                                                     TV dst, hipStream_t);
// NHWC T (first C channels) -> NCHW fp32 [B,C,H,W]; accumulate=1 adds into dst
template <typename T> hipError_t launch_nhwc_to_nchw(TV src, int C, float* dst, int accumulate, hipStream_t);

enum { RS_NONE = 0, RS_DOWN = 1, RS_UP = 2 };

// GroupNorm(+adaptive scale/shift)+SiLU(+dropout)(+2x resample) on NHWC tensors.
struct GNArgs {
  TV x;                       // input [B,H,W,C]
  TV y;                       // output (H,W halved/doubled for RS_DOWN/RS_UP)
  int G; float eps;
  const float* gamma; const float* beta;          // fp32 [C]
  const float* scale; const float* shift;         // fp32 [C] each or null (adaptive: y = silu(gn*(1+scale)+shift))
  int resample;
  float drop_p; uint64_t drop_seed; uint32_t drop_stream;   // drop_p == 0 -> no dropout
  const uint8_t* drop_mask;   // optional injected keep mask, dense NHWC uint8 [B,H,W,C] of this site (null: counter hash)
  uint8_t* keep_bits;         // optional (16-bit engines, resample == 0): [B,H,W,C/8] bytes, bit e of byte i = keep decision of dense element
                              // 8 i + e.  The forward apply WRITES them; every backward kernel READS them instead of re-hashing (the mask
                              // hash was 40 % of the backward kernels' VALU instructions)
  // workspaces (fp32): part [B][nchunk][C][2], stat [B][G][2] (mean, rstd), coef [B][C][4] (A, Bp, mean, rstd; 16-byte aligned)
  float* part; float* stat; float* coef; int nchunk;
  // producer-fused statistics (see ConvArgs::stat_out): channels [0, pc0) from ps0 (ns0 slots per image), the rest from ps1;
  // ps0 == null -> the statistics pass runs here
  const float* ps0; int ns0; int pc0; const float* ps1; int ns1;
  int b0;                     // batch offset of this launch inside the tensor (batch-chunked launches; keeps the dropout index global)
};
template <typename T> hipError_t launch_gn_fwd(const GNArgs&, hipStream_t);
struct GNBwdArgs {
  GNArgs f;                   // same as forward (x, coef, stat are the saved forward values)
  TV dy;                      // grad wrt y (resampled geometry)
  TV dv;                      // scratch [B,H,W,C] (x geometry) for the pre-activation gradient
  TV dx; int accumulate;      // grad wrt x
  TV add;                     // optional extra addend of dx (same geometry; p == null: none): the identity-skip gradient of a block,
                              // folded into pass 2 instead of a separate add kernel (resample == 0 only)
  float* dgamma; float* dbeta; float* dscale; float* dshift;   // fp32 [C], ADDED into (nullable scale/shift)
  float* part2;               // [B][nchunk][C][2]
  float inv_scale;            // parameter gradients are multiplied by this
  float* coef2;               // [B][C][3] (P, Q, R)
  // pass 1 already done by the data-gradient epilogue (GNBwdFuse): `dy` holds dv, rows[b][nrows][C][2] hold its (S1, S2) partials
  const float* rows; int nrows;
};
template <typename T> hipError_t launch_gn_bwd(const GNBwdArgs&, hipStream_t);
template <typename T> hipError_t launch_gn_apply(const GNArgs&, hipStream_t);                  // apply kernel alone (coefficients in place)
template <typename T> hipError_t launch_gn_bwd_parts(const GNBwdArgs&, int parts, hipStream_t); // 1 pass 1 | 4 coefficients | 2 pass 2, or 8 small

template <typename T> hipError_t launch_resample(TV x, TV y, int mode, hipStream_t);                      // y = down/up(x)
template <typename T> hipError_t launch_resample_bwd(TV dy, TV dx, int mode, int accumulate, hipStream_t); // adjoint
template <typename T> hipError_t launch_add(TV src, TV dst, int accumulate, hipStream_t);                  // dst (+)= src
template <typename T> hipError_t launch_maxpool(TV x, TV y, hipStream_t);
template <typename T> hipError_t launch_maxpool_bwd(TV x, TV dy, TV dx, hipStream_t, bool relu_x = false); // dx written; relu_x: x is a ReLU output, its backward is folded in
template <typename T> hipError_t launch_relu_bwd(TV y, TV dy, hipStream_t);                                // dy *= (y>0), in place
// dbias[c] += sum over pixels of dy[..., c]; part: fp32 [nchunk][C]
template <typename T> hipError_t launch_bias_grad(TV dy, float* dbias0, float* dbias1_or_null, float* part, int nchunk, float inv_scale, hipStream_t);

// ---------------------------------------------------------------- Gaussian-encoder heads / latent / losses
// h[b,c] = mean over pixels; mu = Wmu h + bmu; ls = Wls h + bls.  hbuf fp32 [B][C]
template <typename T> hipError_t launch_heads_fwd(TV x, const float* wmu, const float* bmu, const float* wls, const float* bls,
                                                  int L, float* hbuf, float* mu, float* ls, hipStream_t);
// f16 latent-encoder backward: rescale the entry gradients (dmu, dls; n values each) by a power of two so that their max-abs
// lands in [target/2, target]; inv_out[0] = 1 / factor (multiplied into every parameter-gradient write of that sub-graph)
hipError_t launch_enc_rescale(float* dmu, float* dls, int n, float target, float* inv_out, hipStream_t s);
template <typename T> hipError_t launch_heads_bwd(TV x_shape, TV dx, const float* hbuf, const float* wmu, const float* wls,
                                                  const float* dmu, const float* dls, int L,
                                                  float* dwmu, float* dbmu, float* dwls, float* dbls, float inv_scale, hipStream_t, const float* inv_dev = nullptr);
struct LatentArgs {
  const float* mu_q; const float* ls_q; const float* mu_p; const float* ls_p;   // [B,L]
  const float* eps;            // [M,B,L]
  float* z;                    // [M,B,L]
  float* kl; float* kl2;       // [B]
  float* scalars;              // PU_NUM_SCALARS; kernel writes KL_MEAN, KL2_MEAN (fwd)
  int B, L, M;
};
hipError_t launch_latent_fwd(const LatentArgs&, hipStream_t);
struct LatentBwdArgs {
  LatentArgs f;
  const float* dz;             // [M,B,L] (d total / d z), nullable
  float beta1, beta2;
  float* dmu_q; float* dls_q; float* dmu_p; float* dls_p;   // [B,L] written
};
hipError_t launch_latent_bwd(const LatentBwdArgs&, hipStream_t);

// reconstruction loss on preds [B,M,C,HW] fp32 vs target [B,C,HW] fp32.
// Adds beta0-free loss value into scalars[PU_S_RECON] (must be zeroed before) and writes dpred = gscale * dloss/dpred.
hipError_t launch_recon(int kind, const float* preds, const float* target, float* dpred_or_null, float* scalars,
                        int B, int M, int C, long HW, float alpha, float gscale, hipStream_t);
hipError_t launch_finish_scalars(float* scalars, float beta0, float beta1, float beta2, int with_kl2, hipStream_t);

// ---------------------------------------------------------------- Fcomb (fused three 1x1 layers)
struct FcombArgs {
  TV feat;                     // NHWC T [Bf,H,W,F] (Bf == B, or 1 with bcast=1)
  int bcast;
  const float* z;              // [M,B,L]
  const float* w0; const float* b0; const float* w1; const float* b1; const float* w2; const float* b2;  // fp32 params (reference layout)
  int F, L, Cout, B, M;
  float* out;                  // fp32 [B,M,Cout,H,W]
  // optional fused reconstruction to physical units (ClimExDataset.residual_to_hr, climex_utils.py:277-285):
  //   out = hr_base[b,co,p] + out * (hr_std[co,p] + hr_eps), then softplus (climex_utils.py:41-45) when hr_softplus != 0
  const float* hr_base; const float* hr_std; float hr_eps; int hr_softplus; float hr_softplus_c;
};
// the value the Fcomb forward stores for residual `v` of sample b, plane co, pixel p
__device__ __forceinline__ float fcomb_emit(const FcombArgs& a, int b, int co, long p, long HW, float v) {
  if (a.hr_base) {
    v = a.hr_base[((long)b * a.Cout + co) * HW + p] + v * (a.hr_std[(long)co * HW + p] + a.hr_eps);
    if (a.hr_softplus) v = v > 20.f ? v : logf(expf(v) + 1.f) - a.hr_softplus_c;
  }
  return v;
}
template <typename T> hipError_t launch_fcomb_fwd(const FcombArgs&, hipStream_t);
struct FcombBwdArgs {
  FcombArgs f;
  const float* dout;           // fp32 [B,M,Cout,H,W]
  TV dfeat; int dfeat_accumulate;   // NHWC T grad wrt feat (nullable p)
  float* dz;                   // [M,B,L] written (must be zeroed by caller) or null
  float* dw0; float* db0; float* dw1; float* db1; float* dw2; float* db2;   // ADDED (atomics)
  float inv_scale;             // parameter gradients are multiplied by this (dz / dfeat keep the loss scale)
};
template <typename T> hipError_t launch_fcomb_bwd(const FcombBwdArgs&, hipStream_t);

hipError_t launch_fill(float* p, float v, long n, hipStream_t);
hipError_t launch_adamw_flat(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps, float wd,
                             float step_size, float inv_bc2_sqrt, hipStream_t, const float* skip_flag = nullptr);
hipError_t launch_nonfinite_flag(const float* g, long n, float* flag, hipStream_t);
hipError_t launch_zero_ranges(float* g, const long* ranges_dev, int n, hipStream_t);
hipError_t launch_scale_grads(float* g, long n, const float* scale_dev, float host_factor, hipStream_t);
hipError_t launch_adamw_flat_dev(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps, float wd,
                                 float* state, const float* skip_flag, hipStream_t, int phases = 3);   // 1 = advance the counter, 2 = update
hipError_t launch_mask_to_nhwc_u8(const float* src, uint8_t* dst, int B, int C, long HW, hipStream_t);

// ---------------------------------------------------------------- optional per-kernel-class profiling (bench roofline)
// When enabled, every MFMA conv launch is bracketed by HIP events on its own stream; prof_collect() sums elapsed time,
// launches, algorithmic FLOPs and bytes per kernel instantiation (the names match rocprofv3's kernel names).
struct ProfEntry { char name[128]; long launches; double ms; double flops; double bytes; };
void prof_enable(bool on);
bool prof_enabled();
void prof_record(const char* name, double flops, double bytes, hipStream_t s, bool begin);
int prof_collect(ProfEntry* out, int max_entries);   // syncs the recorded events; clears the log

// WMSE-MS-SSIM reconstruction loss (kernels_msssim.hip)
struct MsssimArgs {
  const float* pred;          // fp32 [B,M,C,H,W]
  const float* target;        // fp32 [B,C,H,W]
  int B, M, C, H, W;
  float alpha_w, beta_w, lam_w;
  float data_range;           // <= 0: inferred from the target on the device
  const float* data_range_dev; // optional device float that overrides both (read when the kernels run; clamped to >= 1e-5)
  float gscale;               // dpred = gscale * d(recon)/d(pred)
  float* ws; size_t ws_floats;     // workspace of msssim_ws_floats() floats
  float* scalars;             // PU_S_RECON is ADDED to, PU_S_WMSE added to (zero them first), PU_S_MSSSIM written
  float* dpred;               // fp32 [B,M,C,H,W] written, or null (forward only)
};
size_t msssim_ws_floats(int B, int M, int C, int H, int W);
hipError_t launch_wmse_msssim(const MsssimArgs&, hipStream_t);
hipError_t launch_lrinterp_residuals(const float* hr, int B, int C, int H, int W, int k, const float* mean, const float* stdv, float eps,
                                     float* inputs, float* targets, float* lrinterp, float* lr, hipStream_t s);
hipError_t launch_lr_stats(const float* hr, int N, int C, int H, int W, int k, float* mean_lr, float* std_lr, float* mean_hr, float* std_hr, hipStream_t s);
hipError_t launch_softplus_transform(float* d, long n, int inverse, float thr, float c, hipStream_t s);
hipError_t launch_destandardize(const float* x, const float* base, const float* stdv, const float* mean, float eps, int B, int n, long CHW,
                                float* out, hipStream_t s);

}  // namespace pu
