/*
 * probunet_testing.h — test / micro-benchmark hooks of libprobunet.so.  NOT part of the product ABI (include/probunet.h): they
 * exist so that tests/ can pin each kernel against a torch fp32 reference of the same op and tools/conv_microbench.py can time one
 * convolution launch.  No caller of the reference's interface needs them; signatures may change without an ABI bump.
 */
#ifndef PROBUNET_TESTING_H
#define PROBUNET_TESTING_H

#include "probunet.h"

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)   /* the library is built with -fvisibility=hidden: only this C ABI is exported */

/* ---- single-op entry points (used by tests/ to pin each kernel against a torch fp32 reference) ----------- */
/* 3x3 (ks=3) or 1x1 (ks=1) convolution on NCHW fp32 tensors through the engine's NHWC implicit-GEMM kernels in
 * `dtype`. mode 0: y = conv(x,w)+b (relu optional); 1: dx = dgrad(dy,w); 2: dw = wgrad(dy,x) (w/dw in [Cout,Cin,ks,ks]).
 * All pointers device fp32. Syncs the stream. */
int pu_op_conv(int dtype, int mode, int ks, int relu, int B, int Cin, int Cout, int H, int W,
               const float* x, const float* w, const float* bias, const float* dy, float* out, void* stream);
/* Micro-benchmark: average microseconds (HIP events, `iters` back-to-back launches) of one convolution kernel launch on
 * NHWC data already in HBM. mode 0 forward, 1 data gradient, 2 weight gradient (incl. its slab reduce). Syncs. */
int pu_bench_conv(int dtype, int mode, int ks, int B, int Cin, int Cout, int H, int W, int iters, float* out_us, void* stream);
/* Micro-benchmark of the GroupNorm+SiLU kernels on random NHWC 16-bit data [B,H,W,C] (C % 8 == 0): out_us[0] apply, [1] backward pass 1,
 * [2] backward pass 2, [3] the single-kernel backward of small tensors (0 where it does not apply), each the average of `iters`
 * back-to-back launches.  flags: 1 = pass 2 accumulates into dx, 2 = pass 2 adds an extra tensor. */
int pu_bench_gn(int dtype, int resample, int B, int C, int H, int W, float drop_p, int flags, int iters, float* out_us, void* stream);
/* GroupNorm(+scale/shift)+SiLU(+dropout drop_p with the counter-hash mask of drop_seed; resample 0 only) with optional
 * 2x resample (0 none, 1 avg-pool down, 2 nearest up), forward and backward, on NCHW fp32 tensors. Syncs. */
int pu_op_gnsilu(int dtype, int resample, int B, int C, int H, int W, const float* x, const float* gamma,
                 const float* beta, const float* scale_shift_or_null, float* y,
                 const float* dy_or_null, float* dx, float* dgamma, float* dbeta, float* dscale_shift,
                 float drop_p, uint64_t drop_seed, void* stream);

/* WMSE-MS-SSIM loss and its gradient on fp32 device tensors pred [B,M,C,H,W], target [B,C,H,W] (stand-alone test hook for
 * the kernels behind PU_RECON_WMSE_MSSSIM). out_scalars: PU_NUM_SCALARS floats (PU_S_RECON, PU_S_WMSE, PU_S_MSSSIM filled);
 * dpred (nullable) receives gscale * d(recon)/d(pred). Syncs. */
int pu_op_wmse_msssim(const float* pred, const float* target, int B, int M, int C, int H, int W, float alpha_w, float beta_w,
                      float lam_w, float data_range, float gscale, float* out_scalars, float* dpred, void* stream);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* PROBUNET_TESTING_H */
