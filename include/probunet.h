/*
 * probunet.h — C ABI of libprobunet.so: MI355X (gfx950) engine for the Probabilistic U-Net ELBO
 * forward+backward and prior-sampling hot path.
 *
 * The reference (MaryamAlipourH/prob-unet-climate-downscaling) has no FFI: its seam is the Python class
 * ProbabilisticUNet (src/prob_unet.py:140-267). Each entry point below names the reference interface it
 * replaces; the Python shim prob-unet-climate-downscaling_amd/prob_unet.py binds them with ctypes
 * (see INTEGRATION.md for the binding a maintainer of the reference would add).
 *
 * Conventions
 *   - return 0 on success, negative pu_status on failure; never throws; pu_last_error(ctx) gives a message.
 *   - every pointer is a raw DEVICE pointer owned by the caller unless stated; the library never frees them.
 *   - user-facing tensors are fp32, NCHW, contiguous (what the reference passes, train_prob_unet_model.py:123-125).
 *   - all work is enqueued on the hipStream_t passed in (void* here so that the header needs no HIP include);
 *     no hidden device synchronisation except where stated ("syncs").
 *   - one pu_ctx per (process, device); a ctx is not thread-safe.
 *   - activations inside the engine are NHWC in the ctx compute dtype; parameters always have an fp32 master copy
 *     in the caller's flat buffer (reference state_dict order, see pu_param_table).
 */
#ifndef PROBUNET_H
#define PROBUNET_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)   /* the library is built with -fvisibility=hidden: only this C ABI is exported */

typedef struct pu_ctx pu_ctx;

typedef enum {
  PU_OK = 0,
  PU_ERR_INVALID = -1,      /* bad argument / unsupported configuration */
  PU_ERR_HIP = -2,          /* a HIP runtime call failed */
  PU_ERR_STATE = -3,        /* call order violated (e.g. backward without forward, params not bound) */
  PU_ERR_NOMEM = -4
} pu_status;

typedef enum { PU_F32 = 0, PU_F16 = 1, PU_BF16 = 2 } pu_dtype;
typedef enum { PU_PRIOR = 0, PU_POSTERIOR = 1 } pu_net;
typedef enum { PU_RECON_AFCRPS = 0, PU_RECON_L1 = 1, PU_RECON_WMSE_MSSSIM = 2 } pu_recon;

#define PU_MAX_LEVELS 8

/* ProbabilisticUNet.__init__ arguments (prob_unet.py:146) + engine sizing. */
typedef struct {
  int32_t input_channels;              /* x planes */
  int32_t num_classes;                 /* target / output planes */
  int32_t latent_dim;
  int32_t depth;                       /* len(num_filters) == len(channel_mult) */
  int32_t num_filters[PU_MAX_LEVELS];  /* prior/posterior encoder widths; num_filters[0] = U-Net feature planes */
  int32_t model_channels;
  int32_t channel_mult[PU_MAX_LEVELS];
  int32_t H, W;                        /* field size; must be divisible by 2^(depth-1), deepest level >= 8x8 */
  int32_t max_batch;                   /* activations are planned for this many field pairs */
  int32_t max_members;                 /* max M (ELBO ensemble) / max n (samples per input in pu_sample) */
  int32_t dtype;                       /* pu_dtype: arithmetic type of activations and MFMA operands */
  float   dropout_p;                   /* UNetBlock dropout (networks.py:239), applied when train != 0 */
  float   grad_scale;                  /* static loss scale for the activation gradients of pu_elbo_fwd_bwd (removed again
                                          from every parameter gradient); 0 = automatic: 1 for f32/bf16, a power of two
                                          sized from B*M*C*H*W for f16 so that d(loss)/d(pred) ~ O(1) does not underflow */
} pu_config;

/* One state_dict entry of the reference (prob_unet.py / networks.py registration order). */
typedef struct {
  char    name[96];                    /* e.g. "unet.enc.128x128_block0.conv0.weight" */
  int32_t ndim;
  int32_t shape[4];
  int64_t offset;                      /* element offset into the flat fp32 buffers; -1 for buffers (resample_filter) */
  int32_t is_buffer;                   /* 1 = registered buffer (constant 0.25), not trained */
} pu_param_desc;

/* Scalars written by pu_elbo_fwd_bwd (device array of PU_NUM_SCALARS floats). */
enum { PU_S_TOTAL = 0, PU_S_RECON = 1, PU_S_KL_MEAN = 2, PU_S_KL2_MEAN = 3, PU_S_WMSE = 4, PU_S_MSSSIM = 5, PU_S_NONFINITE = 6, PU_NUM_SCALARS = 8 };

/* ---- lifetime (replaces ProbabilisticUNet.__init__, prob_unet.py:146-189) ------------------------------- */
int pu_create(const pu_config* cfg, int device, pu_ctx** out);
int pu_destroy(pu_ctx* ctx);
const char* pu_last_error(pu_ctx* ctx);            /* ctx may be NULL: last creation error */
int pu_abi_version(void);

/* ---- parameters (replaces nn.Module.state_dict()/parameters()) --------------------------- */
int pu_param_table(pu_ctx* ctx, const pu_param_desc** out, int* n);
int64_t pu_param_count(pu_ctx* ctx);               /* number of fp32 elements in the flat buffers */
/* flat_params / flat_grads: device fp32 arrays of pu_param_count elements. Gradients are WRITTEN (not
 * accumulated) by pu_elbo_fwd_bwd and the *_bwd calls that state so. */
int pu_bind_params(pu_ctx* ctx, float* flat_params, float* flat_grads);
/* Move the gradient destination only (same layout, another device buffer): lets a caller alternate between two gradient buffers so that
 * gradients it still references are not overwritten by the next fused backward.  Does not invalidate the packed weights. */
int pu_bind_grads(pu_ctx* ctx, float* flat_grads);
/* Tell the engine the fp32 master weights changed (optimizer step / load_state_dict): low-precision packed
 * copies are rebuilt on the next call. */
int pu_params_changed(pu_ctx* ctx);

/* ---- sub-modules (replace model.unet(x), model.prior(x)/model.posterior(x,y), model.fcomb(f,z)) --------- */
/* networks.py:299-333. x [B,Cin,H,W] -> feat [B,F0,H,W]. train!=0 enables dropout with drop_seed. */
int pu_unet_fwd(pu_ctx*, const float* x, float* feat, int B, int train, uint64_t drop_seed, void* stream);
/* Injected dropout masks ("the ABI must accept eps and a dropout seed/mask"; networks.py:177 draws them from
 * torch's RNG, which no other implementation can reproduce): the U-Net has one dropout site per UNetBlock, in execution order
 * (encoder blocks, then decoder blocks).  pu_drop_site describes site i (reference state_dict prefix of its block, C, H, W).
 * pu_set_drop_masks takes the sites' keep masks concatenated in that order, each fp32 [B, C_i, H_i, W_i] (non-zero = keep); they
 * replace the counter-hash stream in every later train-mode call (forward AND its backward) with batch size B until cleared
 * with masks == NULL.  Kept elements are scaled by 1 / (1 - dropout_p) as F.dropout does. */
int pu_drop_site_count(pu_ctx*);
int pu_drop_site(pu_ctx*, int i, char name[96], int* C, int* H, int* W);
int pu_set_drop_masks(pu_ctx*, const float* masks_or_null, int B, void* stream);
/* backward of the LAST pu_unet_fwd: dfeat [B,F0,H,W] -> parameter grads ADDED into flat_grads (x needs no grad). */
int pu_unet_bwd(pu_ctx*, const float* dfeat, void* stream);
/* prob_unet.py:56-85. -> mu [B,L], log_sigma [B,L] (sigma = exp(log_sigma) + 1e-7 is the caller's). */
int pu_gauss_fwd(pu_ctx*, int which, const float* x, const float* target_or_null, float* mu, float* log_sigma,
                 int B, void* stream);
int pu_gauss_bwd(pu_ctx*, int which, const float* dmu, const float* dlog_sigma, void* stream);
/* prob_unet.py:120-138. feat [B,F0,H,W] with batch stride feat_bstride elements (0 => one feature map broadcast
 * to all B, the expand() pattern of latent_exploration.py:125), z [B,L] -> out [B,Cout,H,W]. */
int pu_fcomb_fwd(pu_ctx*, const float* feat, int64_t feat_bstride, const float* z, float* out, int B, void* stream);
/* backward of the LAST pu_fcomb_fwd: dout -> dfeat (nullable; [B,F0,H,W] dense), dz (nullable; [B,L]),
 * parameter grads ADDED into flat_grads. */
int pu_fcomb_bwd(pu_ctx*, const float* dout, float* dfeat, float* dz, void* stream);

/* ---- fused training step (replaces model.elbo(...) + loss.backward(), train_prob_unet_model.py:133-140) -- */
/* eps: [M,B,L] reparameterisation noise (required: the caller owns the RNG, explicit for parity).
 * out_scalars: PU_NUM_SCALARS floats; out_kl: [B] KL(q||p); out_kl2 (nullable): [B] KL(q||N(0,I)).
 * with_backward != 0 also runs the whole backward and WRITES d(total)/d(param) into flat_grads (all entries,
 * zeros for dead parameters). */
int pu_elbo_fwd_bwd(pu_ctx*, const float* x, const float* target, const float* eps, int B, int M, int recon_kind,
                    float beta0, float beta1, float beta2, float alpha, int train, uint64_t drop_seed,
                    int with_backward, float* out_scalars, float* out_kl, float* out_kl2, void* stream);

/* ---- data-parallel hand-off: the backward of pu_elbo_fwd_bwd finishes the flat gradient buffer in pieces, deepest
 * decoder level first.  pu_grad_buckets reports up to `max` contiguous element ranges [lo, hi) of flat_grads in the order in which
 * the LAST pu_elbo_fwd_bwd(with_backward) completes them (they partition [0, pu_param_count)); pu_grad_bucket_wait makes `stream`
 * wait (hipStreamWaitEvent, no host sync) until every kernel writing bucket k has finished, so that a collective enqueued on that
 * stream afterwards overlaps the rest of the backward.  pu_set_grad_buckets(n) chooses how many U-Net buckets the plan is cut
 * into (0 = off: no events are recorded; default).  While buckets are on, the f16 overflow flag is NOT computed inside
 * pu_elbo_fwd_bwd (the collective rewrites the buffer in place); use pu_nonfinite_flag on the averaged gradients instead. */
int pu_set_grad_buckets(pu_ctx*, int n_unet_buckets);
int pu_grad_buckets(pu_ctx*, int64_t* lo, int64_t* hi, int max, int* n);
int pu_grad_bucket_wait(pu_ctx*, int k, void* stream);
/* g[0..n) *= host_factor * (scale_dev ? *scale_dev : 1), as ONE conditional pass: when the factor is exactly 1 (the grad_output of
 * loss.backward(), a device scalar the host does not know) the kernel returns after one load per thread and the buffer is not touched.
 * Lets the Python shim hand the engine's gradient buffer to p.grad without a scaled copy.  g must be 16-byte aligned. */
int pu_scale_grads(float* g, int64_t n, const float* scale_dev_or_null, float host_factor, void* stream);
/* flag[0] = 1.0f if any of the n floats is inf / NaN, else unchanged (zero it first); ~60 us for the 76 M gradients of cfg3. */
int pu_nonfinite_flag(const float* g, int64_t n, float* flag, void* stream);

/* Location / scale [B, L] of the prior (PU_PRIOR) or posterior (PU_POSTERIOR) computed by the LAST forward of that encoder
 * (pu_elbo_fwd_bwd, pu_gauss_fwd or pu_sample): what the reference leaves behind in `self.prior_latent_space` /
 * `self.posterior_latent_space` (prob_unet.py:214,220,241-242). scale = exp(log_sigma) + 1e-7 (prob_unet.py:84). */
int pu_last_latent(pu_ctx*, int which, float* mu, float* sigma, int B, void* stream);

/* ---- sampling (replaces n x model(x, training=False), train_prob_unet_model.py:244-247, and
 *      latent_exploration.py:119-129): U-Net + prior (or posterior if target given) ONCE, then n x Fcomb. ---- */
/* eps [n,B,L]; out [B,n,Cout,H,W]; mu/sigma [B,L] nullable.
 * hipGraph: with pu_set_sample_graph(ctx, 1) the launch sequence of pu_sample / pu_sample_hr is captured once per distinct
 * argument tuple (all pointers, B, n) and replayed with one hipGraphLaunch afterwards (BASELINE config 5: "hipGraph-captured
 * sample+fcomb").  Parameters are read at replay time, so weight updates are seen as long as pu_params_changed is NOT pending
 * (a pending re-pack runs eagerly before the replay).  Off by default; the captured graphs die with the ctx. */
int pu_set_sample_graph(pu_ctx*, int on);
/* What the hipGraph path has done since the ctx was created (no guessing from timings): graphs captured and instantiated, launches
 * served by hipGraphLaunch (the launch that follows a capture included), and calls that ran eagerly WHILE the graph path was on
 * (first sight of an argument tuple, a failed capture, profiling enabled).  The cache key is the full argument tuple - every pointer,
 * B, n - so a caller that wants replays must pass the same buffers again (ProbabilisticUNet.sample(out=...)). */
int pu_sample_graph_stats(pu_ctx*, int64_t* captures, int64_t* replays, int64_t* eager_fallbacks);
int pu_sample(pu_ctx*, const float* x, const float* target_or_null, const float* eps, int B, int n,
              float* out, float* mu, float* sigma, void* stream);

/* Same, with ClimExDataset.residual_to_hr (climex_utils.py:277-285; applied per sample on the host in
 * train_prob_unet_model.py:246, latent_exploration.py:142) fused into the Fcomb store:
 *   out[b,s,c,:,:] = lrinterp[b,c,:,:] + residual * (resid_std[c,:,:] + epsilon),
 * optionally followed by climex_utils.softplus (:41-45; value > 20 ? value : log(exp(value) + 1) - softplus_c). */
int pu_sample_hr(pu_ctx*, const float* x, const float* target_or_null, const float* eps, int B, int n,
                 const float* lrinterp, const float* resid_std, float epsilon, int softplus, float softplus_c,
                 float* out, float* mu, float* sigma, void* stream);

/* ---- WMSE-MS-SSIM reconstruction term (the live `elbo`, prob_unet.py:229-267; wmse_ms_ssim_loss, prob_unet_utils.py:270-305):
 *      parameters used by pu_elbo_fwd_bwd(recon_kind = PU_RECON_WMSE_MSSSIM).  data_range <= 0: inferred on the device as
 *      max(target) - min(target) clamped at 1e-5 (prob_unet_utils.py:288-289; under data parallelism pass a fixed value).
 *      Scalars: PU_S_RECON = mean over the M members of lam * WMSE + (1 - lam) * (1 - MS-SSIM); PU_S_WMSE / PU_S_MSSSIM =
 *      the LAST member's WMSE and (1 - MS-SSIM), which is what the reference returns for logging.  Requires H, W > 96.
 *      MS-SSIM restates pytorch-msssim 1.0.0 (absent from this image): parity unpinned, see oracle/probunet_oracle.py. */
int pu_set_recon_wmse_msssim(pu_ctx*, float alpha_w, float beta_w, float lam_w, float data_range);
/* Value range of the WMSE-MS-SSIM term taken from a DEVICE float (read when the loss kernels run, clamped to >= 1e-5) instead of the
 * data_range argument above or the on-device inference from the local target; null restores those.  Used under data parallelism: the
 * reference infers max(target) - min(target) over the batch it sees (prob_unet_utils.py:288-289), so ranks exchange their minima and
 * maxima (two scalar all-reduces on the device) and hand the global range in here without a host round trip.  The pointer must stay
 * valid until the call that uses it has run. */
int pu_set_recon_range_dev(pu_ctx*, const float* range_dev_or_null);

/* ---- ClimEx-shaped data transforms on the device (climex_utils.py:197-225, :255-264); all pointers device fp32 NCHW ----
 * pu_lr_stats: mean / unbiased std over the N items of the k x k block means (AvgPool2d(k)); *_lr are [C,H/k,W/k],
 *   *_hr their repeat_interleave expansion [C,H,W] (any output may be null).
 * pu_lrinterp_to_residuals: lr = AvgPool2d(k)(hr); lrinterp = nearest upsample; inputs = (lrinterp - mean)/(std + epsilon);
 *   targets = (hr - mean)/(std + epsilon) - inputs.  lrinterp [B,C,H,W] and lr [B,C,H/k,W/k] are optional outputs. */
int pu_lr_stats(const float* hr, int N, int C, int H, int W, int k, float* mean_lr, float* std_lr, float* mean_hr, float* std_hr,
                void* stream);
int pu_lrinterp_to_residuals(const float* hr, int B, int C, int H, int W, int k, const float* mean_hr, const float* std_hr,
                             float epsilon, float* inputs, float* targets, float* lrinterp, float* lr, void* stream);

/* climex_utils.softplus (:41-45, inverse = 0) / softplus_inv (:36-40, inverse != 0), in place on n floats like the reference:
 *   softplus: v > threshold ? v : log(exp(v) + 1) - c;     softplus_inv: v > threshold ? v : log(exp(v + c) - 1). */
int pu_softplus_transform(float* data, int64_t n, int inverse, float threshold, float c, void* stream);
/* ClimExDataset.invstand_residual / residual_to_hr (climex_utils.py:270-285) on tensors that already exist:
 *   out[b,s,c,:,:] = (base ? base[b,c,:,:] : 0) + x[b,s,c,:,:] * (std[c,:,:] + epsilon) + (mean ? mean[c,:,:] : 0)
 * x, out [B,n,C,H,W]; base (lrinterp) [B,C,H,W] nullable; mean [C,H,W] nullable (the *_to_hr pipelines add it back). */
int pu_destandardize(const float* x, const float* base, const float* std_hr, const float* mean_hr, float epsilon, int B, int n, int C,
                     int H, int W, float* out, void* stream);

/* ---- optimizer (replaces torch.optim.AdamW(model.parameters(), lr=1e-4).step(), main.py:103, train_prob_unet_model.py:141) on
 *      the flat buffers: one fused pass; exp_avg / exp_avg_sq are caller-owned fp32 arrays of n elements; step counts from 1. */
int pu_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int64_t step, void* stream);
/* Same; when skip_flag (device float, nullable) is non-zero the kernel leaves parameters and moments untouched.  The f16 engine
 * sets out_scalars[PU_S_NONFINITE] = 1 in pu_elbo_fwd_bwd when a parameter gradient came out inf / NaN (fp16 overflow): passing
 * that address skips the poisoned step on the device, the way torch.cuda.amp.GradScaler does with a host sync. */
int pu_adamw_step_guarded(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                          float beta2, float eps, float weight_decay, int64_t step, const float* skip_flag, void* stream);

/* Same update with the step counter kept ON THE DEVICE: state is a device array of 4 floats, state[0] = number of updates applied
 * so far (start at 0).  When skip_flag is non-zero nothing changes, the counter included, so the bias corrections follow the
 * number of APPLIED updates exactly as torch.optim.AdamW under a GradScaler that skipped the step. */
int pu_adamw_step_dev(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2,
                      float eps, float weight_decay, float* state, const float* skip_flag, void* stream);
/* The two halves of pu_adamw_step_dev, for updating only SOME ranges of the flat buffers (torch.optim skips parameters whose
 * .grad is None: no decay, no moment update): pu_adamw_prepare advances the counter once (unless skipped), then pu_adamw_apply
 * runs the update on each contiguous range that has gradients (pointers already offset by the caller). */
int pu_adamw_prepare(float* state, const float* skip_flag, float lr, float beta1, float beta2, void* stream);
int pu_adamw_apply(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2,
                   float eps, float weight_decay, const float* state, void* stream);

/* ---- introspection for bench/roofline ------------------------------------------------------------------ */
int64_t pu_workspace_bytes(pu_ctx*);
/* conv + matmul FLOPs (2*MAC) of one ELBO forward for batch B, members M (BASELINE.md §2 counting). */
double pu_elbo_fwd_flops(pu_ctx*, int B, int M);

/* Per-kernel-class timing of the MFMA convolution launches (HIP events on the launch stream), for bench.py's roofline.
 * pu_profile_enable(1) ... run steps ... pu_profile_collect() waits for the recorded events and returns entries. */
typedef struct { char name[128]; long launches; double ms; double flops; double bytes; } pu_prof_entry;
int pu_profile_enable(int on);
/* Side-stream overlap (weight gradients / latent encoders beside the U-Net chain) on or off; default on for f16/bf16.
 * bench.py switches it off for the per-kernel roofline step so that kernel durations are not inflated by co-running kernels. */
int pu_set_overlap(pu_ctx* ctx, int on);
int pu_profile_collect(pu_prof_entry* out, int max_entries);

/* The single-op test hooks and the convolution micro-benchmark (pu_op_*, pu_bench_conv) are NOT part of this ABI: they are
 * declared in include/probunet_testing.h. */

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* PROBUNET_H */
