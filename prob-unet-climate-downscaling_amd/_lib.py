"""ctypes binding of libprobunet.so (include/probunet.h; the pu_op_* / pu_bench_conv test hooks are declared in
include/probunet_testing.h). No CPU fallback: loading fails loudly."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PU_LIB_PATH") or os.path.join(_HERE, "libprobunet.so")   # PU_LIB_PATH: diagnostic builds (tools/ablate_conv.sh)

PU_MAX_LEVELS = 8
PU_F32, PU_F16, PU_BF16 = 0, 1, 2
PU_PRIOR, PU_POSTERIOR = 0, 1
PU_RECON_AFCRPS, PU_RECON_L1, PU_RECON_WMSE_MSSSIM = 0, 1, 2
PU_S_TOTAL, PU_S_RECON, PU_S_KL_MEAN, PU_S_KL2_MEAN, PU_S_WMSE, PU_S_MSSSIM, PU_S_NONFINITE, PU_NUM_SCALARS = 0, 1, 2, 3, 4, 5, 6, 8
DTYPES = {"f32": PU_F32, "fp32": PU_F32, "float32": PU_F32, "f16": PU_F16, "fp16": PU_F16, "float16": PU_F16,
          "bf16": PU_BF16, "bfloat16": PU_BF16}


class PuConfig(C.Structure):
    _fields_ = [("input_channels", C.c_int32), ("num_classes", C.c_int32), ("latent_dim", C.c_int32), ("depth", C.c_int32),
                ("num_filters", C.c_int32 * PU_MAX_LEVELS), ("model_channels", C.c_int32),
                ("channel_mult", C.c_int32 * PU_MAX_LEVELS), ("H", C.c_int32), ("W", C.c_int32),
                ("max_batch", C.c_int32), ("max_members", C.c_int32), ("dtype", C.c_int32), ("dropout_p", C.c_float),
                ("grad_scale", C.c_float)]


class PuParamDesc(C.Structure):
    _fields_ = [("name", C.c_char * 96), ("ndim", C.c_int32), ("shape", C.c_int32 * 4), ("offset", C.c_int64),
                ("is_buffer", C.c_int32)]


class PuProfEntry(C.Structure):
    _fields_ = [("name", C.c_char * 128), ("launches", C.c_long), ("ms", C.c_double), ("flops", C.c_double), ("bytes", C.c_double)]


class ProbUNetLibraryError(RuntimeError):
    pass


_lib = None


def lib():
    """Load libprobunet.so once. torch must already be imported so that the HIP runtime it bundles
    (libamdhip64.so.7) is the one the library binds to."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ProbUNetLibraryError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C prob-unet-climate-downscaling_amd/csrc`). There is no CPU fallback.")
    import torch  # noqa: F401  (loads the HIP runtime first)
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, u64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_float
    L.pu_abi_version.restype = i32
    L.pu_last_error.restype = C.c_char_p; L.pu_last_error.argtypes = [vp]
    L.pu_create.restype = i32; L.pu_create.argtypes = [C.POINTER(PuConfig), i32, C.POINTER(vp)]
    L.pu_destroy.restype = i32; L.pu_destroy.argtypes = [vp]
    L.pu_param_table.restype = i32; L.pu_param_table.argtypes = [vp, C.POINTER(C.POINTER(PuParamDesc)), C.POINTER(i32)]
    L.pu_param_count.restype = i64; L.pu_param_count.argtypes = [vp]
    L.pu_workspace_bytes.restype = i64; L.pu_workspace_bytes.argtypes = [vp]
    L.pu_bind_params.restype = i32; L.pu_bind_params.argtypes = [vp, vp, vp]
    L.pu_bind_grads.restype = i32; L.pu_bind_grads.argtypes = [vp, vp]
    L.pu_params_changed.restype = i32; L.pu_params_changed.argtypes = [vp]
    L.pu_unet_fwd.restype = i32; L.pu_unet_fwd.argtypes = [vp, vp, vp, i32, i32, u64, vp]
    L.pu_unet_bwd.restype = i32; L.pu_unet_bwd.argtypes = [vp, vp, vp]
    L.pu_gauss_fwd.restype = i32; L.pu_gauss_fwd.argtypes = [vp, i32, vp, vp, vp, vp, i32, vp]
    L.pu_gauss_bwd.restype = i32; L.pu_gauss_bwd.argtypes = [vp, i32, vp, vp, vp]
    L.pu_fcomb_fwd.restype = i32; L.pu_fcomb_fwd.argtypes = [vp, vp, i64, vp, vp, i32, vp]
    L.pu_fcomb_bwd.restype = i32; L.pu_fcomb_bwd.argtypes = [vp, vp, vp, vp, vp]
    L.pu_elbo_fwd_bwd.restype = i32
    L.pu_elbo_fwd_bwd.argtypes = [vp, vp, vp, vp, i32, i32, i32, f32, f32, f32, f32, i32, u64, i32, vp, vp, vp, vp]
    L.pu_sample.restype = i32; L.pu_sample.argtypes = [vp, vp, vp, vp, i32, i32, vp, vp, vp, vp]
    L.pu_last_latent.restype = i32; L.pu_last_latent.argtypes = [vp, i32, vp, vp, i32, vp]
    L.pu_sample_hr.restype = i32
    L.pu_sample_hr.argtypes = [vp, vp, vp, vp, i32, i32, vp, vp, f32, i32, f32, vp, vp, vp, vp]
    L.pu_set_recon_wmse_msssim.restype = i32; L.pu_set_recon_wmse_msssim.argtypes = [vp, f32, f32, f32, f32]
    L.pu_set_recon_range_dev.restype = i32; L.pu_set_recon_range_dev.argtypes = [vp, vp]
    L.pu_lr_stats.restype = i32; L.pu_lr_stats.argtypes = [vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp]
    L.pu_lrinterp_to_residuals.restype = i32
    L.pu_lrinterp_to_residuals.argtypes = [vp, i32, i32, i32, i32, i32, vp, vp, f32, vp, vp, vp, vp, vp]
    L.pu_op_wmse_msssim.restype = i32
    L.pu_op_wmse_msssim.argtypes = [vp, vp, i32, i32, i32, i32, i32, f32, f32, f32, f32, f32, vp, vp, vp]
    L.pu_elbo_fwd_flops.restype = C.c_double; L.pu_elbo_fwd_flops.argtypes = [vp, i32, i32]
    L.pu_profile_enable.restype = i32; L.pu_profile_enable.argtypes = [i32]
    L.pu_adamw_step.restype = i32
    L.pu_adamw_step.argtypes = [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i64, vp]
    L.pu_adamw_step_guarded.restype = i32
    L.pu_adamw_step_guarded.argtypes = [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i64, vp, vp]
    L.pu_set_overlap.restype = i32; L.pu_set_overlap.argtypes = [vp, i32]
    L.pu_adamw_step_dev.restype = i32
    L.pu_adamw_step_dev.argtypes = [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, vp, vp, vp]
    L.pu_adamw_prepare.restype = i32; L.pu_adamw_prepare.argtypes = [vp, vp, f32, f32, f32, vp]
    L.pu_adamw_apply.restype = i32; L.pu_adamw_apply.argtypes = [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, vp, vp]
    L.pu_scale_grads.restype = i32; L.pu_scale_grads.argtypes = [vp, i64, vp, f32, vp]
    L.pu_nonfinite_flag.restype = i32; L.pu_nonfinite_flag.argtypes = [vp, i64, vp, vp]
    L.pu_set_grad_buckets.restype = i32; L.pu_set_grad_buckets.argtypes = [vp, i32]
    L.pu_grad_buckets.restype = i32; L.pu_grad_buckets.argtypes = [vp, C.POINTER(i64), C.POINTER(i64), i32, C.POINTER(i32)]
    L.pu_grad_bucket_wait.restype = i32; L.pu_grad_bucket_wait.argtypes = [vp, i32, vp]
    L.pu_set_sample_graph.restype = i32; L.pu_set_sample_graph.argtypes = [vp, i32]
    L.pu_sample_graph_stats.restype = i32; L.pu_sample_graph_stats.argtypes = [vp, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)]
    L.pu_drop_site_count.restype = i32; L.pu_drop_site_count.argtypes = [vp]
    L.pu_drop_site.restype = i32; L.pu_drop_site.argtypes = [vp, i32, C.c_char_p, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    L.pu_set_drop_masks.restype = i32; L.pu_set_drop_masks.argtypes = [vp, vp, i32, vp]
    L.pu_softplus_transform.restype = i32; L.pu_softplus_transform.argtypes = [vp, i64, i32, f32, f32, vp]
    L.pu_destandardize.restype = i32; L.pu_destandardize.argtypes = [vp, vp, vp, vp, f32, i32, i32, i32, i32, i32, vp, vp]
    L.pu_profile_collect.restype = i32; L.pu_profile_collect.argtypes = [C.POINTER(PuProfEntry), i32]
    L.pu_op_conv.restype = i32
    L.pu_op_conv.argtypes = [i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp]
    L.pu_bench_conv.restype = i32
    L.pu_bench_conv.argtypes = [i32, i32, i32, i32, i32, i32, i32, i32, i32, C.POINTER(C.c_float), vp]
    L.pu_bench_gn.restype = i32
    L.pu_bench_gn.argtypes = [i32, i32, i32, i32, i32, i32, C.c_float, i32, i32, C.POINTER(C.c_float), vp]
    L.pu_op_gnsilu.restype = i32
    L.pu_op_gnsilu.argtypes = [i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, f32, u64, vp]
    _lib = L
    return L


def check(rc, ctx=None, what=""):
    if rc != 0:
        msg = lib().pu_last_error(ctx)
        raise ProbUNetLibraryError(f"{what} failed (status {rc}): {msg.decode() if msg else ''}")


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def current_stream(device=None):
    """torch's current stream ON `device` (the model's device, which need not be torch's current device)."""
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
