"""ClimEx-shaped data transforms on the device.

The reference builds every training item on the host in `ClimExDataset.__getitem__` (climex_utils.py:197-225, type
"lrinterp_to_residuals"): AvgPool2d(k) -> nearest upsample -> per-pixel standardisation -> residual, one item at a time in the
DataLoader's main process (`num_workers=0`, main.py:77-100), and computes the standardisation statistics once with
`compute_stats` (:255-264).  Here the same arithmetic runs as two HIP kernels on whole batches that already live in HBM
(pu_lr_stats / pu_lrinterp_to_residuals in include/probunet.h).  There is no CPU fallback.
"""
from typing import Dict, Optional

import torch

from . import _lib as L


def _dev32(t: torch.Tensor) -> torch.Tensor:
    if not t.is_cuda:
        raise L.ProbUNetLibraryError("the data transforms run on the GPU: move the tensor to the device first (no CPU fallback)")
    return t.contiguous().float()


def lr_stats(hr: torch.Tensor, lowres_scale: int):
    """compute_stats (climex_utils.py:255-264): hr [N, C, H, W] -> ((mean_lr, std_lr), (mean_hr, std_hr)); the std is the
    unbiased one over the N items, of the k x k block means."""
    hr = _dev32(hr)
    N, C, H, W = hr.shape
    k = int(lowres_scale)
    if H % k or W % k:
        raise ValueError(f"H={H}, W={W} must be divisible by lowres_scale={k}")
    mean_lr = torch.empty(C, H // k, W // k, device=hr.device); std_lr = torch.empty_like(mean_lr)
    mean_hr = torch.empty(C, H, W, device=hr.device); std_hr = torch.empty_like(mean_hr)
    L.check(L.lib().pu_lr_stats(L.ptr(hr), N, C, H, W, k, L.ptr(mean_lr), L.ptr(std_lr), L.ptr(mean_hr), L.ptr(std_hr), L.current_stream()),
            None, "pu_lr_stats")
    return (mean_lr, std_lr), (mean_hr, std_hr)


def lrinterp_to_residuals(hr: torch.Tensor, lowres_scale: int, mean_hr: torch.Tensor, std_hr: torch.Tensor,
                          epsilon: float = 1e-10, want_lr: bool = True) -> Dict[str, torch.Tensor]:
    """The "lrinterp_to_residuals" item transform (climex_utils.py:197-225) on a batch hr [B, C, H, W].  Returns the keys the
    reference's item dict holds: inputs, targets, hr, lrinterp (and lr)."""
    hr = _dev32(hr)
    B, C, H, W = hr.shape
    k = int(lowres_scale)
    if H % k or W % k:
        raise ValueError(f"H={H}, W={W} must be divisible by lowres_scale={k}")
    mean_hr = _dev32(mean_hr); std_hr = _dev32(std_hr)
    if tuple(mean_hr.shape) != (C, H, W) or tuple(std_hr.shape) != (C, H, W):
        raise ValueError("mean_hr / std_hr must be [C, H, W]")
    inputs = torch.empty_like(hr); targets = torch.empty_like(hr); lrinterp = torch.empty_like(hr)
    lr = torch.empty(B, C, H // k, W // k, device=hr.device) if want_lr else None
    L.check(L.lib().pu_lrinterp_to_residuals(L.ptr(hr), B, C, H, W, k, L.ptr(mean_hr), L.ptr(std_hr), float(epsilon), L.ptr(inputs),
                                             L.ptr(targets), L.ptr(lrinterp), L.ptr(lr), L.current_stream()), None, "pu_lrinterp_to_residuals")
    out = {"inputs": inputs, "targets": targets, "hr": hr, "lrinterp": lrinterp}
    if lr is not None:
        out["lr"] = lr
    return out


def softplus(data: torch.Tensor, threshold: float = 20.0, c: float = 1e-7) -> torch.Tensor:
    """climex_utils.softplus (:41-45), IN PLACE like the reference: v > threshold ? v : log(exp(v) + 1) - c."""
    if not data.is_cuda or data.dtype != torch.float32 or not data.is_contiguous():
        raise L.ProbUNetLibraryError("softplus works in place on a contiguous fp32 device tensor (no CPU fallback)")
    L.check(L.lib().pu_softplus_transform(L.ptr(data), data.numel(), 0, float(threshold), float(c), L.current_stream(data.device)), None,
            "pu_softplus_transform")
    return data


def softplus_inv(data: torch.Tensor, threshold: float = 20.0, c: float = 1e-7) -> torch.Tensor:
    """climex_utils.softplus_inv (:36-40), IN PLACE: v > threshold ? v : log(exp(v + c) - 1) - the load-time pre-transform of
    precipitation and of tasmax - tasmin (climex_utils.py:141-143, `transfo=True`)."""
    if not data.is_cuda or data.dtype != torch.float32 or not data.is_contiguous():
        raise L.ProbUNetLibraryError("softplus_inv works in place on a contiguous fp32 device tensor (no CPU fallback)")
    L.check(L.lib().pu_softplus_transform(L.ptr(data), data.numel(), 1, float(threshold), float(c), L.current_stream(data.device)), None,
            "pu_softplus_transform")
    return data


def destandardize(x: torch.Tensor, std_hr: torch.Tensor, base: Optional[torch.Tensor] = None, mean_hr: Optional[torch.Tensor] = None,
                  epsilon: float = 1e-10) -> torch.Tensor:
    """invstand_residual / residual_to_hr (climex_utils.py:270-285) in one pass:
    out = (base if given) + x * (std_hr + epsilon) + (mean_hr if given).  x [B, C, H, W] or [B, n, C, H, W]; base [B, C, H, W]."""
    x = _dev32(x)
    five = x.dim() == 5
    xx = x if five else x.unsqueeze(1)
    B, n, C_, H, W = xx.shape
    std_hr = _dev32(std_hr)
    if tuple(std_hr.shape) != (C_, H, W):
        raise ValueError("std_hr must be [C, H, W]")
    base = _dev32(base) if base is not None else None
    mean_hr = _dev32(mean_hr) if mean_hr is not None else None
    if base is not None and tuple(base.shape) != (B, C_, H, W):
        raise ValueError("base (lrinterp) must be [B, C, H, W]")
    out = torch.empty_like(xx)
    L.check(L.lib().pu_destandardize(L.ptr(xx.contiguous()), L.ptr(base), L.ptr(std_hr), L.ptr(mean_hr), float(epsilon), B, n, C_, H, W,
                                     L.ptr(out), L.current_stream(x.device)), None, "pu_destandardize")
    return out if five else out[:, 0]


class ClimExTransform:
    """Device-side stand-in for the parts of ClimExDataset the training / sampling loops touch: statistics, the batched item
    transform of the "lrinterp_to_residuals" pipeline (kind="lrinterp_to_hr": targets are the standardised hr itself,
    climex_utils.py:228-250), `invstand_residual` and `residual_to_hr`."""

    def __init__(self, lowres_scale: int = 4, epsilon: float = 1e-10, kind: str = "lrinterp_to_residuals"):
        if kind not in ("lrinterp_to_residuals", "lrinterp_to_hr"):
            raise ValueError("kind must be 'lrinterp_to_residuals' or 'lrinterp_to_hr'")
        self.lowres_scale = int(lowres_scale)
        self.epsilon = float(epsilon)
        self.kind = kind
        self.lrstats = None

    def fit(self, hr: torch.Tensor):
        self.lrstats = lr_stats(hr, self.lowres_scale)
        return self

    def __call__(self, hr: torch.Tensor, timestamps: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        if self.lrstats is None:
            raise RuntimeError("call fit(hr) first (the reference computes the statistics lazily from its whole hr array)")
        out = lrinterp_to_residuals(hr, self.lowres_scale, self.lrstats[1][0], self.lrstats[1][1], self.epsilon)
        if self.kind == "lrinterp_to_hr":
            out["targets"] = out["targets"] + out["inputs"]           # hr_stand = residual + lrinterp_stand (climex_utils.py:241-242)
        if timestamps is not None:
            out["timestamps"] = timestamps
        return out

    def invstand_residual(self, standardized_residual: torch.Tensor) -> torch.Tensor:        # climex_utils.py:270-274
        mean = self.lrstats[1][0] if self.kind == "lrinterp_to_hr" else None
        return destandardize(standardized_residual, self.lrstats[1][1], None, mean, self.epsilon)

    def residual_to_hr(self, residual: torch.Tensor, lrinterp: torch.Tensor) -> torch.Tensor:  # climex_utils.py:277-278
        mean = self.lrstats[1][0] if self.kind == "lrinterp_to_hr" else None
        return destandardize(residual, self.lrstats[1][1], lrinterp, mean, self.epsilon)
