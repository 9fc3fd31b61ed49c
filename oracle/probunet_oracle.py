"""CPU oracle for the Probabilistic U-Net ELBO / sampling hot path.

*** TEST INFRASTRUCTURE ONLY ***  Only tests/, __graft_entry__.smoke() and the
`cpu_baseline` leg of bench.py may import this module, and only as the checker.
The product (prob-unet-climate-downscaling_amd/) never imports it and has no CPU
fallback: it raises if the HIP library is missing.

This is a *restatement* (torch-CPU functional fp32), not a copy, of the algorithm in
the reference (paths relative to /root/reference):

  src/networks.py:49-92    Conv2d (3x3 / 1x1 / 2x resample)      -> conv / avgpool2 / nearest2
  src/networks.py:97-107   GroupNorm (G = min(32, C//4), eps 1e-5) -> group_norm
  src/networks.py:134-187  UNetBlock.forward                       -> unet_block
  src/networks.py:226-333  UNet.__init__/forward                   -> unet_layout / unet_forward
  src/prob_unet.py:12-85   AxisAlignedConvGaussian                 -> gauss_layout / gauss_forward
  src/prob_unet.py:87-138  Fcomb (tile+cat+3x 1x1)                 -> fcomb_forward (z folded into a bias)
  src/prob_unet.py:194-224 ProbabilisticUNet.forward               -> sample_forward
  src/prob_unet.py:273-317 elbo, afCRPS variant (commented at HEAD; what train_prob_unet_model.py:133 unpacks)
  src/prob_unet.py:325-381 elbo, L1 variant                        -> elbo(recon="l1")
  src/prob_unet_utils.py:171-234 afcrps_loss                       -> afcrps (pair form, no [B,M,M,...] temp)
  torch.distributions.kl._kl_normal_normal + Independent           -> kl_diag_normal

Restated simplifications (each is pinned by the golden vectors in tests/golden/, which were
captured from the *imported reference*, see tools/make_golden.py):
  * emb = silu(map_label(zeros)) == 0 (networks.py:310-316; map_label has no bias) so the adaptive
    scale/shift of every block is just affine.bias -> (scale, shift) = bias[:C], bias[C:].
  * depthwise 2x2 stride-2 conv with the 0.25 filter == 2x2 average pool; transposed one with
    filter*4 == nearest-neighbour 2x upsample (networks.py:84-87).
  * Fcomb.tile + cat + conv1x1 == W[:, :F] . feat + (W[:, F:] . z + b) since z is spatially constant.
  * posterior stem generalised to (Cin + Cout) input planes; for Cout == Cin identical to the
    reference, for Cout < Cin identical to the reference fed a zero-padded target.

  src/prob_unet.py:229-267 elbo, live WMSE-MS-SSIM variant          -> elbo(recon="wmse_msssim")
  src/prob_unet_utils.py:270-305 wmse_ms_ssim_loss                  -> wmse_ms_ssim_loss
  src/climex_utils.py:36-46,197-225,255-285 data transforms         -> softplus_climex, lrinterp_to_residuals, lr_stats, residual_to_hr

The ClimEx transforms are PINNED too: tools/make_golden_climex.py imports src/climex_utils.py (absent xarray / dask / cartopy /
bottleneck / cftime replaced by empty in-memory modules; the NetCDF constructor is not run) and captures softplus, softplus_inv,
compute_stats, the item transform, invstand_residual and residual_to_hr into tests/golden/climex.npz
(tests/test_oracle_golden.py::test_climex_transforms_match_reference).

Parity status: PINNED by golden vectors generated from the reference import (no denial encountered), with ONE exception:
  ms_ssim() below restates the third-party dependency `pytorch-msssim==1.0.0` (pinned in uv.lock:786-794, imported at
  prob_unet_utils.py:8, called at :297 with win_size=7, size_average=True).  The package is absent from this image and cannot
  be installed (no network), and the reference holds no test or golden value for it: the MS-SSIM arithmetic is
  **PARITY UNPINNED** — restated from the package's published algorithm (Wang et al. 2003 as implemented by pytorch-msssim:
  5 scales, weights (0.0448, 0.2856, 0.3001, 0.2363, 0.1333), K = (0.01, 0.03), Gaussian window sigma 1.5 applied separably with
  no padding, relu on the per-scale means, 2x2 average pooling with padding = size % 2).  Everything around it — the WMSE half,
  the lam-combination, the data_range inference, the ensemble-mean rule — IS pinned (tests/golden/wmse.json, captured from the
  imported reference with ms_ssim replaced by a recording constant).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]


# ----------------------------------------------------------------------------- config / layouts
@dataclass
class Config:
    input_channels: int
    num_classes: int
    latent_dim: int
    num_filters: List[int]
    model_channels: int
    channel_mult: List[int]
    num_blocks: int = 2
    base_res: int = 128          # prob_unet.py:159 hard-codes img_resolution=(128,128): only names keys
    eps: float = 1e-5
    dropout: float = 0.10        # networks.py:239

    @property
    def feat_channels(self) -> int:   # UNet out_channels = num_filters[0] (prob_unet.py:161)
        return self.num_filters[0]


@dataclass
class BlockSpec:
    name: str                    # state_dict prefix, e.g. "unet.enc.128x128_block0"
    kind: str                    # "conv" (plain 3x3) | "block" (UNetBlock)
    cin: int
    cout: int
    up: bool = False
    down: bool = False
    concat: bool = False         # decoder blocks whose input is cat([x, skip])

    @property
    def skip_kind(self) -> str:  # networks.py:156-159
        if self.kind != "block":
            return "none"
        if self.cin != self.cout:
            return "conv1x1"
        if self.up or self.down:
            return "resample"
        return "identity"


def gn_groups(c: int) -> int:    # networks.py:100
    return min(32, c // 4)


def unet_layout(cfg: Config) -> Tuple[List[BlockSpec], List[BlockSpec]]:
    """Encoder / decoder block lists in execution order (networks.py:259-295)."""
    mc, mults = cfg.model_channels, cfg.channel_mult
    enc: List[BlockSpec] = []
    cout = cfg.input_channels
    for level, mult in enumerate(mults):
        r = cfg.base_res >> level
        p = f"unet.enc.{r}x{r}"
        if level == 0:
            cin, cout = cout, mc * mult
            enc.append(BlockSpec(f"{p}_conv", "conv", cin, cout))
        else:
            enc.append(BlockSpec(f"{p}_down", "block", cout, cout, down=True))
        for i in range(cfg.num_blocks):
            cin, cout = cout, mc * mult
            enc.append(BlockSpec(f"{p}_block{i}", "block", cin, cout))
    skips = [b.cout for b in enc]
    dec: List[BlockSpec] = []
    for level, mult in reversed(list(enumerate(mults))):
        r = cfg.base_res >> level
        p = f"unet.dec.{r}x{r}"
        if level == len(mults) - 1:
            dec.append(BlockSpec(f"{p}_in0", "block", cout, cout))
            dec.append(BlockSpec(f"{p}_in1", "block", cout, cout))
        else:
            dec.append(BlockSpec(f"{p}_up", "block", cout, cout, up=True))
        for i in range(cfg.num_blocks + 1):
            cin = cout + skips.pop()
            cout = mc * mult
            dec.append(BlockSpec(f"{p}_block{i}", "block", cin, cout, concat=True))
    return enc, dec


def gauss_conv_indices(n_levels: int) -> List[List[int]]:
    """nn.Sequential indices of the three convs of each level (prob_unet.py:34-48)."""
    out, idx = [], 0
    for i in range(n_levels):
        if i != 0:
            idx += 1                       # MaxPool2d slot
        lv = []
        for _ in range(3):
            lv.append(idx)
            idx += 2                       # conv + ReLU
        out.append(lv)
    return out


def param_shapes(cfg: Config) -> "Dict[str, Tuple[int, ...]]":
    """Every state_dict entry (parameters AND the resample_filter buffers) with the reference's key
    and shape, in the reference's registration order."""
    shapes: Dict[str, Tuple[int, ...]] = {}
    emb = cfg.model_channels * 4                      # channel_mult_emb=4 (networks.py:236,245)
    shapes["unet.map_label.weight"] = (emb, 1)
    enc, dec = unet_layout(cfg)

    def block(b: BlockSpec):
        p = b.name
        if b.kind == "conv":
            shapes[f"{p}.weight"] = (b.cout, b.cin, 3, 3)
            shapes[f"{p}.bias"] = (b.cout,)
            return
        shapes[f"{p}.norm0.weight"] = (b.cin,)
        shapes[f"{p}.norm0.bias"] = (b.cin,)
        shapes[f"{p}.conv0.weight"] = (b.cout, b.cin, 3, 3)
        shapes[f"{p}.conv0.bias"] = (b.cout,)
        if b.up or b.down:
            shapes[f"{p}.conv0.resample_filter"] = (1, 1, 2, 2)
        shapes[f"{p}.affine.weight"] = (2 * b.cout, emb)
        shapes[f"{p}.affine.bias"] = (2 * b.cout,)
        shapes[f"{p}.norm1.weight"] = (b.cout,)
        shapes[f"{p}.norm1.bias"] = (b.cout,)
        shapes[f"{p}.conv1.weight"] = (b.cout, b.cout, 3, 3)
        shapes[f"{p}.conv1.bias"] = (b.cout,)
        if b.skip_kind == "conv1x1":
            shapes[f"{p}.skip.weight"] = (b.cout, b.cin, 1, 1)
            shapes[f"{p}.skip.bias"] = (b.cout,)
            if b.up or b.down:
                shapes[f"{p}.skip.resample_filter"] = (1, 1, 2, 2)
        elif b.skip_kind == "resample":
            shapes[f"{p}.skip.resample_filter"] = (1, 1, 2, 2)

    for b in enc + dec:
        block(b)
    c_last = dec[-1].cout
    shapes["unet.out_norm.weight"] = (c_last,)
    shapes["unet.out_norm.bias"] = (c_last,)
    shapes["unet.out_conv.weight"] = (cfg.feat_channels, c_last, 3, 3)
    shapes["unet.out_conv.bias"] = (cfg.feat_channels,)

    for net, cin0 in (("prior", cfg.input_channels), ("posterior", cfg.input_channels + cfg.num_classes)):
        cin = cin0
        for lv, idxs in enumerate(gauss_conv_indices(len(cfg.num_filters))):
            for j, idx in enumerate(idxs):
                shapes[f"{net}.encoder.{idx}.weight"] = (cfg.num_filters[lv], cin, 3, 3)
                shapes[f"{net}.encoder.{idx}.bias"] = (cfg.num_filters[lv],)
                cin = cfg.num_filters[lv]
        shapes[f"{net}.conv_mu.weight"] = (cfg.latent_dim, cfg.num_filters[-1], 1, 1)
        shapes[f"{net}.conv_mu.bias"] = (cfg.latent_dim,)
        shapes[f"{net}.conv_log_sigma.weight"] = (cfg.latent_dim, cfg.num_filters[-1], 1, 1)
        shapes[f"{net}.conv_log_sigma.bias"] = (cfg.latent_dim,)

    f0 = cfg.feat_channels
    shapes["fcomb.layers.0.weight"] = (f0, f0 + cfg.latent_dim, 1, 1)
    shapes["fcomb.layers.0.bias"] = (f0,)
    shapes["fcomb.layers.2.weight"] = (f0, f0, 1, 1)
    shapes["fcomb.layers.2.bias"] = (f0,)
    shapes["fcomb.layers.4.weight"] = (cfg.num_classes, f0, 1, 1)
    shapes["fcomb.layers.4.bias"] = (cfg.num_classes,)
    return shapes


# ----------------------------------------------------------------------------- primitive ops
def avgpool2(x: Tensor) -> Tensor:
    return F.avg_pool2d(x, 2)


def nearest2(x: Tensor) -> Tensor:
    return x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)


def group_norm(x: Tensor, w: Tensor, b: Tensor, eps: float) -> Tensor:
    B, C, H, W = x.shape
    G = gn_groups(C)
    xg = x.reshape(B, G, -1)
    mean = xg.mean(dim=2, keepdim=True)
    var = xg.var(dim=2, unbiased=False, keepdim=True)
    xh = ((xg - mean) * torch.rsqrt(var + eps)).reshape(B, C, H, W)
    return xh * w.view(1, C, 1, 1) + b.view(1, C, 1, 1)


def silu(x: Tensor) -> Tensor:
    return x * torch.sigmoid(x)


def conv3x3(x: Tensor, w: Tensor, b: Optional[Tensor]) -> Tensor:
    return F.conv2d(x, w, b, padding=1)


def conv1x1(x: Tensor, w: Tensor, b: Optional[Tensor]) -> Tensor:
    return F.conv2d(x, w, b)


# ----------------------------------------------------------------------------- U-Net
def unet_block(P: Params, b: BlockSpec, x: Tensor, eps: float,
               drop_mask: Optional[Tensor] = None, drop_p: float = 0.0) -> Tensor:
    """networks.py:166-187 with emb == 0. drop_mask (0/1, conv1-input shaped) overrides RNG dropout."""
    p = b.name
    orig = x
    h = silu(group_norm(x, P[f"{p}.norm0.weight"], P[f"{p}.norm0.bias"], eps))
    if b.down:
        h = avgpool2(h)
    if b.up:
        h = nearest2(h)
    h = conv3x3(h, P[f"{p}.conv0.weight"], P[f"{p}.conv0.bias"])
    ab = P[f"{p}.affine.bias"]
    scale, shift = ab[: b.cout], ab[b.cout:]
    h = group_norm(h, P[f"{p}.norm1.weight"], P[f"{p}.norm1.bias"], eps)
    h = silu(h * (scale.view(1, -1, 1, 1) + 1) + shift.view(1, -1, 1, 1))
    if drop_mask is not None:
        h = h * drop_mask / (1.0 - drop_p)
    h = conv3x3(h, P[f"{p}.conv1.weight"], P[f"{p}.conv1.bias"])
    sk = orig
    if b.down:
        sk = avgpool2(sk)
    if b.up:
        sk = nearest2(sk)
    if b.skip_kind == "conv1x1":
        sk = conv1x1(sk, P[f"{p}.skip.weight"], P[f"{p}.skip.bias"])
    return h + sk


def unet_forward(P: Params, cfg: Config, x: Tensor,
                 drop_masks: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """networks.py:299-333. drop_masks: optional {block name: 0/1 mask} for injected dropout."""
    enc, dec = unet_layout(cfg)
    skips: List[Tensor] = []
    dm = drop_masks or {}
    for b in enc:
        if b.kind == "conv":
            x = conv3x3(x, P[f"{b.name}.weight"], P[f"{b.name}.bias"])
        else:
            x = unet_block(P, b, x, cfg.eps, dm.get(b.name), cfg.dropout)
        skips.append(x)
    for b in dec:
        if b.concat:
            x = torch.cat([x, skips.pop()], dim=1)
        x = unet_block(P, b, x, cfg.eps, dm.get(b.name), cfg.dropout)
    x = silu(group_norm(x, P["unet.out_norm.weight"], P["unet.out_norm.bias"], cfg.eps))
    return conv3x3(x, P["unet.out_conv.weight"], P["unet.out_conv.bias"])


# ----------------------------------------------------------------------------- Gaussian encoders
def gauss_forward(P: Params, cfg: Config, net: str, x: Tensor, target: Optional[Tensor] = None
                  ) -> Tuple[Tensor, Tensor]:
    """prob_unet.py:56-85 -> (mu [B,L], log_sigma [B,L]). sigma = exp(log_sigma) + 1e-7."""
    if net == "posterior":
        assert target is not None
        x = torch.cat([x, target], dim=1)
    for lv, idxs in enumerate(gauss_conv_indices(len(cfg.num_filters))):
        if lv != 0:
            x = F.max_pool2d(x, 2)
        for idx in idxs:
            x = torch.relu(conv3x3(x, P[f"{net}.encoder.{idx}.weight"], P[f"{net}.encoder.{idx}.bias"]))
    h = x.mean(dim=(2, 3))                                             # [B, Cf]
    mu = h @ P[f"{net}.conv_mu.weight"].flatten(1).t() + P[f"{net}.conv_mu.bias"]
    ls = h @ P[f"{net}.conv_log_sigma.weight"].flatten(1).t() + P[f"{net}.conv_log_sigma.bias"]
    return mu, ls


def sigma_of(log_sigma: Tensor) -> Tensor:
    return torch.exp(log_sigma) + 1e-7                                  # prob_unet.py:84


def kl_diag_normal(mu_q: Tensor, sig_q: Tensor, mu_p: Tensor, sig_p: Tensor) -> Tensor:
    """KL(q||p) summed over the latent axis -> [B] (torch kl._kl_normal_normal + Independent(…,1))."""
    var_ratio = (sig_q / sig_p) ** 2
    t1 = ((mu_q - mu_p) / sig_p) ** 2
    return (0.5 * (var_ratio + t1 - 1.0 - torch.log(var_ratio))).sum(dim=1)


# ----------------------------------------------------------------------------- Fcomb
def fcomb_forward(P: Params, cfg: Config, feat: Tensor, z: Tensor) -> Tensor:
    """prob_unet.py:120-138 with the z-tile folded into a per-sample bias."""
    f0 = cfg.feat_channels
    w0 = P["fcomb.layers.0.weight"].flatten(1)                         # [F0, F0+L]
    zb = z @ w0[:, f0:].t() + P["fcomb.layers.0.bias"]                  # [B, F0]
    h = torch.einsum("oc,bchw->bohw", w0[:, :f0], feat) + zb[:, :, None, None]
    h = torch.relu(h)
    h = torch.relu(conv1x1(h, P["fcomb.layers.2.weight"], P["fcomb.layers.2.bias"]))
    return conv1x1(h, P["fcomb.layers.4.weight"], P["fcomb.layers.4.bias"])


# ----------------------------------------------------------------------------- losses
def afcrps(ens: Tensor, target: Tensor, alpha: float = 0.95) -> Tensor:
    """prob_unet_utils.py:171-234 in pair form:
    sum_{j!=k}(|xj-y|+|xk-y|-(1-e)|xj-xk|) = 2(M-1) sum_j |xj-y| - 2(1-e) sum_{j<k} |xj-xk|."""
    B, M, C, H, W = ens.shape
    e = (1.0 - alpha) / M
    d = (ens - target.unsqueeze(1)).abs().sum(dim=1)                    # [B,C,H,W]
    pair = torch.zeros_like(d)
    for j in range(M):
        for k in range(j + 1, M):
            pair = pair + (ens[:, j] - ens[:, k]).abs()
    per_b = ((M - 1) * d - (1.0 - e) * pair).sum(dim=(1, 2, 3)) / (M * (M - 1) * C * H * W)
    return per_b.mean()


def l1_recon(pred: Tensor, target: Tensor) -> Tensor:
    return (pred - target).abs().mean()                                # prob_unet.py:362


# ----------------------------------------------------------------------------- model-level entry points
# ---------------------------------------------------------------------------- WMSE-MS-SSIM (prob_unet_utils.py:270-305)
MS_SSIM_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)


def gauss_window(size: int = 7, sigma: float = 1.5) -> Tensor:
    """pytorch-msssim `_fspecial_gauss_1d`: exp(-(i - size//2)^2 / (2 sigma^2)), normalised, fp32."""
    c = torch.arange(size, dtype=torch.float32) - size // 2
    g = torch.exp(-(c ** 2) / (2 * sigma ** 2))
    return g / g.sum()


def gaussian_filter(x: Tensor, win: Tensor) -> Tensor:
    """Separable depthwise 'valid' filtering (rows, then columns), as pytorch-msssim `gaussian_filter`."""
    C = x.shape[1]
    k = win.numel()
    x = F.conv2d(x, win.view(1, 1, k, 1).repeat(C, 1, 1, 1), groups=C)
    return F.conv2d(x, win.view(1, 1, 1, k).repeat(C, 1, 1, 1), groups=C)


def ssim_maps(X: Tensor, Y: Tensor, win: Tensor, data_range: float):
    """pytorch-msssim `_ssim`: per-(image, channel) spatial means of the ssim map and of the cs map."""
    C1, C2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    mu1, mu2 = gaussian_filter(X, win), gaussian_filter(Y, win)
    mu1_sq, mu2_sq, mu12 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    s1 = gaussian_filter(X * X, win) - mu1_sq
    s2 = gaussian_filter(Y * Y, win) - mu2_sq
    s12 = gaussian_filter(X * Y, win) - mu12
    cs_map = (2 * s12 + C2) / (s1 + s2 + C2)
    ssim_map = ((2 * mu12 + C1) / (mu1_sq + mu2_sq + C1)) * cs_map
    return ssim_map.flatten(2).mean(-1), cs_map.flatten(2).mean(-1)


def ms_ssim(X: Tensor, Y: Tensor, data_range: float, win_size: int = 7, win_sigma: float = 1.5, size_average: bool = True) -> Tensor:
    """pytorch-msssim 1.0.0 `ms_ssim` (PARITY UNPINNED, see the module header)."""
    if min(X.shape[-2:]) <= (win_size - 1) * 2 ** 4:
        raise AssertionError("Image size should be larger than %d due to the 4 downsamplings in ms-ssim" % ((win_size - 1) * 2 ** 4))
    win = gauss_window(win_size, win_sigma).to(X.dtype)
    w = torch.tensor(MS_SSIM_WEIGHTS, dtype=X.dtype)
    mcs = []
    for i in range(5):
        ssim_pc, cs = ssim_maps(X, Y, win, data_range)
        if i < 4:
            mcs.append(torch.relu(cs))
            pad = [s % 2 for s in X.shape[2:]]
            X = F.avg_pool2d(X, kernel_size=2, padding=pad)
            Y = F.avg_pool2d(Y, kernel_size=2, padding=pad)
    stack = torch.stack(mcs + [torch.relu(ssim_pc)], dim=0)          # [level, image, channel]
    val = torch.prod(stack ** w.view(-1, 1, 1), dim=0)
    return val.mean() if size_average else val.mean(1)


def wmse_ms_ssim_loss(pred: Tensor, target: Tensor, alpha: float = 0.007, beta: float = 0.048, lam: float = 0.0,
                      data_range: Optional[float] = None, ms_ssim_fn=None):
    """prob_unet_utils.py:270-305.  Returns (combined, wmse, 1 - ms_ssim).  `ms_ssim_fn` lets a test substitute the
    third-party call the way tools/make_golden.py did when it captured tests/golden/wmse.json."""
    if pred.dim() == 5:
        pred = pred.mean(1)
    if data_range is None:
        data_range = float((target.max() - target.min()).clamp(min=1e-5))
    weights = torch.clamp(alpha * torch.exp(beta * target), max=1.0)
    wmse = (weights * (pred - target).pow(2)).mean()
    msv = (ms_ssim_fn or ms_ssim)(pred, target, data_range=data_range, win_size=7, size_average=True)
    msl = 1.0 - msv
    return lam * wmse + (1.0 - lam) * msl, wmse, msl


# ---------------------------------------------------------------------------- ClimEx data transforms (climex_utils.py)
def softplus_climex(d: Tensor, threshold: float = 20.0, c: float = 1e-7) -> Tensor:
    """climex_utils.py:41-45 (out of place)."""
    return torch.where(d > threshold, d, torch.log(torch.exp(d) + 1.0) - c)


def softplus_inv_climex(d: Tensor, threshold: float = 20.0, c: float = 1e-7) -> Tensor:
    """climex_utils.py:36-40 (out of place): the pre-transform applied to precipitation / tasmax - tasmin at load time."""
    return torch.where(d > threshold, d, torch.log(torch.exp(d + c) - 1.0))


def invstand_residual(x: Tensor, mean_hr: Tensor, std_hr: Tensor, kind: str = "lrinterp_to_residuals", epsilon: float = 1e-10) -> Tensor:
    """climex_utils.py:270-274: the *_to_hr pipelines add the mean back, the *_to_residuals ones only rescale."""
    if kind in ("lr_to_hr", "lrinterp_to_hr"):
        return x * (std_hr + epsilon) + mean_hr
    return x * (std_hr + epsilon)


def lr_stats(hr: Tensor, k: int):
    """climex_utils.py:255-264 compute_stats: hr [N,C,H,W] -> (mean_lr, std_lr), (mean_hr, std_hr)."""
    lr = F.avg_pool2d(hr, k)
    mean, std = lr.mean(dim=0), lr.std(dim=0)
    rep = lambda t: t.repeat_interleave(k, dim=1).repeat_interleave(k, dim=2)
    return (mean, std), (rep(mean), rep(std))


def lrinterp_to_residuals(hr: Tensor, k: int, mean_hr: Tensor, std_hr: Tensor, epsilon: float = 1e-10):
    """climex_utils.py:197-225 item transform, batched: hr [B,C,H,W] -> dict(inputs, targets, lrinterp, lr)."""
    lr = F.avg_pool2d(hr, k)
    lrinterp = F.interpolate(lr, scale_factor=k)                     # nearest
    li = (lrinterp - mean_hr) / (std_hr + epsilon)
    hs = (hr - mean_hr) / (std_hr + epsilon)
    return dict(inputs=li, targets=hs - li, lrinterp=lrinterp, lr=lr)


def residual_to_hr(residual: Tensor, lrinterp: Tensor, std_hr: Tensor, epsilon: float = 1e-10) -> Tensor:
    """climex_utils.py:270-285 for the residual dataset types: lrinterp + residual * (std + eps).
    residual [B,n,C,H,W] or [B,C,H,W]; lrinterp [B,C,H,W]; std_hr [C,H,W]."""
    if residual.dim() == 5:
        return lrinterp.unsqueeze(1) + residual * (std_hr + epsilon)
    return lrinterp + residual * (std_hr + epsilon)


def elbo(P: Params, cfg: Config, x: Tensor, target: Tensor, eps: Tensor,
         beta0: float, beta1: float, beta2: float = 0.0, alpha: float = 0.95, recon: str = "afcrps",
         drop_masks: Optional[Dict[str, Tensor]] = None, alpha_w: float = 0.007, beta_w: float = 0.048, lam_w: float = 0.0,
         data_range: Optional[float] = None):
    """afCRPS ELBO (prob_unet.py:273-317) or L1 ELBO (:325-381) with explicit noise eps [M,B,L].
    Returns dict(total, recon, kl[B], kl2[B], feat, mu_p, ls_p, mu_q, ls_q, preds[B,M,C,H,W])."""
    feat = unet_forward(P, cfg, x, drop_masks)
    mu_p, ls_p = gauss_forward(P, cfg, "prior", x)
    mu_q, ls_q = gauss_forward(P, cfg, "posterior", x, target)
    sq, sp = sigma_of(ls_q), sigma_of(ls_p)
    M = eps.shape[0]
    preds = torch.stack([fcomb_forward(P, cfg, feat, mu_q + sq * eps[m]) for m in range(M)], dim=1)
    if recon == "afcrps":
        if M < 2:
            raise ValueError(f"M must be at least 2 to compute afCRPS but got M={M}")
        rec = afcrps(preds, target, alpha)
    elif recon == "l1":
        rec = l1_recon(preds[:, 0], target)
    elif recon == "wmse_msssim":                                     # prob_unet.py:244-252: per-member loss, averaged
        parts = [wmse_ms_ssim_loss(preds[:, m], target, alpha_w, beta_w, lam_w, data_range) for m in range(M)]
        rec = torch.stack([p[0] for p in parts]).mean()
        extra = dict(wmse=parts[-1][1], msssim=parts[-1][2])         # the reference returns the LAST member's components
    else:
        raise ValueError(recon)
    kl = kl_diag_normal(mu_q, sq, mu_p, sp)
    kl2 = kl_diag_normal(mu_q, sq, torch.zeros_like(mu_q), torch.ones_like(sq))
    total = beta0 * rec + beta1 * kl.mean()
    if recon == "l1":
        total = total + beta2 * kl2.mean()
    out = dict(total=total, recon=rec, kl=kl, kl2=kl2, feat=feat, mu_p=mu_p, ls_p=ls_p,
               mu_q=mu_q, ls_q=ls_q, preds=preds)
    if recon == "wmse_msssim":
        out.update(extra)
    return out


def sample_forward(P: Params, cfg: Config, x: Tensor, eps: Tensor, target: Optional[Tensor] = None):
    """ProbabilisticUNet.forward (prob_unet.py:194-224): posterior if target given, else prior.
    eps [n,B,L] -> outputs [B,n,C,H,W] with UNet/encoder evaluated ONCE (latent_exploration.py:119-129)."""
    feat = unet_forward(P, cfg, x)
    if target is not None:
        mu, ls = gauss_forward(P, cfg, "posterior", x, target)
    else:
        mu, ls = gauss_forward(P, cfg, "prior", x)
    sg = sigma_of(ls)
    outs = torch.stack([fcomb_forward(P, cfg, feat, mu + sg * eps[i]) for i in range(eps.shape[0])], dim=1)
    return dict(out=outs, mu=mu, sigma=sg, feat=feat)


def elbo_with_grads(P: Params, cfg: Config, x: Tensor, target: Tensor, eps: Tensor, **kw):
    """Run elbo() with autograd on every floating parameter; returns (result dict, {name: grad})."""
    Pg = {k: (v.detach().clone().requires_grad_(True) if v.is_floating_point() and "resample_filter" not in k else v)
          for k, v in P.items()}
    r = elbo(Pg, cfg, x, target, eps, **kw)
    r["total"].backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v))
             for k, v in Pg.items() if isinstance(v, Tensor) and v.requires_grad}
    return {k: (v.detach() if isinstance(v, Tensor) else v) for k, v in r.items()}, grads
