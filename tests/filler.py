"""Closed-form parameter / input fillers shared by tools/make_golden.py (applied to the imported
reference model) and the tests (applied to the oracle and the HIP engine).

Needed because the reference zero-initialises every block's conv1 and the out_conv
(networks.py:154,247,297), so at init the U-Net outputs exactly 0 and a parity test would pass vacuously.
"""
import math
import torch


def fill_tensor(name: str, shape, k: int) -> torch.Tensor:
    n = 1
    for s in shape:
        n *= s
    i = torch.arange(n, dtype=torch.float64)
    base = torch.sin(0.37 * i + 1.3 * k + 0.11 * (i % 7))
    if name.endswith("resample_filter"):
        return torch.full(tuple(shape), 0.25, dtype=torch.float32)
    if name.endswith(".weight") and len(shape) == 4:            # conv weights
        fan_in = shape[1] * shape[2] * shape[3]
        v = base * math.sqrt(2.0 / fan_in)
    elif name.endswith(".weight") and len(shape) == 2:          # affine / map_label (dead: emb == 0)
        v = base * 0.05
    elif name.endswith(".weight"):                              # GroupNorm gamma
        v = 1.0 + 0.2 * base
    elif "affine.bias" in name:                                 # (scale, shift)
        v = 0.15 * base
    elif "conv_log_sigma.bias" in name:
        v = -0.5 + 0.1 * base
    else:                                                       # biases / GN beta
        v = 0.05 * base
    return v.reshape(tuple(shape)).to(torch.float32)


def fill_state(shapes: dict) -> dict:
    """shapes: ordered {key: shape}. Returns {key: tensor} (deterministic, order-dependent)."""
    return {name: fill_tensor(name, shp, k) for k, (name, shp) in enumerate(shapes.items())}


def make_fields(B, cin, cout, H, W, seed=1234):
    """Closed-form smooth 'lo-res' inputs + residual-like targets (no RNG: reproducible anywhere)."""
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float64), torch.arange(W, dtype=torch.float64), indexing="ij")
    xs, ys = [], []
    for b in range(B):
        xc = [torch.sin(0.11 * (c + 1) * xx + 0.07 * (b + 1) * yy + 0.3 * c + 0.001 * seed)
              + 0.5 * torch.cos(0.23 * yy - 0.05 * (c + 2) * xx + b) for c in range(cin)]
        xs.append(torch.stack(xc))
        yc = [0.7 * xc[c % cin] + 0.4 * torch.sin(0.9 * xx + 1.1 * yy + c + 0.5 * b)
              + 0.2 * torch.cos(2.1 * xx - 1.7 * yy + 0.3 * b) for c in range(cout)]
        ys.append(torch.stack(yc))
    return torch.stack(xs).float(), torch.stack(ys).float()


def make_eps(M, B, L):
    i = torch.arange(M * B * L, dtype=torch.float64)
    return (1.2 * torch.sin(1.7 * i + 0.4) + 0.3 * torch.cos(0.31 * i * i)).reshape(M, B, L).float()


def checksum(t: torch.Tensor):
    t = t.detach().double().flatten()
    n = t.numel()
    idx = torch.linspace(0, n - 1, steps=min(n, 16)).long()
    return dict(sum=float(t.sum()), abssum=float(t.abs().sum()), sample=t[idx].tolist(), n=n)
