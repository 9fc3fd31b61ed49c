"""Shared test helpers: golden loading + filled parameter construction in the reference's key order."""
import json, os
import numpy as np
import torch
from tests.filler import fill_tensor
from oracle import probunet_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GOLDEN_NAMES = ["tiny22", "tiny41", "mid11"]


def load_golden(name):
    with open(os.path.join(GOLDEN, f"{name}.json")) as f:
        meta = json.load(f)
    arr = dict(np.load(os.path.join(GOLDEN, f"{name}.npz")))
    return meta, arr


def oracle_cfg(c) -> "O.Config":
    return O.Config(c["input_channels"], c["num_classes"], c["latent_dim"], list(c["num_filters"]),
                    c["model_channels"], list(c["channel_mult"]))


def filled_params(cfg: "O.Config"):
    """Parameters filled exactly as tools/make_golden.py filled the reference model.
    The reference posterior stem has 2*Cin planes (prob_unet.py:27-28); the filler is applied at that
    shape and sliced to Cin+Cout planes (zero-padded-target equivalence)."""
    shapes = O.param_shapes(cfg)
    out = {}
    for k, (name, shp) in enumerate(shapes.items()):
        if name == "posterior.encoder.0.weight":
            full = (shp[0], 2 * cfg.input_channels, 3, 3)
            out[name] = fill_tensor(name, full, k)[:, : shp[1]].contiguous()
        else:
            out[name] = fill_tensor(name, shp, k)
    return out


def random_params(cfg, seed):
    """Random (non closed-form) parameters: kaiming-scaled normals, GroupNorm affine around (1, 0), non-zero everything."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, shp in O.param_shapes(cfg).items():
        if name.endswith("resample_filter"):
            out[name] = torch.full(tuple(shp), 0.25)
        elif len(shp) == 4:
            fan_in = shp[1] * shp[2] * shp[3]
            out[name] = torch.randn(shp, generator=g) * (1.0 / fan_in) ** 0.5
        elif len(shp) == 2:
            out[name] = torch.randn(shp, generator=g) * 0.05
        elif name.endswith("norm0.weight") or name.endswith("norm1.weight") or name.endswith("out_norm.weight"):
            out[name] = 1.0 + 0.1 * torch.randn(shp, generator=g)
        else:
            out[name] = 0.05 * torch.randn(shp, generator=g)
    return out


def t(a):
    return torch.from_numpy(np.asarray(a))


def assert_close(a, b, rtol=1e-3, atol=1e-5, what=""):
    """|a-b| <= atol*max(1, max|b|) + rtol*|b|  (north_star: rtol 1e-3 / atol 1e-5, fp32).
    atol is scaled by the tensor's max-abs: after ~60 fp32 layers elements that cross zero carry an
    absolute rounding error proportional to the tensor's scale, not to their own magnitude."""
    a = torch.as_tensor(a).double(); b = torch.as_tensor(b).double()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    err = (a - b).abs()
    tol = atol * max(1.0, float(b.abs().max())) + rtol * b.abs()
    if not bool((err <= tol).all()):
        i = int((err - tol).argmax())
        raise AssertionError(f"{what}: max|err|={float(err.max()):.3e} worst a={float(a.flatten()[i]):.6e} "
                             f"b={float(b.flatten()[i]):.6e} frac_bad={float((err > tol).double().mean()):.3e}")
