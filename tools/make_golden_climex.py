"""Generate tests/golden/climex.npz by IMPORTING THE REFERENCE's src/climex_utils.py (this container only; never on the GPU box).

climex_utils imports xarray / dask / dask.distributed / bottleneck / cftime / cartopy at module level; none of them is
installed here and none is touched by the functions captured below, so they are replaced by empty in-memory modules
(sys.modules stubs, nothing is written anywhere).  The dataset constructor (NetCDF I/O, dask Client) is NOT run: an
instance is made with object.__new__ and given exactly the attributes the captured methods read
(hr, lowres_scale, epsilon, type, lrstats, timestamps, timestamps_float).

Captured (rows f3 / f4 of SURVEY.md §8): softplus, softplus_inv (climex_utils.py:36-46), compute_stats (:255-264),
__getitem__ of the "lrinterp_to_residuals" and "lrinterp_to_hr" pipelines (:197-250), invstand_residual and residual_to_hr
(:270-285) for both pipeline kinds.  Only numbers are written.
"""
import os, sys, types

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/src")


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


_stub("dask")
_stub("dask.distributed", Client=object)
_stub("xarray")
_stub("bottleneck")
_stub("cftime")
_cart = _stub("cartopy")
_cart.crs = _stub("cartopy.crs")

import numpy as np          # noqa: E402
import torch                # noqa: E402
import climex_utils as cu   # noqa: E402  (the reference)

OUT = os.path.join(ROOT, "tests", "golden")


def synthetic_hr(N, C, H, W):
    """Closed-form ClimEx-like array [time, var, H, W]: precipitation-like (>= 0, skewed), tasmin, tasmax (> tasmin)."""
    t = torch.arange(N, dtype=torch.float64).view(N, 1, 1)
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float64), torch.arange(W, dtype=torch.float64), indexing="ij")
    pr = torch.relu(torch.sin(0.31 * xx + 0.17 * yy + 0.9 * t) + 0.3 * torch.cos(0.05 * xx * yy + t)) ** 2 * 6.0
    tmin = 5.0 + 8.0 * torch.sin(0.11 * xx - 0.07 * yy + 0.5 * t) + 0.4 * t
    tmax = tmin + 2.0 + 3.0 * torch.cos(0.23 * yy + 0.3 * t) ** 2
    return torch.stack([pr, tmin, tmax], dim=1)[:, :C].float().contiguous()


def make_dataset(hr, kind, k):
    ds = object.__new__(cu.climex2torch)
    ds.hr = hr.clone()
    ds.lowres_scale = k
    ds.epsilon = 1e-10
    ds.type = kind
    ds.lrstats = None
    ds.timestamps = torch.linspace(-1.0, 1.0, hr.shape[0])
    ds.timestamps_float = np.arange(hr.shape[0], dtype=np.float64)
    return ds


def main():
    out = {}
    N, C, H, W, k = 7, 3, 32, 32, 4
    hr = synthetic_hr(N, C, H, W)
    out["hr"] = hr.numpy(); out["k"] = np.int64(k)
    # softplus / softplus_inv (in place in the reference: hand them clones)
    v = torch.cat([torch.linspace(-12.0, 30.0, 85), torch.tensor([19.999, 20.0, 20.001, 0.0, 1e-6, 25.0])]).float()
    out["sp_in"] = v.numpy()
    out["sp_out"] = cu.softplus(v.clone()).numpy()
    out["sp_out_c0"] = cu.softplus(v.clone(), c=0.).numpy()
    vp = torch.cat([torch.linspace(1e-3, 30.0, 60), torch.tensor([19.999, 20.0, 20.001])]).float()
    out["spinv_in"] = vp.numpy()
    out["spinv_out"] = cu.softplus_inv(vp.clone()).numpy()
    out["spinv_out_c0"] = cu.softplus_inv(vp.clone(), c=0.).numpy()
    # statistics + item transform, residual pipeline
    ds = make_dataset(hr, "lrinterp_to_residuals", k)
    (m_lr, s_lr), (m_hr, s_hr) = ds.compute_stats()
    out.update(mean_lr=m_lr.numpy(), std_lr=s_lr.numpy(), mean_hr=m_hr.numpy(), std_hr=s_hr.numpy())
    items = [ds[i] for i in range(3)]                      # first access computes ds.lrstats lazily, like the DataLoader does
    for key in ("inputs", "targets", "lrinterp", "lr", "hr"):
        out["item_" + key] = torch.stack([it[key] for it in items]).numpy()
    out["item_timestamps"] = torch.stack([it["timestamps"] for it in items]).numpy()
    resid = torch.stack([it["targets"] for it in items])
    lrint = torch.stack([it["lrinterp"] for it in items])
    out["invstand_residual"] = ds.invstand_residual(resid).numpy()
    out["residual_to_hr"] = ds.residual_to_hr(resid, lrint).numpy()          # == hr items, up to rounding
    # a "predicted" residual that is not the exact one, 5-D [B, n, C, H, W] as the sampling loops stack them
    g = torch.Generator().manual_seed(3)
    pred = resid.unsqueeze(1) + 0.25 * torch.randn(3, 4, C, H, W, generator=g)
    out["pred_residual"] = pred.numpy()
    out["pred_to_hr"] = torch.stack([ds.residual_to_hr(pred[:, j], lrint) for j in range(4)], dim=1).numpy()
    # the *_to_hr pipeline kind (invstand adds the mean back)
    ds2 = make_dataset(hr, "lrinterp_to_hr", k)
    it2 = [ds2[i] for i in range(2)]
    out["hrkind_inputs"] = torch.stack([it["inputs"] for it in it2]).numpy()
    out["hrkind_targets"] = torch.stack([it["targets"] for it in it2]).numpy()
    out["hrkind_invstand"] = ds2.invstand_residual(torch.stack([it["targets"] for it in it2])).numpy()
    np.savez_compressed(os.path.join(OUT, "climex.npz"), **out)
    print("wrote", os.path.join(OUT, "climex.npz"), {k_: np.asarray(v_).shape for k_, v_ in out.items()})


if __name__ == "__main__":
    main()
