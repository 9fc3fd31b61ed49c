"""Generate tests/golden/*.npz|json by IMPORTING THE REFERENCE (this container only).

Run:  python tools/make_golden.py            (needs /root/reference; never runs on the GPU box)

What is captured: for each tiny config, with closed-form filled parameters (tests/filler.py),
closed-form inputs and explicit reparameterisation noise eps:
  unet features, prior/posterior (mu, log_sigma), fcomb outputs, KL[B], afCRPS / L1 loss, total loss and
  every parameter gradient of the afCRPS-ELBO assembled line by line from LIVE reference functions
  (prob_unet.py:283-314 is commented out at HEAD; its body is reproduced here as a call sequence on the
  imported modules: unet, prior, posterior, fcomb, afcrps_loss, kl.kl_divergence).
Also: state_dict keys/shapes, seed-42 init checksums, one AdamW step, latent-grid decode, forward(training=False).
Only data (numbers) is written; no reference source text travels.
"""
import json, os, sys, types

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/src")
stub = types.ModuleType("pytorch_msssim")          # absent offline (uv.lock:786-794); MS-SSIM path is unpinned
def _no_msssim(*a, **k):
    raise RuntimeError("pytorch_msssim is not available offline")
stub.ms_ssim = _no_msssim
sys.modules["pytorch_msssim"] = stub

import numpy as np
import torch
from torch.distributions import kl
import prob_unet as ref                            # noqa: E402  (the reference)
import prob_unet_utils as ref_utils                # noqa: E402
from tests.filler import fill_tensor, make_fields, make_eps, checksum  # noqa: E402

torch.set_num_threads(8)
OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)

CONFIGS = {
    # name: (cin, cout, L, num_filters, mc, mult, H, B, M)
    "tiny22": (2, 2, 6, [8, 16, 32], 8, [1, 2, 4], 32, 2, 3),
    "tiny41": (4, 1, 6, [8, 16, 32], 8, [1, 2, 4], 32, 2, 2),
    "mid11": (1, 1, 6, [32, 64, 128], 32, [1, 2, 4], 32, 2, 2),   # cfg1 widths (GN with 6 ch/group) at 32x32
}


def build(cin, cout, L, nf, mc, mult):
    return ref.ProbabilisticUNet(cin, cout, L, nf, mc, mult, 1.0, 1.0, 0.0)


def fill_reference(model):
    sd = model.state_dict()
    new = {}
    for k, (name, v) in enumerate(sd.items()):
        new[name] = fill_tensor(name, tuple(v.shape), k)
    model.load_state_dict(new)
    return new


def pad_target(y, cin):
    B, c, H, W = y.shape
    if c == cin:
        return y
    return torch.cat([y, torch.zeros(B, cin - c, H, W)], dim=1)


def run_config(name, spec):
    cin, cout, L, nf, mc, mult, H, B, M = spec
    torch.manual_seed(42)
    model = build(cin, cout, L, nf, mc, mult)
    sd_init = {k: v.clone() for k, v in model.state_dict().items()}
    meta = dict(config=dict(input_channels=cin, num_classes=cout, latent_dim=L, num_filters=nf, model_channels=mc,
                            channel_mult=mult, H=H, W=H, B=B, M=M),
                keys=[[k, list(v.shape)] for k, v in sd_init.items()],
                init_seed42={k: checksum(v) for k, v in sd_init.items()})
    filled = fill_reference(model)
    x, y = make_fields(B, cin, cout, H, H)
    eps = make_eps(M, B, L)
    yp = pad_target(y, cin)
    arrays = {}

    def elbo_ref(train_mode):
        model.train(train_mode)
        for m in model.modules():                                   # dropout off in both modes
            if hasattr(m, "dropout") and isinstance(getattr(m, "dropout"), float):
                m.dropout = 0.0
        model.zero_grad()
        feat = model.unet(x)
        p = model.prior(x)
        q = model.posterior(x, yp)
        preds = []
        for m_ in range(M):
            z = q.base_dist.loc + q.base_dist.scale * eps[m_]        # == rsample() with captured noise
            preds.append(model.fcomb(feat, z))
        ens = torch.stack(preds, dim=1)
        crps = ref_utils.afcrps_loss(ens, y, alpha=0.95)
        kld = kl.kl_divergence(q, p)
        total = 0.7 * crps + 1.3 * kld.mean()
        return feat, p, q, ens, crps, kld, total

    feat, p, q, ens, crps, kld, total = elbo_ref(True)
    total.backward()
    arrays.update(x=x.numpy(), y=y.numpy(), eps=eps.numpy(), feat=feat.detach().numpy(),
                  mu_p=p.base_dist.loc.detach().numpy(), sig_p=p.base_dist.scale.detach().numpy(),
                  mu_q=q.base_dist.loc.detach().numpy(), sig_q=q.base_dist.scale.detach().numpy(),
                  preds=ens.detach().numpy(), crps=np.float64(crps.item()), kl=kld.detach().numpy(),
                  total=np.float64(total.item()), beta0=0.7, beta1=1.3, alpha=0.95)
    gsum = {}
    for k, prm in model.named_parameters():
        g = prm.grad if prm.grad is not None else torch.zeros_like(prm)
        if k == "posterior.encoder.0.weight":
            g = g[:, : cin + cout]
        gsum[k] = checksum(g)
        if g.numel() <= 20000 or name.startswith("tiny"):
            arrays["grad::" + k] = g.numpy().copy()
    meta["grad_checksums"] = gsum

    # L1 ELBO (prob_unet.py:341-379), M = 1, with beta2 KL(q || N(0,I))
    model.zero_grad()
    z = q.base_dist.loc.detach() * 0  # placeholder to keep names explicit
    feat1 = model.unet(x); p1 = model.prior(x); q1 = model.posterior(x, yp)
    z1 = q1.base_dist.loc + q1.base_dist.scale * eps[0]
    out1 = model.fcomb(feat1, z1)
    l1 = torch.nn.L1Loss()(out1, y)
    from torch.distributions import Normal, Independent
    std = Independent(Normal(torch.zeros_like(q1.base_dist.loc), torch.ones_like(q1.base_dist.scale)), 1)
    kl2 = kl.kl_divergence(q1, std)
    tot1 = 0.7 * l1 + 1.3 * kl.kl_divergence(q1, p1).mean() + 0.2 * kl2.mean()
    arrays.update(l1=np.float64(l1.item()), kl2=kl2.detach().numpy(), total_l1=np.float64(tot1.item()), beta2=0.2)

    # forward(training=False) == prior sample (prob_unet.py:218-223); eval mode
    model.eval()
    with torch.no_grad():
        featE = model.unet(x)
        pE = model.prior(x)
        zE = pE.base_dist.loc + pE.base_dist.scale * eps[0]
        outE = model.fcomb(featE, zE)
        arrays.update(sample_prior=outE.numpy())
        # latent grid decode with expanded (stride-0) features (latent_exploration.py:119-129)
        K = 5
        grid = torch.stack([torch.linspace(-2, 2, K)] * L, dim=1)           # [K, L]
        feat_rep = featE[:1].expand(K, -1, -1, -1)
        arrays.update(grid_z=grid.numpy(), grid_out=model.fcomb(feat_rep, grid).numpy())
        # Fcomb.tile == broadcast
        t = model.fcomb.tile(torch.arange(6.).reshape(2, 3, 1), 2, 4)
        arrays.update(tile_out=t.numpy())

    # one AdamW step (train_prob_unet_model.py:133-141, main.py:103), dropout off
    model.load_state_dict(filled)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4)
    feat, p, q, ens, crps, kld, total = elbo_ref(True)
    opt.zero_grad(); total.backward(); opt.step()
    meta["after_adamw"] = {k: checksum(v if k != "posterior.encoder.0.weight" else v[:, : cin + cout])
                           for k, v in model.state_dict().items() if "resample_filter" not in k}
    meta["filled_checksums"] = {k: checksum(v if k != "posterior.encoder.0.weight" else v[:, : cin + cout])
                                for k, v in filled.items()}
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **arrays)
    with open(os.path.join(OUT, f"{name}.json"), "w") as f:
        json.dump(meta, f)
    print(name, "total", float(total), "crps", float(crps), "feat|max|", float(feat.abs().max()))


def run_wmse():
    """WMSE half + the loss combination + the data_range inference of wmse_ms_ssim_loss (prob_unet_utils.py:270-305), with the
    absent third-party ms_ssim replaced by a recording constant (its own arithmetic stays unpinned)."""
    seen = {}
    def fake_ms_ssim(pred, target, data_range=None, size_average=True, win_size=None):
        seen.update(data_range=float(data_range), win_size=win_size, size_average=size_average)
        return torch.tensor(0.25)
    ref_utils.ms_ssim = fake_ms_ssim
    out = {}
    for tag, (B, C, H, lam) in {"a": (2, 1, 24, 1.0), "b": (3, 2, 16, 0.3)}.items():
        x, y = make_fields(B, C, C, H, H, seed=77)
        pred = (0.8 * y + 0.3 * x).requires_grad_(True)
        tgt = 3.0 * y + 1.0                                     # wide range so that min(alpha e^{beta y}, 1) is exercised
        loss, wmse, msl = ref_utils.wmse_ms_ssim_loss(pred, tgt, alpha=0.5, beta=0.4, lam=lam, return_components=True)
        loss.backward()
        out[tag] = dict(B=B, C=C, H=H, lam=lam, alpha=0.5, beta=0.4, loss=float(loss), wmse=float(wmse), msssim_loss=float(msl),
                        data_range=seen["data_range"], win_size=seen["win_size"], grad=checksum(pred.grad))
        ens = torch.stack([pred.detach(), pred.detach() * 0.5], dim=1)          # 5-D input: the ensemble mean is used (:283-284)
        l5, w5, _ = ref_utils.wmse_ms_ssim_loss(ens, tgt, alpha=0.5, beta=0.4, lam=lam, return_components=True)
        out[tag]["wmse_ens_mean"] = float(w5)
    with open(os.path.join(OUT, "wmse.json"), "w") as f:
        json.dump(out, f)
    print("wmse", out["a"]["wmse"], out["b"]["loss"])


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "wmse":
        run_wmse()
    else:
        for n, s in CONFIGS.items():
            run_config(n, s)
        run_wmse()
