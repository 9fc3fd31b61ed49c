import sys, os; sys.path.insert(0, '/root/repo')
import torch
import probunet_amd as pa
dev = torch.device("cuda", 0)
def data():
    g = torch.Generator(device="cpu").manual_seed(0)
    N = 64
    lo = torch.randn(N, 4, 16, 16, generator=g)
    big = torch.nn.functional.interpolate(lo, scale_factor=16.0, mode="bicubic", align_corners=False)
    fine = torch.nn.functional.interpolate(torch.randn(N, 4, 64, 64, generator=g), scale_factor=4.0, mode="bilinear", align_corners=False)
    hr = (big + 0.35 * fine * (1.0 + 0.5 * torch.tanh(big))).to(dev)
    tr = pa.data.ClimExTransform(lowres_scale=4).fit(hr)
    it = tr(hr[:32])
    return it["inputs"], it["targets"][:, :1].contiguous()
x, y = data()
for dtype in ("f16", "bf16"):
    torch.manual_seed(42)
    m = pa.ProbabilisticUNet(4, 1, 12, [32, 64, 128, 256, 512], 32, [1, 2, 4, 8, 16], 1.0, 0.0, 0.0, dtype=dtype, max_batch=32, max_members=5).to(dev).train()
    opt = pa.FlatAdamW(m, lr=2e-4)
    m.sync_scalars = False
    for step in range(30):
        m.beta_1 = 0.0 if step < 21 else 0.25
        loss, rl, kl = m.elbo(x, y, None, M=5)
        opt.zero_grad(); loss.backward()
        bad = [n for n, p in m.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
        gmax = max(float(p.grad.abs().max()) for p in m.parameters() if p.grad is not None and torch.isfinite(p.grad).all())
        if step >= 19 or bad:
            print(dtype, step, 'beta1', m.beta_1, 'loss', float(loss), 'recon', float(rl[0]), 'klmean', float(kl.mean()), 'klmax', float(kl.max()), 'gmax', gmax, 'nonfinite', len(bad), bad[:4], flush=True)
        if bad: break
        opt.step()
    # ---- at the failing state: which part produces the non-finite values?
    with torch.no_grad():
        p = m.prior(x); q = m.posterior(x, y)
    print(dtype, 'prior  mu|max', float(p.base_dist.loc.abs().max()), 'sigma min/max', float(p.base_dist.scale.min()), float(p.base_dist.scale.max()))
    print(dtype, 'post   mu|max', float(q.base_dist.loc.abs().max()), 'sigma min/max', float(q.base_dist.scale.min()), float(q.base_dist.scale.max()))
    opt.zero_grad()
    p = m.prior(x); q = m.posterior(x, y)
    kl = torch.distributions.kl.kl_divergence(q, p).mean()
    kl.backward()
    bad2 = [n for n, pp in m.named_parameters() if pp.grad is not None and not torch.isfinite(pp.grad).all()]
    print(dtype, 'sub-module path: kl', float(kl.detach()), 'nonfinite', len(bad2), bad2[:3])
    # engine path with beta0 = 0 (KL only) and with beta1 = 0 (recon only)
    for b0, b1 in ((0.0, 0.25), (1.0, 0.0), (1.0, 0.25)):
        m.beta_0, m.beta_1 = b0, b1
        opt.zero_grad()
        loss, rl, klv = m.elbo(x, y, None, M=5); loss.backward()
        bad3 = [n for n, pp in m.named_parameters() if pp.grad is not None and not torch.isfinite(pp.grad).all()]
        print(dtype, 'engine beta0', b0, 'beta1', b1, 'nonfinite', len(bad3), bad3[:3], flush=True)
    m.beta_0 = 1.0
