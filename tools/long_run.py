import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import probunet_amd as pa
dev = torch.device('cuda', 0)
dtype = sys.argv[1] if len(sys.argv) > 1 else 'f16'
cfg = dict(bench.CFG3)
m = bench.build_model(cfg, dtype, dev); m.sync_scalars = False
opt = pa.FlatAdamW(m, lr=1e-4)
x, y = bench.synthetic_fields(32, 4, 1, 256, 256, 1234, dev)
torch.manual_seed(1234)
for step in range(301):
    loss, rl, kl = m.elbo(x, y, None, M=5)
    opt.zero_grad(); loss.backward(); opt.step()
    if step % 25 == 0 or step < 5:
        p, q = m.prior_latent_space, m.posterior_latent_space
        print(dtype, step, 'loss %.4f recon %.4f kl mean %.4f max %.3f | prior sig [%.3g, %.3g] mu|max %.3f | post sig [%.3g, %.3g] mu|max %.3f | flag %s' % (
            float(loss), float(rl[0]), float(kl.mean()), float(kl.max()), float(p.base_dist.scale.min()), float(p.base_dist.scale.max()), float(p.base_dist.loc.abs().max()),
            float(q.base_dist.scale.min()), float(q.base_dist.scale.max()), float(q.base_dist.loc.abs().max()), float(m._last_scalars[6])), flush=True)
