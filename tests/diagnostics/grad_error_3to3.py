import sys, os, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import probunet_amd as pa
from oracle import probunet_oracle as O
from tests.helpers import filled_params
from tests.filler import make_eps
cfg = O.Config(3, 3, 32, [32, 64, 128, 256], 32, [1, 2, 4, 8]); P = filled_params(cfg); B = 1
yy, xx = torch.meshgrid(torch.arange(64.), torch.arange(128.), indexing="ij")
x = torch.stack([torch.stack([torch.sin(0.13 * (c + 1) * xx + 0.09 * (b + 2) * yy) + 0.3 * torch.cos(0.31 * yy * (c + 1)) for c in range(3)]) for b in range(B)])
y = 0.6 * x + 0.3 * torch.sin(1.3 * xx - 0.7 * yy)[None, None]; eps = make_eps(4, B, 32)
m = pa.ProbabilisticUNet(3, 3, 32, [32, 64, 128, 256], 32, [1, 2, 4, 8], 0.7, 1.3, 0.0, dtype="f32", init=False)
m.load_state_dict(P); m = m.to("cuda:0").train(); m.dropout = 0.0
total, _, _ = m.elbo(x.cuda(), y.cuda(), None, M=4, eps=eps.cuda()); total.backward()
_, g32 = O.elbo_with_grads(P, cfg, x, y, eps, beta0=0.7, beta1=1.3)
_, g64 = O.elbo_with_grads({k: v.double() for k, v in P.items()}, cfg, x.double(), y.double(), eps.double(), beta0=0.7, beta1=1.3)
gh = {k: p.grad.detach().cpu().double() for k, p in m.named_parameters()}
rows = [(k, float((gh[k] - g64[k]).norm() / (g64[k].norm() + 1e-30)), float((g32[k].double() - g64[k]).norm() / (g64[k].norm() + 1e-30))) for k in g64 if float(g64[k].norm()) > 1e-7]
rows.sort(key=lambda r: -r[1])
for r in rows[:8]: print("%-46s hip %.2e torch32 %.2e" % r)
print("median hip %.2e torch32 %.2e ; max hip %.2e torch32 %.2e" % (statistics.median(r[1] for r in rows), statistics.median(r[2] for r in rows), max(r[1] for r in rows), max(r[2] for r in rows)))
