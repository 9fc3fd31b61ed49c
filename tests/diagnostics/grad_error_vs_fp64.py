import torch, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
import probunet_amd as pa
from oracle import probunet_oracle as O
from tests.helpers import filled_params
from tests.filler import make_fields, make_eps
cfg = O.Config(1, 1, 6, [32, 64, 128], 32, [1, 2, 4])
m = pa.ProbabilisticUNet(1, 1, 6, [32, 64, 128], 32, [1, 2, 4], 0.7, 1.3, 0.2, dtype="f32", init=False)
P = filled_params(cfg); m.load_state_dict(P); m = m.to("cuda:0").train(); m.dropout = 0.0
x, y = make_fields(4, 1, 1, 64, 64); eps = make_eps(5, 4, 6)
total, recon, kl = m.elbo(x.cuda(), y.cuda(), None, M=5, eps=eps.cuda()); total.backward()
r32, g32 = O.elbo_with_grads(P, cfg, x, y, eps, beta0=0.7, beta1=1.3)
P64 = {k: v.double() for k, v in P.items()}
r64, g64 = O.elbo_with_grads(P64, cfg, x.double(), y.double(), eps.double(), beta0=0.7, beta1=1.3)
print("total hip %.8f o32 %.8f o64 %.8f" % (float(total), float(r32["total"]), float(r64["total"])))
gh = {k: p.grad.detach().cpu().double() for k, p in m.named_parameters()}
rows = []
for k in g64:
    ref = g64[k]; n = float(ref.norm()) + 1e-30
    rows.append((k, float((gh[k] - ref).norm()) / n, float((g32[k].double() - ref).norm()) / n, n))
rows.sort(key=lambda r: -r[1])
for k, eh, eo, n in rows[:25]:
    print("%-48s hip_rel %.2e  torch32_rel %.2e  norm %.2e" % (k, eh, eo, n))
import statistics
print("median hip", statistics.median(r[1] for r in rows), "median torch32", statistics.median(r[2] for r in rows))
