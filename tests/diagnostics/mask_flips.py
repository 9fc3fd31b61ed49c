import torch, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from oracle import probunet_oracle as O
from tests.helpers import filled_params
from tests.filler import make_fields, make_eps
cfg = O.Config(1, 1, 6, [32, 64, 128], 32, [1, 2, 4])
P = filled_params(cfg)
x, y = make_fields(4, 1, 1, 64, 64)
def run(dtype, dev):
    Pg = {k: v.detach().clone().to(dev, dtype).detach().requires_grad_(True) if v.is_floating_point() and 'resample' not in k else v.to(dev) for k, v in P.items()}
    mu_p, ls_p = O.gauss_forward(Pg, cfg, "prior", x.to(dev, dtype))
    mu_q, ls_q = O.gauss_forward(Pg, cfg, "posterior", x.to(dev, dtype), y.to(dev, dtype))
    O.kl_diag_normal(mu_q, O.sigma_of(ls_q), mu_p, O.sigma_of(ls_p)).mean().backward()
    return {k: v.grad.detach().cpu().double() for k, v in Pg.items() if torch.is_tensor(v) and v.requires_grad and v.grad is not None}
g64 = run(torch.float64, "cpu"); g32c = run(torch.float32, "cpu"); g32g = run(torch.float32, "cuda:0")
for k in ["prior.encoder.18.weight","prior.encoder.11.weight","prior.encoder.9.weight","prior.encoder.7.weight","prior.encoder.4.weight","prior.encoder.0.weight"]:
    r = lambda a: float((a[k]-g64[k]).norm()/g64[k].norm())
    print("%-28s torch32cpu %.2e torch32gpu %.2e" % (k, r(g32c), r(g32g)))
# count near-zero pre-activations / near ties in the fp64 forward
import torch.nn.functional as F
xx = x.double()
P64 = {k: v.double() for k, v in P.items()}
h = xx
for lv, idxs in enumerate(O.gauss_conv_indices(3)):
    if lv: 
        # near ties in 2x2 windows
        u = F.unfold(h, 2, stride=2).reshape(h.shape[0], h.shape[1], 4, -1)
        top2 = u.topk(2, dim=2).values
        gap = (top2[:, :, 0] - top2[:, :, 1]).abs()
        pos = top2[:, :, 0] > 0
        print("level", lv, "pool windows:", gap.numel(), "near-ties(<1e-6, max>0):", int(((gap < 1e-6) & pos).sum()), "exact ties", int(((gap == 0) & pos).sum()))
        h = F.max_pool2d(h, 2)
    for idx in idxs:
        pre = F.conv2d(h, P64[f"prior.encoder.{idx}.weight"], P64[f"prior.encoder.{idx}.bias"], padding=1)
        print("  conv", idx, "elements", pre.numel(), "|pre|<1e-6:", int((pre.abs() < 1e-6).sum()), "|pre|<1e-4:", int((pre.abs() < 1e-4).sum()))
        h = torch.relu(pre)
