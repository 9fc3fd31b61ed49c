"""Race screen: the cfg3 ELBO forward + backward (B = 32, side streams on, dropout on with a fixed seed) repeated N times from the same
parameters, inputs, noise and dropout seed.  The path is order-fixed except for its float atomics (loss scalars, GroupNorm / Fcomb
parameter gradients, the latent-bias gradient that feeds the two latent encoders: DESIGN.md section 5), so repetitions agree to
rounding noise, not bit for bit: the screen fails when a repetition's gradient differs from the reference repetition by more than
1e-4 of the gradient norm (a race shows up as whole tiles of wrong values, orders of magnitude above that).
usage: python tools/determinism_soak.py [N=40] [dtype=f16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dtype = sys.argv[2] if len(sys.argv) > 2 else "f16"
dev = torch.device("cuda", 0)
cfg = dict(bench.CFG3)
m = bench.build_model(cfg, dtype, dev); m.sync_scalars = False
x, y = bench.synthetic_fields(cfg["batch"], cfg["input_channels"], cfg["num_classes"], cfg["H"], cfg["W"], 5, dev)
eps = torch.randn(cfg["M"], cfg["batch"], cfg["latent_dim"], device=dev)
ref = None; bad = 0; first = None; worst = 0.0
for it in range(N + 1):
    m._step = 0                                        # same dropout seed every repetition
    torch.manual_seed(99)
    for p in m.parameters():
        p.grad = None
    loss = m.elbo(x, y, None, M=cfg["M"], eps=eps)[0]
    loss.backward()
    g = torch.cat([p.grad.flatten() for p in m.parameters()])
    cur = (g.clone(), loss.detach().clone())
    if first is None:
        first = cur                                    # the very first call also builds the plan: compared separately below
    elif ref is None:
        ref = cur
    else:
        rel = float((cur[0].double() - ref[0].double()).norm() / ref[0].double().norm())
        worst = max(worst, rel)
        if rel > 1e-4 or not torch.isfinite(cur[0]).all():
            bad += 1
            nzr = int(((cur[0] - ref[0]) != 0).sum())
            print(f"repetition {it}: rel {rel:.3e}, differs in {nzr} gradient elements (max |dg| {float((cur[0] - ref[0]).abs().max()):.3e}, loss {float(cur[1])} vs {float(ref[1])})", flush=True)
torch.cuda.synchronize()
d0 = float((first[0] - ref[0]).abs().max())
nz = int(((first[0] - ref[0]) != 0).sum())
print(f"first call vs second: max |dg| {d0:.3e} in {nz} of {first[0].numel()} gradient elements, loss {float(first[1])} vs {float(ref[1])}")
print(f"{dtype}: {N} repetitions, {bad} beyond 1e-4 of the gradient norm (worst {worst:.3e}); loss {float(ref[1]):.6f}, |g|max {float(ref[0].abs().max()):.4e}, finite {bool(torch.isfinite(ref[0]).all())}")
sys.exit(1 if bad else 0)
