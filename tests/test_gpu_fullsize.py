"""GPU: size-independent properties at BASELINE.json's FULL configuration (cfg3/cfg5: 4 -> 1 planes, 256 x 256, latent 12, depth-5
U-Net [32..512], seed-42 reference initialisation with the zero-initialised tensors perturbed as in bench.py).  The oracle
cannot run this size in seconds, so the checks are identities the domain offers:
  * samples are independent (per-sample GroupNorm): permuting the batch permutes features / KL bit-for-bit;
  * data-parallel semantics: the gradient of a batch is the mean of its shards' gradients (what the RCCL all-reduce assumes);
  * afCRPS of an ensemble of identical members equals the L1 loss of that member (prob_unet_utils.py:197-224 with x_j == x_k);
  * total == beta0 * recon + beta1 * mean(KL), and gradients are linear in (beta0, beta1);
  * a central finite difference along a random direction matches <grad, direction> (fp32 engine);
  * two identical calls give identical features / KL and the same loss up to the float-atomic order noise of the scalar."""
import numpy as np
import pytest, torch
import probunet_amd as pa

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
NF, MULT = [32, 64, 128, 256, 512], [1, 2, 4, 8, 16]


def _fields(B, seed):
    g = torch.Generator().manual_seed(seed)
    lo = torch.randn(B, 4, 16, 16, generator=g)
    x = torch.nn.functional.interpolate(lo, scale_factor=16.0, mode="nearest") + 0.1 * torch.randn(B, 4, 256, 256, generator=g)
    y = x[:, :1] + 0.5 * torch.randn(B, 1, 256, 256, generator=g)
    return x.to(DEV), y.to(DEV)


def _model(dtype, recon="afcrps", max_batch=4, max_members=3):
    torch.manual_seed(42)
    m = pa.ProbabilisticUNet(4, 1, 12, NF, 32, MULT, 1.0, 1.0, 0.0, dtype=dtype, recon=recon, max_batch=max_batch, max_members=max_members)
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for name, p in m.named_parameters():                      # conv1 / out_conv are zero-initialised in the reference
            if float(p.abs().sum()) == 0.0 and p.dim() == 4:
                p.copy_(torch.randn(p.shape, generator=g) * (0.5 / np.sqrt(p[0].numel())))
    m = m.to(DEV).train(); m.dropout = 0.0
    return m


def _flat_grad(m):
    return torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).flatten() for p in m.parameters()]).clone()


def _eps(M, B, seed=3):
    return torch.randn(M, B, 12, generator=torch.Generator().manual_seed(seed)).to(DEV)


def test_fullsize_batch_permutation_and_determinism():
    m = _model("f16")
    x, y = _fields(4, 1); eps = _eps(3, 4)
    perm = torch.tensor([2, 0, 3, 1], device=DEV)
    with torch.no_grad():
        f1 = m.unet(x).clone(); f2 = m.unet(x).clone(); fp = m.unet(x[perm].contiguous()).clone()
        t1, r1, k1 = m.elbo(x, y, None, M=3, eps=eps)
        t2, r2, k2 = m.elbo(x, y, None, M=3, eps=eps)
        tp, rp, kp = m.elbo(x[perm].contiguous(), y[perm].contiguous(), None, M=3, eps=eps[:, perm].contiguous())
    assert torch.equal(f1, f2) and torch.equal(k1, k2)
    assert abs(r1[0] - r2[0]) <= 1e-5 * abs(r1[0])                      # the loss scalar is a float-atomic sum (ulp-level order noise)
    assert torch.equal(fp, f1[perm]) and torch.equal(kp, k1[perm])
    assert abs(rp[0] - r1[0]) <= 1e-5 * abs(r1[0])                      # batch mean: summation order only
    assert torch.isfinite(f1).all() and float(f1.abs().mean()) > 1e-3


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_fullsize_gradient_of_batch_is_mean_of_shard_gradients(dtype):
    m = _model(dtype)
    x, y = _fields(4, 2); eps = _eps(3, 4)
    def grad(sl):
        m.zero_grad(set_to_none=True)
        t, _, _ = m.elbo(x[sl].contiguous(), y[sl].contiguous(), None, M=3, eps=eps[:, sl].contiguous())
        t.backward()
        return float(t.detach()), _flat_grad(m).double()
    tf, gf = grad(slice(0, 4)); ta, ga = grad(slice(0, 2)); tb, gb = grad(slice(2, 4))
    assert abs(tf - 0.5 * (ta + tb)) <= 2e-3 * abs(tf)
    gm = 0.5 * (ga + gb)
    rel = float((gf - gm).norm() / gm.norm())
    assert rel < (2e-2 if dtype == "f16" else 6e-2), rel          # low-precision activations; identical in exact arithmetic


def test_fullsize_afcrps_of_identical_members_is_l1():
    ma = _model("f16", "afcrps"); ml = _model("f16", "l1")
    x, y = _fields(2, 3)
    z = torch.zeros(3, 2, 12, device=DEV)
    with torch.no_grad():
        ta, ra, ka = ma.elbo(x, y, None, M=3, eps=z)
        tl, rl, kl, kl2 = ml.elbo(x, y, None, eps=z[:1])
        s = ma.sample(x, 3, eps=z)
    assert torch.equal(s[:, 0], s[:, 1]) and torch.equal(s[:, 0], s[:, 2])
    assert abs(ra[0] - rl[0]) <= 1e-5 * abs(rl[0]), (ra[0], rl[0])
    assert torch.allclose(ka, kl, rtol=1e-6, atol=1e-7)


def test_fullsize_total_decomposition_and_beta_linearity():
    m = _model("bf16")
    x, y = _fields(2, 4); eps = _eps(3, 2)
    def run(b0, b1):
        m.beta_0, m.beta_1 = b0, b1
        m.zero_grad(set_to_none=True)
        t, r, k = m.elbo(x, y, None, M=3, eps=eps)
        t.backward()
        return float(t.detach()), r[0], float(k.mean()), _flat_grad(m).double()
    t, r, k, g = run(0.7, 1.3)
    assert abs(t - (0.7 * r + 1.3 * k)) <= 1e-5 * abs(t)
    _, _, _, ga = run(0.7, 0.0); _, _, _, gb = run(0.0, 1.3)
    rel = float((g - (ga + gb)).norm() / g.norm())
    assert rel < 2e-2, rel


def test_fullsize_directional_finite_difference_fp32_engine():
    m = _model("f32", max_batch=1, max_members=2)
    x, y = _fields(1, 5); eps = _eps(2, 1)
    names = [n for n, _ in m.named_parameters()]
    sel = [p for n, p in m.named_parameters() if n.startswith("fcomb.") or n.startswith("unet.out_conv") or n.startswith("unet.dec.128x128_block2.conv1")
           or n.startswith("posterior.conv_mu") or n.startswith("unet.enc.16x16_block1.conv0")]
    assert len(sel) >= 8 and len(names) == 469
    def loss():
        with torch.no_grad():
            return float(m.elbo(x, y, None, M=2, eps=eps)[0])
    m.zero_grad(set_to_none=True)
    t, _, _ = m.elbo(x, y, None, M=2, eps=eps); t.backward()
    g = torch.Generator().manual_seed(9)
    dirs = [torch.randn(p.shape, generator=g).to(DEV) * p.detach().abs().mean() for p in sel]
    dd = sum(float((p.grad.double() * d.double()).sum()) for p, d in zip(sel, dirs))
    h = 2e-2
    with torch.no_grad():
        for p, d in zip(sel, dirs): p.add_(h * d)
        lp = loss()
        for p, d in zip(sel, dirs): p.sub_(2 * h * d)
        lm = loss()
        for p, d in zip(sel, dirs): p.add_(h * d)
    fd = (lp - lm) / (2 * h)
    assert abs(fd - dd) <= 3e-2 * abs(dd) + 1e-6, (fd, dd)
