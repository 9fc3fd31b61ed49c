"""GPU: the BASELINE.json configurations that no other test runs at their stated size.

  cfg2  4 -> 1 planes, 128 x 128, depth-4 U-Net [32, 64, 128, 256], latent 6, bf16 (and the fp32 parity engine), B = 2, M = 2:
        the whole ELBO forward + backward against the CPU oracle (about a second of CPU work).
  cfg5  256 x 256, 64 prior samples per lo-res input: sample(x, 64) against fcomb(unet(x), z_i) member by member (the
        latent_exploration.py:119-129 pattern), graph replay against the eager launch sequence, and the same call on a
        64 x 64 crop configuration against the oracle.
  cfg3  at its stated batch, B = 32 per GPU (the other full-size tests run B <= 4): gradient of the batch == mean of the two
        16-pair shard gradients (what the data-parallel all-reduce assumes; exercises the batch-dependent split-K / grid sizing),
        and call-twice determinism of everything that is not a float-atomic sum.
Tolerances are written at each assertion.
"""
import numpy as np
import pytest, torch
import probunet_amd as pa
from oracle import probunet_oracle as O
from tests.helpers import filled_params, random_params, assert_close
from tests.filler import make_fields, make_eps

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
NF3, MULT3 = [32, 64, 128, 256, 512], [1, 2, 4, 8, 16]


# ------------------------------------------------------------------------------------------------ cfg2
CFG2 = O.Config(4, 1, 6, [32, 64, 128, 256], 32, [1, 2, 4, 8])


# Random (kaiming-scaled) parameters: the closed-form filler of the golden tests builds a network that amplifies rounding ~1e4 x
# at this depth (torch fp32 itself is 2e-3 away from fp64 on it), which says nothing about the engine.
P2 = random_params(CFG2, seed=21)


def _cfg2_model(dtype):
    m = pa.ProbabilisticUNet(4, 1, 6, [32, 64, 128, 256], 32, [1, 2, 4, 8], 0.7, 1.3, 0.0, dtype=dtype, init=False)
    m.load_state_dict(P2)
    m = m.to(DEV).train(); m.dropout = 0.0
    return m


@pytest.fixture(scope="module")
def cfg2_oracle():
    x, y = make_fields(2, 4, 1, 128, 128, seed=22); eps = make_eps(2, 2, 6)
    r, g = O.elbo_with_grads(P2, CFG2, x, y, eps, beta0=0.7, beta1=1.3)
    return x, y, eps, r, g


def test_cfg2_fp32_engine_matches_oracle(cfg2_oracle):
    x, y, eps, r, og = cfg2_oracle
    m = _cfg2_model("f32")
    total, recon, kl = m.elbo(x.to(DEV), y.to(DEV), None, M=2, eps=eps.to(DEV))
    total.backward()
    assert_close(total.detach().cpu(), r["total"], what="total")          # rtol 1e-3 / atol 1e-5 (north_star)
    assert_close(recon[0], r["recon"], what="afcrps"); assert_close(kl.cpu(), r["kl"], what="kl")
    rels = []
    for k, p in m.named_parameters():
        v = og[k].double()
        if float(v.norm()) < 1e-9: continue
        rels.append(float((p.grad.detach().cpu().double() - v).norm() / v.norm()))
    # per-tensor relative L2: fp32 summation-order noise through 4 levels; isolated ReLU / max-pool ties move single tensors
    assert float(np.median(rels)) < 1e-3 and max(rels) < 2e-2, (float(np.median(rels)), max(rels))


def test_cfg2_bf16_engine_tracks_oracle(cfg2_oracle):
    """BASELINE config 2 names bf16: bf16 operands (8 significant bits), fp32 accumulate, fp32 statistics and losses."""
    x, y, eps, r, og = cfg2_oracle
    m = _cfg2_model("bf16")
    total, recon, kl = m.elbo(x.to(DEV), y.to(DEV), None, M=2, eps=eps.to(DEV))
    total.backward()
    assert_close(recon[0], r["recon"], rtol=5e-2, atol=2e-2, what="afcrps (bf16)")
    assert_close(kl.cpu(), r["kl"], rtol=1.5e-1, atol=5e-2, what="kl (bf16)")
    num = da = db = 0.0
    for k, p in m.named_parameters():
        a, b = p.grad.detach().cpu().double().flatten(), og[k].double().flatten()
        assert torch.isfinite(a).all(), k
        num += float(a @ b); da += float(a @ a); db += float(b @ b)
    cos = num / np.sqrt(da * db)
    assert cos > 0.98, cos                                               # direction of the full 19.3 M-parameter gradient
    assert 0.8 < np.sqrt(da / db) < 1.25, np.sqrt(da / db)
    with torch.no_grad():
        m.eval()
        s = m.sample(x.to(DEV), 3, eps=make_eps(3, 2, 6).to(DEV))
    ref = O.sample_forward(P2, CFG2, x, make_eps(3, 2, 6))["out"]
    rel = float((s.cpu().double() - ref.double()).norm() / ref.double().norm())
    assert rel < 6e-2, rel


# ------------------------------------------------------------------------------------------------ cfg5
def _cfg3_model(dtype, max_batch, max_members, recon="afcrps"):
    torch.manual_seed(42)
    m = pa.ProbabilisticUNet(4, 1, 12, NF3, 32, MULT3, 1.0, 1.0, 0.0, dtype=dtype, recon=recon, max_batch=max_batch, max_members=max_members)
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for name, p in m.named_parameters():                      # conv1 / out_conv are zero-initialised in the reference
            if float(p.abs().sum()) == 0.0 and p.dim() == 4:
                p.copy_(torch.randn(p.shape, generator=g) * (0.5 / np.sqrt(p[0].numel())))
    return m.to(DEV)


def _fields256(B, seed):
    g = torch.Generator().manual_seed(seed)
    lo = torch.randn(B, 4, 16, 16, generator=g)
    x = torch.nn.functional.interpolate(lo, scale_factor=16.0, mode="nearest") + 0.1 * torch.randn(B, 4, 256, 256, generator=g)
    y = x[:, :1] + 0.5 * torch.randn(B, 1, 256, 256, generator=g)
    return x.to(DEV), y.to(DEV)


@pytest.mark.parametrize("dtype", ["f16", "f32"])
def test_cfg5_64_prior_samples_at_256(dtype):
    B, n = (2, 64) if dtype == "f16" else (1, 64)
    m = _cfg3_model(dtype, B, n).eval()
    x, _ = _fields256(B, 51)
    eps = torch.randn(n, B, 12, generator=torch.Generator().manual_seed(5)).to(DEV)
    with torch.no_grad():
        out = m.sample(x, n, eps=eps)
        assert tuple(out.shape) == (B, n, 1, 256, 256) and torch.isfinite(out).all()
        feat = m.unet(x); p = m.prior(x)
        mu, sg = p.base_dist.loc, p.base_dist.scale
        worst = 0.0
        for i in range(n):
            z = mu + sg * eps[i]
            ref = m.fcomb(feat, z)
            err = float((out[:, i] - ref).abs().max()); sc = float(ref.abs().max())
            worst = max(worst, err / max(sc, 1e-6))
        # same kernels, same operands: the fused call only differs in how many members share one W0.feat product
        # (f16: hidden activations of the three 1x1 layers are rounded to 11 bits in a different order)
        assert worst < (1e-2 if dtype == "f16" else 1e-5), worst
        # members differ from each other (the latent actually reaches the output), replays are bit-identical
        assert float((out[:, 0] - out[:, 1]).abs().max()) > 0
        m.assume_static_parameters = True
        out2 = m.sample(x, n, eps=eps); out3 = m.sample(x, n, eps=eps)
        assert torch.equal(out2, out) and torch.equal(out3, out)
        # hipGraph path observed, not assumed (pu_sample_graph_stats): with caller-stable buffers (x, eps, out=) the first call of a
        # tuple is eager, the second captures + launches, every later one replays; results stay bit-identical to the eager call
        m.use_sample_graph = True
        from probunet_amd import _lib as L_
        L_.lib().pu_set_sample_graph(m._ctx, 1)
        c0, r0, e0 = m.sample_graph_stats()
        buf = torch.empty_like(out)
        for it in range(4):
            m.sample(x, n, eps=eps, out=buf)
            assert torch.equal(buf, out), it
        c1, r1, e1 = m.sample_graph_stats()
        assert (c1 - c0, r1 - r0, e1 - e0) == (1, 3, 1), (c0, r0, e0, c1, r1, e1)
        L_.lib().pu_set_sample_graph(m._ctx, 0); m.use_sample_graph = False
        # the 64-sample latent_exploration call on ONE feature map broadcast with expand() (stride 0)
        z64 = mu[:1] + sg[:1] * eps[:, 0]
        g64 = m.fcomb(feat[:1].expand(n, -1, -1, -1), z64)
        assert float((g64[:, 0] - out[0, :, 0]).abs().max()) <= (1e-2 if dtype == "f16" else 1e-5) * float(out[0].abs().max())


def test_cfg5_crop_against_oracle():
    """64 prior samples per input on a 64 x 64 configuration the oracle finishes in a second: fp32 engine at rtol 1e-3 / atol 1e-5,
    f16 engine (the dtype cfg5 runs in) at a 16-bit bound."""
    cfg = O.Config(4, 1, 12, [32, 64, 128], 32, [1, 2, 4])
    P = random_params(cfg, seed=23)
    x, _ = make_fields(1, 4, 1, 64, 64, seed=55); eps = make_eps(64, 1, 12)
    ref = O.sample_forward(P, cfg, x, eps)["out"]
    for dtype in ("f32", "f16"):
        m = pa.ProbabilisticUNet(4, 1, 12, [32, 64, 128], 32, [1, 2, 4], 1.0, 1.0, 0.0, dtype=dtype, init=False)
        m.load_state_dict(P); m = m.to(DEV).eval()
        with torch.no_grad():
            out = m.sample(x.to(DEV), 64, eps=eps.to(DEV)).cpu()
        assert tuple(out.shape) == tuple(ref.shape)
        if dtype == "f32":
            assert_close(out, ref, what="64 prior samples (fp32 engine)")
        else:
            rel = float((out.double() - ref.double()).norm() / ref.double().norm())
            assert rel < 2e-2, rel


# ------------------------------------------------------------------------------------------------ cfg3 at B = 32
def _flat_grad(m):
    return torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).flatten() for p in m.parameters()]).clone()


def test_cfg3_batch32_shard_mean_and_determinism():
    m = _cfg3_model("f16", 32, 5).train(); m.dropout = 0.0
    x, y = _fields256(32, 61)
    eps = torch.randn(5, 32, 12, generator=torch.Generator().manual_seed(6)).to(DEV)

    def run(sl):
        m.zero_grad(set_to_none=True)
        t, r, k = m.elbo(x[sl].contiguous(), y[sl].contiguous(), None, M=5, eps=eps[:, sl].contiguous())
        t.backward()
        return float(t.detach()), r[0], k.clone(), _flat_grad(m)

    tf, rf, kf, gf = run(slice(0, 32))
    tf2, rf2, kf2, gf2 = run(slice(0, 32))
    assert torch.isfinite(gf).all() and np.isfinite(tf)
    assert torch.equal(kf, kf2)                                            # KL: no atomics anywhere on its path
    assert abs(rf - rf2) <= 1e-5 * abs(rf)                                 # loss scalar: float-atomic sum (order noise only)
    # weight-gradient slabs, GroupNorm rows are order-fixed; the remaining float atomics (GroupNorm / Fcomb parameter
    # gradients) give ulp-level noise: two runs agree to 1e-4 of the gradient norm
    assert float((gf.double() - gf2.double()).norm() / gf.double().norm()) < 1e-4
    ta, ra, ka, ga = run(slice(0, 16)); tb, rb, kb, gb = run(slice(16, 32))
    assert torch.equal(torch.cat([ka, kb]), kf)                            # samples are independent: bit-identical KL per pair
    assert abs(tf - 0.5 * (ta + tb)) <= 2e-3 * abs(tf)
    gm = 0.5 * (ga.double() + gb.double())
    rel = float((gf.double() - gm).norm() / gm.norm())
    assert rel < 2e-2, rel                                                 # f16 activations; identical in exact arithmetic


def test_create_rejects_plans_beyond_the_32_bit_pixel_offset_range():
    """pu_create fails loudly (PU_ERR_INVALID, host-side, before any allocation) when the largest activation of the plan reaches 2^32
    elements: the convolution staging plans keep per-pixel element offsets in 32 bits (B > 682 at cfg3: 683 x 256 x 256 x 96)."""
    import ctypes as C
    from probunet_amd import _lib as L_
    m = pa.ProbabilisticUNet(4, 1, 12, NF3, 32, MULT3, 1.0, 1.0, 0.0, dtype="f16", init=False)
    ctx = C.c_void_p()
    cfg = m._cfg_struct(256, 256, 683, 1)
    rc = L_.lib().pu_create(C.byref(cfg), 0, C.byref(ctx))
    assert rc == -1 and b"2^32" in L_.lib().pu_last_error(None)


_MASK_SCRIPT = r"""
import sys, torch
sys.path.insert(0, sys.argv[1])
import probunet_amd as pa
from oracle import probunet_oracle as O
from tests.helpers import random_params
args = (4, 1, 6, [32, 64, 128], 32, [1, 2, 4])
P = random_params(O.Config(*args), 3)
g = torch.Generator().manual_seed(11)
x = torch.randn(2, 4, 64, 64, generator=g); y = torch.randn(2, 1, 64, 64, generator=g); eps = torch.randn(2, 2, 6, generator=g)
m = pa.ProbabilisticUNet(*args, 0.7, 1.3, 0.0, dtype="f16", init=False, max_batch=2, max_members=2)
m.load_state_dict(P); m = m.to("cuda:0").train(); m.dropout = 0.0
# the sub-module path (no Fcomb in the graph: its float atomics would put order noise into dz and from there into these gradients)
p = m.prior(x.cuda()); q = m.posterior(x.cuda(), y.cuda())
torch.distributions.kl.kl_divergence(q, p).mean().backward(); torch.cuda.synchronize()
torch.save({k: p.grad.detach().cpu() for k, p in m.named_parameters() if k.startswith(("prior", "posterior")) and p.grad is not None}, sys.argv[2])
"""


def test_relu_backward_folded_into_the_data_gradient_is_bit_identical_to_its_own_pass(tmp_path):
    """Latent encoders (conv -> ReLU chains, src/prob_unet.py:60-105): the f16 engine folds each ReLU backward into the epilogue of
    the data gradient above it (ConvArgs::relu_mask).  Zeroing before or after the rounding to f16 is the same value, so the encoder
    gradients must equal those of the build that runs every ReLU backward as its own pass (PU_NO_RELU_MASK_FUSE=1) bit for bit.
    Two child processes, one after the other (the switch is read once per process)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for tag, env in (("fused", {}), ("separate", {"PU_NO_RELU_MASK_FUSE": "1"})):
        f = str(tmp_path / f"{tag}.pt")
        e = dict(os.environ); e.update(env)
        r = subprocess.run([sys.executable, "-c", _MASK_SCRIPT, root, f], env=e, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(torch.load(f, weights_only=True))
    assert outs[0].keys() == outs[1].keys() and len(outs[0]) > 8
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), k
    assert any(float(v.abs().max()) > 0 for v in outs[0].values())
