"""GPU: the HIP engine, through the drop-in ProbabilisticUNet host class (ctypes -> C ABI), against
 (a) golden vectors captured from the imported reference (tests/golden, tools/make_golden.py), and
 (b) the CPU oracle on the same seeded inputs.
fp32 engine: north_star tolerance rtol 1e-3 / atol 1e-5 (atol scaled by the tensor's max-abs, see helpers.assert_close).
fp16 / bf16 engines (MFMA low-precision operands, fp32 accumulate): looser, dtype-appropriate bounds stated below."""
import numpy as np
import pytest, torch
import probunet_amd as pa
from oracle import probunet_oracle as O
from tests.helpers import load_golden, oracle_cfg, filled_params, t, assert_close, GOLDEN_NAMES
from tests.filler import checksum, make_fields, make_eps

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def build(meta, dtype="f32", recon="afcrps", **kw):
    c = meta["config"]
    m = pa.ProbabilisticUNet(c["input_channels"], c["num_classes"], c["latent_dim"], c["num_filters"], c["model_channels"],
                             c["channel_mult"], 0.7, 1.3, 0.2, dtype=dtype, recon=recon, init=False, **kw)
    m.load_state_dict(filled_params(oracle_cfg(c)))
    return m.to(DEV)


def grads_of(m):
    return {k: (p.grad.detach().cpu() if p.grad is not None else torch.zeros_like(p).cpu()) for k, p in m.named_parameters()}


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_f32_elbo_matches_reference_golden(name):
    meta, g = load_golden(name)
    m = build(meta).train()
    m.dropout = 0.0                                   # golden was captured with dropout off
    x, y, eps = t(g["x"]).to(DEV), t(g["y"]).to(DEV), t(g["eps"]).to(DEV)
    total, recon, kl = m.elbo(x, y, None, M=eps.shape[0], alpha=float(g["alpha"]), eps=eps)
    assert isinstance(recon[0], float)
    assert_close(total.detach().cpu(), g["total"], what="total")
    assert_close(recon[0], g["crps"], what="crps")
    assert_close(kl.cpu(), g["kl"], what="kl")
    total.backward()
    gr = grads_of(m)
    for k, cs in meta["grad_checksums"].items():
        scale = cs["abssum"] / max(cs["n"], 1)
        key = "grad::" + k
        if key in g:
            assert_close(gr[k], g[key], rtol=1e-3, atol=1e-5 + 2e-3 * scale, what=key)
        mine = checksum(gr[k])
        assert abs(mine["abssum"] - cs["abssum"]) <= 3e-3 * cs["abssum"] + 1e-5 * cs["n"], (k, mine["abssum"], cs["abssum"])
    # sub-modules (model.unet / prior / posterior / fcomb) against the same golden
    with torch.no_grad():
        m.eval()
        feat = m.unet(x)
        assert_close(feat.cpu(), g["feat"], what="feat")
        p = m.prior(x); q = m.posterior(x, y)
        assert_close(p.base_dist.loc.cpu(), g["mu_p"], what="mu_p"); assert_close(p.base_dist.scale.cpu(), g["sig_p"], what="sig_p")
        assert_close(q.base_dist.loc.cpu(), g["mu_q"], what="mu_q"); assert_close(q.base_dist.scale.cpu(), g["sig_q"], what="sig_q")
        for mm in range(eps.shape[0]):
            z = q.base_dist.loc + q.base_dist.scale * eps[mm]
            assert_close(m.fcomb(feat, z).cpu(), g["preds"][:, mm], what=f"preds[{mm}]")
        # forward(training=False) == one prior sample; sample() with explicit eps
        s = m.sample(x, 1, eps=eps[:1])
        assert_close(s[:, 0].cpu(), g["sample_prior"], what="sample_prior")
        K = g["grid_z"].shape[0]
        out = m.fcomb(feat[:1].expand(K, -1, -1, -1), t(g["grid_z"]).to(DEV))       # stride-0 expand (latent_exploration.py:125)
        assert_close(out.cpu(), g["grid_out"], what="grid_out")
        assert torch.equal(m.fcomb.tile(torch.arange(6.).reshape(2, 3, 1), 2, 4), t(g["tile_out"]))


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_f32_l1_elbo(name):
    meta, g = load_golden(name)
    m = build(meta, recon="l1").train(); m.dropout = 0.0
    x, y, eps = t(g["x"]).to(DEV), t(g["y"]).to(DEV), t(g["eps"]).to(DEV)
    total, recon, kl, kl2 = m.elbo(x, y, None, eps=eps[:1])
    assert_close(recon[0], g["l1"], what="l1"); assert_close(kl2.cpu(), g["kl2"], what="kl2")
    assert_close(total.detach().cpu(), g["total_l1"], what="total_l1")
    total.backward()
    P = filled_params(oracle_cfg(meta["config"]))
    r, og = O.elbo_with_grads(P, oracle_cfg(meta["config"]), t(g["x"]), t(g["y"]), t(g["eps"])[:1], beta0=0.7, beta1=1.3, beta2=0.2, recon="l1")
    gr = grads_of(m)
    for k, v in og.items():
        scale = float(v.abs().mean())
        assert_close(gr[k], v, rtol=1e-3, atol=1e-5 + 2e-3 * scale, what="l1 grad " + k)


@pytest.mark.parametrize("name", ["tiny22", "tiny41"])
def test_adamw_step_matches_reference(name):
    """elbo -> zero_grad -> backward -> step (train_prob_unet_model.py:133-141) with torch.optim.AdamW(lr=1e-4)."""
    meta, g = load_golden(name)
    m = build(meta).train(); m.dropout = 0.0
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4)
    x, y, eps = t(g["x"]).to(DEV), t(g["y"]).to(DEV), t(g["eps"]).to(DEV)
    loss, rl, kl = m.elbo(x, y, None, M=eps.shape[0], eps=eps)
    opt.zero_grad(); loss.backward(); opt.step()
    sd = m.state_dict()
    for k, cs in meta["after_adamw"].items():
        mine = checksum(sd[k])
        assert abs(mine["sum"] - cs["sum"]) <= 1e-5 * max(1.0, cs["abssum"]), k
        assert abs(mine["abssum"] - cs["abssum"]) <= 1e-5 * max(1.0, cs["abssum"]), k
    # a second step must see the updated weights (packed low-precision copies are rebuilt)
    loss2, _, _ = m.elbo(x, y, None, M=eps.shape[0], eps=eps)
    assert abs(float(loss2.detach()) - float(loss.detach())) > 0


@pytest.mark.parametrize("dtype,rtol,atol", [("f16", 3e-2, 2e-2), ("bf16", 1e-1, 8e-2)])
@pytest.mark.parametrize("name", ["tiny22", "mid11"])
def test_low_precision_engines_track_oracle(name, dtype, rtol, atol):
    meta, g = load_golden(name)
    m = build(meta, dtype=dtype).train(); m.dropout = 0.0
    x, y, eps = t(g["x"]).to(DEV), t(g["y"]).to(DEV), t(g["eps"]).to(DEV)
    total, recon, kl = m.elbo(x, y, None, M=eps.shape[0], eps=eps)
    assert_close(recon[0], g["crps"], rtol=rtol, atol=atol, what="crps")
    assert_close(kl.cpu(), g["kl"], rtol=3 * rtol, atol=atol, what="kl")
    total.backward()
    gr = grads_of(m)
    # direction of the full gradient vs the fp32 reference
    num = den1 = den2 = 0.0
    for k, cs in meta["grad_checksums"].items():
        key = "grad::" + k
        if key in g:
            a = gr[k].double().flatten(); b = t(g[key]).double().flatten()
            num += float(a @ b); den1 += float(a @ a); den2 += float(b @ b)
    cos = num / (den1 ** 0.5 * den2 ** 0.5)
    assert cos > (0.995 if dtype == "f16" else 0.97), cos
    with torch.no_grad():
        m.eval()
        assert_close(m.unet(x).cpu(), g["feat"], rtol=rtol, atol=atol, what="feat")


def test_submodule_autograd_fcomb_and_encoders():
    """gradient_magnitude_ratio-style use (latent_exploration.py:467-481): d fcomb / d z and d fcomb / d features,
    plus kl_divergence on the distribution objects, through autograd."""
    meta, g = load_golden("tiny22")
    cfg = oracle_cfg(meta["config"]); P = filled_params(cfg)
    m = build(meta).eval()
    x, y = t(g["x"]).to(DEV), t(g["y"]).to(DEV)
    feat = m.unet(x).detach().requires_grad_(True)
    z = t(g["eps"])[0].to(DEV).requires_grad_(True)
    out = m.fcomb(feat, z)
    out.square().sum().backward()
    fo = t(g["feat"]).clone().requires_grad_(True); zo = t(g["eps"])[0].clone().requires_grad_(True)
    oo = O.fcomb_forward(P, cfg, fo, zo); oo.square().sum().backward()
    assert_close(z.grad.cpu(), zo.grad, rtol=2e-3, what="dz")
    assert_close(feat.grad.cpu(), fo.grad, rtol=2e-3, what="dfeat")
    m.zero_grad()
    p = m.prior(x); q = m.posterior(x, y)
    kl = torch.distributions.kl.kl_divergence(q, p).mean()
    kl.backward()
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items() if "resample" not in k}
    mu_p, ls_p = O.gauss_forward(Pg, cfg, "prior", t(g["x"])); mu_q, ls_q = O.gauss_forward(Pg, cfg, "posterior", t(g["x"]), t(g["y"]))
    O.kl_diag_normal(mu_q, O.sigma_of(ls_q), mu_p, O.sigma_of(ls_p)).mean().backward()
    for k in ("prior.encoder.0.weight", "posterior.encoder.7.weight", "posterior.conv_mu.bias", "prior.conv_log_sigma.weight"):
        v = Pg[k].grad
        assert_close(dict(m.named_parameters())[k].grad.cpu(), v, rtol=2e-3, atol=1e-5 + 2e-3 * float(v.abs().mean()), what=k)


def test_forward_api_and_errors():
    meta, g = load_golden("tiny22")
    m = build(meta)
    x, y = t(g["x"]).to(DEV), t(g["y"]).to(DEV)
    with torch.no_grad():
        out = m(x, t=None, training=False)
        assert out.shape == y.shape and m.prior_latent_space is not None
        out2 = m(x, y, training=True)
        assert m.posterior_latent_space is not None and out2.shape == y.shape
    out3 = m(x, y)                                        # autograd path through the three sub-modules
    out3.mean().backward()
    assert dict(m.named_parameters())["fcomb.layers.4.bias"].grad is not None
    with pytest.raises(ValueError):
        m.elbo(x, y, None, M=1)
    cpu_model = pa.ProbabilisticUNet(2, 2, 6, [8, 16, 32], 8, [1, 2, 4], 1, 1, 0)
    with pytest.raises(pa._lib.ProbUNetLibraryError):
        cpu_model.elbo(t(g["x"]), t(g["y"]), None, M=2)


def test_dropout_is_applied_in_train_mode_only():
    meta, g = load_golden("tiny22")
    m = build(meta)
    x = t(g["x"]).to(DEV)
    with torch.no_grad():
        m.eval(); a = m.unet(x); b = m.unet(x)
        assert torch.equal(a, b)
        m.train(); c1 = m.unet(x); c2 = m.unet(x)
        assert not torch.equal(c1, c2) and not torch.equal(a, c1)
        assert float((c1 - a).abs().mean()) < float(a.abs().mean())              # a p=0.1 perturbation, not noise


def test_cfg1_elbo_against_oracle_with_dropout_off():
    """BASELINE config 1 (1->1, 64x64, depth 3, L=6, B=4, M=5): fp32 engine vs the CPU oracle, seeded inputs.
    Forward quantities: north_star tolerance.  Gradients: per-tensor relative L2 error and cosine.
    At this size a handful of ReLU / max-pool / |.| masks sit within fp32 rounding of their switching point
    (tests/diagnostics/mask_flips.py counts them: ~400 of 262k pre-activations within 1e-4 of zero per layer), so two
    fp32 evaluations with different summation order legitimately differ by isolated mask flips: torch-CPU fp32 itself
    jumps from 1e-6 to 3e-4 relative error (vs an fp64 run) exactly at those layers.  Bound: every tensor within
    3e-2 relative L2 and cosine >= 0.9995 of the oracle, median relative error <= 2e-3."""
    cfg = O.Config(1, 1, 6, [32, 64, 128], 32, [1, 2, 4])
    meta = dict(config=dict(input_channels=1, num_classes=1, latent_dim=6, num_filters=[32, 64, 128], model_channels=32, channel_mult=[1, 2, 4]))
    m = build(meta).train(); m.dropout = 0.0
    x, y = make_fields(4, 1, 1, 64, 64); eps = make_eps(5, 4, 6)
    total, recon, kl = m.elbo(x.to(DEV), y.to(DEV), None, M=5, eps=eps.to(DEV))
    total.backward()
    r, og = O.elbo_with_grads(filled_params(cfg), cfg, x, y, eps, beta0=0.7, beta1=1.3)
    assert_close(total.detach().cpu(), r["total"], what="total"); assert_close(kl.cpu(), r["kl"], what="kl")
    assert_close(recon[0], r["recon"], what="crps")
    gr = grads_of(m)
    rels = []
    for k, v in og.items():
        rel = float((gr[k].double() - v.double()).norm()) / (float(v.double().norm()) + 1e-12)
        rels.append(rel)
        if float(v.norm()) < 1e-6:
            continue
        cos = float((gr[k].double().flatten() @ v.double().flatten()) / (gr[k].double().norm() * v.double().norm()))
        assert rel < 3e-2 and cos > 0.9995, (k, rel, cos)
    assert float(np.median(rels)) < 2e-3, float(np.median(rels))


def test_f16_static_loss_scale_keeps_tiny_gradients():
    """With beta_0 = 1e-4 the activation gradients (~1e-8) would flush to zero in fp16; the engine's automatic static
    loss scale (pu_config.grad_scale = 0) keeps them and removes the scale from every parameter gradient."""
    meta, g = load_golden("mid11")
    cfg = oracle_cfg(meta["config"]); P = filled_params(cfg)
    m = build(meta, dtype="f16").train(); m.dropout = 0.0
    m.beta_0, m.beta_1 = 1e-4, 0.0
    x, y, eps = t(g["x"]).to(DEV), t(g["y"]).to(DEV), t(g["eps"]).to(DEV)
    total, recon, kl = m.elbo(x, y, None, M=eps.shape[0], eps=eps)
    total.backward()
    r, og = O.elbo_with_grads(P, cfg, t(g["x"]), t(g["y"]), t(g["eps"]), beta0=1e-4, beta1=0.0)
    gr = grads_of(m)
    num = den1 = den2 = 0.0
    for k, v in og.items():
        if k.startswith("unet.") or k.startswith("fcomb."):
            a = gr[k].double().flatten(); b = v.double().flatten()
            num += float(a @ b); den1 += float(a @ a); den2 += float(b @ b)
    assert den1 > 0.25 * den2, (den1, den2)                    # not flushed to zero
    # the closed-form filler network amplifies rounding ~1e4x (torch fp32 vs fp64: 2e-3), so fp16 U-Net gradients agree
    # with the fp32 oracle only to a cosine of ~0.985 with or without the loss scale
    assert num / (den1 ** 0.5 * den2 ** 0.5) > 0.97


def test_side_stream_overlap_is_race_free():
    """The f16/bf16 engines run weight gradients and the latent encoders on side HIP streams (fork/join by events).
    Same inputs with overlap off and on must give the same loss and gradients (slab reductions are order-fixed; only the
    Fcomb weight-gradient atomics may differ in the last bits)."""
    from probunet_amd import _lib as L
    c = dict(input_channels=4, num_classes=1, latent_dim=6, num_filters=[32, 64, 128], model_channels=32, channel_mult=[1, 2, 4])
    cfg = oracle_cfg(c)
    P = filled_params(cfg)
    x, y = make_fields(8, 4, 1, 64, 64); eps = make_eps(3, 8, 6)
    res = []
    for overlap in (0, 1, 1):
        m = pa.ProbabilisticUNet(4, 1, 6, [32, 64, 128], 32, [1, 2, 4], 0.7, 1.3, 0.0, dtype="f16", init=False)
        m.load_state_dict(P); m = m.to(DEV).train(); m.dropout = 0.0
        m._ensure(64, 64, 8, 3)
        L.lib().pu_set_overlap(m._ctx, overlap)
        for _ in range(2):                                   # second call reuses every buffer: stale-data hazards would show
            m.zero_grad()
            total, _, kl = m.elbo(x.to(DEV), y.to(DEV), None, M=3, eps=eps.to(DEV))
            total.backward()
        torch.cuda.synchronize()
        res.append((float(total.detach()), torch.cat([p.grad.flatten() for p in m.parameters()]).double().cpu()))
    for k in (1, 2):
        assert abs(res[k][0] - res[0][0]) <= 1e-6 * abs(res[0][0])
        rel = float((res[k][1] - res[0][1]).norm() / res[0][1].norm())
        assert rel < 1e-5, rel


def test_bf16_engine_and_posterior_sampling():
    meta, g = load_golden("mid11")
    m = build(meta, dtype="bf16", max_members=40).eval()
    x, y = t(g["x"]).to(DEV), t(g["y"]).to(DEV)
    with torch.no_grad():
        eps = torch.randn(40, x.shape[0], 6, device=DEV)
        out = m.sample(x, 40, target=y, eps=eps)              # posterior samples, n > 16
        assert out.shape == (x.shape[0], 40, 1, 32, 32) and torch.isfinite(out).all()
        feat = m.unet(x); q = m.posterior(x, y)
        z = q.base_dist.loc + q.base_dist.scale * eps[7]
        assert_close(out[:, 7].cpu(), m.fcomb(feat, z).cpu(), rtol=5e-2, atol=5e-2, what="sample vs sub-modules")
    with pytest.raises(pa._lib.ProbUNetLibraryError):
        m.train(); m.elbo(x, y, None, M=33)                    # fused afCRPS kernel supports M <= 32


@pytest.mark.parametrize("name", ["tiny22"])
def test_flat_adamw_matches_reference_step(name):
    """pu_adamw_step on the flat buffers == torch.optim.AdamW(lr=1e-4) of the reference trainer (golden after_adamw), two steps
    cross-checked against torch's own AdamW on a twin model."""
    meta, g = load_golden(name)
    x, y, eps = t(g["x"]).to(DEV), t(g["y"]).to(DEV), t(g["eps"]).to(DEV)
    m = build(meta).train(); m.dropout = 0.0
    opt = pa.FlatAdamW(m, lr=1e-4)
    twin = build(meta).train(); twin.dropout = 0.0
    topt = torch.optim.AdamW(twin.parameters(), lr=1e-4)
    for it in range(2):
        for mod, o in ((m, opt), (twin, topt)):
            loss, _, _ = mod.elbo(x, y, None, M=eps.shape[0], eps=eps)
            o.zero_grad(); loss.backward(); o.step()
        if it == 0:
            sd = m.state_dict()
            for k, cs in meta["after_adamw"].items():
                mine = checksum(sd[k])
                assert abs(mine["sum"] - cs["sum"]) <= 1e-5 * max(1.0, cs["abssum"]), k
    a = torch.cat([p.detach().flatten() for p in m.parameters()]); b = torch.cat([p.detach().flatten() for p in twin.parameters()])
    assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(b.abs().max()))


@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_reference_3to3_config_nonsquare_batch1(dtype):
    """The configuration the reference actually trains (main.py:35-45: 3 -> 3 planes, num_filters [32,64,128,256],
    model_channels 32, channel_mult [1,2,4,8]) on a NON-square 64x128 field, batch 1 and batch 3, latent 32, M = 4,
    against the oracle (seeded closed-form inputs)."""
    cfg = O.Config(3, 3, 32, [32, 64, 128, 256], 32, [1, 2, 4, 8])
    P = filled_params(cfg)
    for B in (1, 3):
        yy, xx = torch.meshgrid(torch.arange(64.), torch.arange(128.), indexing="ij")
        x = torch.stack([torch.stack([torch.sin(0.13 * (c + 1) * xx + 0.09 * (b + 2) * yy) + 0.3 * torch.cos(0.31 * yy * (c + 1)) for c in range(3)]) for b in range(B)])
        y = 0.6 * x + 0.3 * torch.sin(1.3 * xx - 0.7 * yy)[None, None]
        eps = make_eps(4, B, 32)
        m = pa.ProbabilisticUNet(3, 3, 32, [32, 64, 128, 256], 32, [1, 2, 4, 8], 0.7, 1.3, 0.0, dtype=dtype, init=False)
        m.load_state_dict(P); m = m.to(DEV).train(); m.dropout = 0.0
        total, recon, kl = m.elbo(x.to(DEV), y.to(DEV), None, M=4, eps=eps.to(DEV))
        total.backward()
        r, og = O.elbo_with_grads(P, cfg, x, y, eps, beta0=0.7, beta1=1.3)
        if dtype == "f32":
            assert_close(total.detach().cpu(), r["total"], what="total"); assert_close(kl.cpu(), r["kl"], what="kl")
            assert_close(recon[0], r["recon"], what="crps")
            gr = grads_of(m)
            rels = []
            for k, v in og.items():
                if float(v.norm()) < 1e-7: continue
                rels.append(float((gr[k].double() - v.double()).norm() / v.double().norm()))
            assert max(rels) < 3e-2 and float(np.median(rels)) < 2e-3, (max(rels), float(np.median(rels)))
        else:
            assert_close(recon[0], r["recon"], rtol=5e-2, atol=2e-2, what="crps f16")
            assert_close(kl.cpu(), r["kl"], rtol=1e-1, atol=5e-2, what="kl f16")
        with torch.no_grad():
            m.eval()
            out = m(x.to(DEV), training=False)
            assert out.shape == (B, 3, 64, 128) and torch.isfinite(out).all()
