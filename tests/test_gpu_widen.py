"""GPU: the rows SURVEY.md §8(f) marks "next", through the same C ABI:
  f1 trainer call pattern (train_probunet_step / eval_probunet_model / beta schedule),
  f2 WMSE-MS-SSIM reconstruction term (live elbo; MS-SSIM arithmetic is PARITY UNPINNED - pytorch-msssim 1.0.0 is absent, the
     oracle restates its published algorithm; the WMSE half / combination / data_range are pinned by tests/golden/wmse.json),
  f3 residual_to_hr fused into sampling,
  f4 ClimEx item transform + statistics on the device.
Tolerances: fp32 kernels vs the fp32 torch oracle, rtol 1e-3 / atol 1e-5 (scaled), stated per test."""
import ctypes as C
import numpy as np
import pytest, torch
import probunet_amd as pa
from probunet_amd import _lib as L
from oracle import probunet_oracle as O
from tests.helpers import oracle_cfg, filled_params, assert_close
from tests.filler import make_fields, make_eps

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _fields(B, C, H, W, seed):
    x, y = make_fields(B, C, C, H, W, seed=seed)
    return x, y


@pytest.mark.parametrize("B,M,C,H,W,lam,dr", [(2, 1, 1, 128, 128, 0.0, None), (2, 2, 2, 104, 120, 0.3, None), (1, 3, 1, 256, 128, 0.5, 7.5),
                                             (3, 1, 1, 97, 113, 0.0, None)])
def test_wmse_msssim_kernels_match_oracle(B, M, C, H, W, lam, dr):
    """Loss, logged components and d(loss)/d(pred) of the HIP kernels vs torch autograd on the oracle restatement.
    Odd sizes (104 -> 52 -> 26 -> 13 -> 7, 97 -> 49 -> 25 -> 13 -> 7) exercise the padded 2x2 pooling of the pyramid."""
    x, y = _fields(B, C, H, W, 5)
    tgt = (2.0 * y + 0.5)
    preds = torch.stack([0.9 * tgt + 0.25 * torch.roll(x, m + 1, -1) * (1 + 0.1 * m) for m in range(M)], dim=1).requires_grad_(True)
    parts = [O.wmse_ms_ssim_loss(preds[:, m], tgt, 0.3, 0.2, lam, dr) for m in range(M)]
    loss = torch.stack([p[0] for p in parts]).mean()
    loss.backward()
    pd, td = preds.detach().to(DEV).contiguous(), tgt.to(DEV).contiguous()
    scal = torch.zeros(L.PU_NUM_SCALARS, device=DEV); dp = torch.full_like(pd, float("nan"))
    L.check(L.lib().pu_op_wmse_msssim(L.ptr(pd), L.ptr(td), B, M, C, H, W, 0.3, 0.2, lam, -1.0 if dr is None else dr, 1.0,
                                      L.ptr(scal), L.ptr(dp), L.current_stream()), None, "pu_op_wmse_msssim")
    s = scal.cpu()
    assert_close(s[L.PU_S_RECON], loss.detach(), rtol=1e-4, atol=1e-6, what="recon")
    assert_close(s[L.PU_S_WMSE], parts[-1][1].detach(), rtol=1e-4, atol=1e-6, what="wmse(last member)")
    assert_close(s[L.PU_S_MSSSIM], parts[-1][2].detach(), rtol=1e-4, atol=1e-6, what="1-msssim(last member)")
    g = preds.grad
    assert torch.isfinite(dp).all()
    assert_close(dp.cpu(), g, rtol=1e-3, atol=2e-5, what="dpred")
    rel = float((dp.cpu().double() - g.double()).norm() / g.double().norm())
    assert rel < 1e-4, rel
    # forward-only call leaves no gradient behind and gives the same scalars
    scal2 = torch.zeros_like(scal)
    L.check(L.lib().pu_op_wmse_msssim(L.ptr(pd), L.ptr(td), B, M, C, H, W, 0.3, 0.2, lam, -1.0 if dr is None else dr, 1.0,
                                      L.ptr(scal2), None, L.current_stream()), None, "pu_op_wmse_msssim")
    assert_close(scal2.cpu()[L.PU_S_RECON], s[L.PU_S_RECON], rtol=1e-6, atol=1e-7, what="fwd-only recon")


def test_msssim_identical_and_constant_images():
    """Known answers of MS-SSIM itself: identical images -> 1 (loss 0); constant images c1 vs c2 -> every cs = 1 and
    ms = l^w5 with l = (2 c1 c2 + C1) / (c1^2 + c2^2 + C1)."""
    B, H = 2, 128
    a = make_fields(B, 1, 1, H, H, seed=9)[1].to(DEV).contiguous()
    scal = torch.zeros(L.PU_NUM_SCALARS, device=DEV)
    L.check(L.lib().pu_op_wmse_msssim(L.ptr(a), L.ptr(a), B, 1, 1, H, H, 0.007, 0.048, 0.0, 2.0, 1.0, L.ptr(scal), None, L.current_stream()), None, "op")
    assert abs(float(scal[L.PU_S_RECON])) < 1e-6
    c1, c2, R = 0.8, 0.5, 2.0
    x = torch.full((B, 1, H, H), c1, device=DEV); y = torch.full((B, 1, H, H), c2, device=DEV)
    L.check(L.lib().pu_op_wmse_msssim(L.ptr(x), L.ptr(y), B, 1, 1, H, H, 0.007, 0.048, 0.0, R, 1.0, L.ptr(scal), None, L.current_stream()), None, "op")
    C1 = (0.01 * R) ** 2
    want = 1.0 - ((2 * c1 * c2 + C1) / (c1 * c1 + c2 * c2 + C1)) ** 0.1333
    assert abs(float(scal[L.PU_S_RECON]) - want) < 1e-4, (float(scal[L.PU_S_RECON]), want)   # fp32 E[x^2] - mu^2 noise against C2


def _small_model(recon, dtype="f32", H=128, cin=2, cout=1):
    c = dict(input_channels=cin, num_classes=cout, latent_dim=4, num_filters=[8, 16], model_channels=8, channel_mult=[1, 2])
    m = pa.ProbabilisticUNet(cin, cout, 4, [8, 16], 8, [1, 2], 0.7, 1.3, 0.0, dtype=dtype, recon=recon, init=False)
    cfg = oracle_cfg(c)
    P = filled_params(cfg)
    m.load_state_dict(P)
    return m.to(DEV), cfg, P


@pytest.mark.parametrize("M,lam", [(1, 0.0), (2, 0.4)])
def test_live_elbo_wmse_msssim_f32_against_oracle(M, lam):
    """The live elbo (prob_unet.py:229-267): 5-tuple return, total = beta0 * recon + beta1 * mean KL, parameter gradients."""
    m, cfg, P = _small_model("wmse_msssim")
    m.train(); m.dropout = 0.0
    x, y = make_fields(2, 2, 1, 128, 128, seed=3); y = 1.5 * y
    eps = make_eps(M, 2, 4)
    out = m.elbo(x.to(DEV), y.to(DEV), None, M=M, eps=eps.to(DEV), alpha_w=0.3, beta_w=0.2, lam_w=lam)
    assert len(out) == 5 and isinstance(out[1][0], float) and isinstance(out[3], float) and isinstance(out[4], float)
    total, recon, kl, wmse, msl = out
    total.backward()
    r, og = O.elbo_with_grads(P, cfg, x, y, eps, beta0=0.7, beta1=1.3, recon="wmse_msssim", alpha_w=0.3, beta_w=0.2, lam_w=lam)
    assert_close(total.detach().cpu(), r["total"], what="total"); assert_close(kl.cpu(), r["kl"], what="kl")
    assert_close(recon[0], r["recon"], what="recon"); assert_close(wmse, r["wmse"], what="wmse"); assert_close(msl, r["msssim"], what="msssim")
    rels = []
    for k, p in m.named_parameters():
        v = og[k]; gk = p.grad.detach().cpu()
        if float(v.norm()) < 1e-7:
            continue
        rel = float((gk.double() - v.double()).norm() / v.double().norm()); rels.append(rel)
        assert rel < 2e-2, (k, rel)
    assert float(np.median(rels)) < 2e-3, float(np.median(rels))
    # no_grad (eval_probunet_model): forward only, same scalars, nothing delivered
    with torch.no_grad():
        o2 = m.elbo(x.to(DEV), y.to(DEV), None, M=M, eps=eps.to(DEV), alpha_w=0.3, beta_w=0.2, lam_w=lam)
    assert not o2[0].requires_grad and abs(o2[1][0] - recon[0]) < 1e-6
    with pytest.raises(AssertionError):
        _small_model("wmse_msssim", H=64)[0].elbo(x[:, :, :64, :64].to(DEV), y[:, :, :64, :64].to(DEV), None)


def test_live_elbo_f16_engine_tracks_f32():
    m32, cfg, P = _small_model("wmse_msssim")
    m16, _, _ = _small_model("wmse_msssim", dtype="f16")
    x, y = make_fields(2, 2, 1, 128, 128, seed=3); eps = make_eps(1, 2, 4)
    res = []
    for m in (m32, m16):
        m.train(); m.dropout = 0.0
        t_, r_, kl_, w_, s_ = m.elbo(x.to(DEV), y.to(DEV), None, eps=eps.to(DEV), lam_w=0.2, alpha_w=0.3, beta_w=0.2)
        t_.backward()
        res.append((float(t_.detach()), torch.cat([p.grad.flatten() for p in m.parameters()]).cpu()))
    assert abs(res[0][0] - res[1][0]) < 2e-2 * abs(res[0][0]) + 1e-3
    cos = float(res[0][1].double() @ res[1][1].double() / (res[0][1].double().norm() * res[1][1].double().norm()))
    assert cos > 0.98, cos


def test_sample_hr_fused_reconstruction():
    m, cfg, P = _small_model("afcrps", H=32)
    m.eval()
    x, y = make_fields(2, 2, 1, 32, 32, seed=4)
    eps = make_eps(3, 2, 4)
    lrinterp = (0.5 * y + 1.0); std = (0.3 + 0.1 * torch.cos(torch.arange(32 * 32, dtype=torch.float32)).reshape(1, 32, 32).abs())
    ref = O.sample_forward(P, cfg, x, eps)["out"]
    want = O.residual_to_hr(ref, lrinterp, std, 1e-10)
    got = m.sample_hr(x.to(DEV), 3, lrinterp.to(DEV), std.to(DEV), eps=eps.to(DEV))
    assert_close(got.cpu(), want, what="sample_hr")
    got_sp = m.sample_hr(x.to(DEV), 3, lrinterp.to(DEV), std.to(DEV), eps=eps.to(DEV), softplus=True)
    assert_close(got_sp.cpu(), O.softplus_climex(want), what="sample_hr+softplus")
    plain = m.sample(x.to(DEV), 3, eps=eps.to(DEV))
    assert_close(pa.ProbabilisticUNet.reconstruct(plain, lrinterp.to(DEV).unsqueeze(1), std.to(DEV)).cpu(), want, what="reconstruct")
    # f16 engine takes the MFMA Fcomb kernel: same epilogue
    m16, _, _ = _small_model("afcrps", dtype="f16", H=32)
    got16 = m16.eval().sample_hr(x.to(DEV), 3, lrinterp.to(DEV), std.to(DEV), eps=eps.to(DEV))
    assert_close(got16.cpu(), want, rtol=3e-2, atol=2e-2, what="sample_hr f16")


@pytest.mark.parametrize("N,C,H,W,k", [(5, 3, 32, 48, 4), (3, 1, 64, 64, 8), (2, 2, 16, 16, 2)])
def test_climex_transforms_on_device(N, C, H, W, k):
    g = torch.Generator().manual_seed(11)
    hr = torch.randn(N, C, H, W, generator=g) * 2.0 + torch.linspace(-1, 3, C).view(1, C, 1, 1)
    (ml, sl), (mh, sh) = O.lr_stats(hr, k)
    (ml2, sl2), (mh2, sh2) = pa.data.lr_stats(hr.to(DEV), k)
    for a, b, w in ((ml2, ml, "mean_lr"), (sl2, sl, "std_lr"), (mh2, mh, "mean_hr"), (sh2, sh, "std_hr")):
        assert_close(a.cpu(), b, rtol=1e-5, atol=1e-6, what=w)
    want = O.lrinterp_to_residuals(hr, k, mh, sh, 1e-10)
    got = pa.data.lrinterp_to_residuals(hr.to(DEV), k, mh.to(DEV), sh.to(DEV), 1e-10)
    for key in ("inputs", "targets", "lrinterp", "lr"):
        assert_close(got[key].cpu(), want[key], rtol=1e-5, atol=1e-5, what=key)
    tr = pa.data.ClimExTransform(k).fit(hr.to(DEV))
    item = tr(hr[:2].to(DEV))
    assert_close(item["targets"].cpu(), want["targets"][:2], rtol=1e-5, atol=1e-5, what="transform.targets")
    # round trip: residual_to_hr(targets, lrinterp) == hr   (climex_utils.py:270-278)
    back = tr.residual_to_hr(item["targets"], item["lrinterp"])
    assert_close(back.cpu(), hr[:2], rtol=1e-4, atol=1e-4, what="residual_to_hr round trip")
    with pytest.raises(L.ProbUNetLibraryError):
        pa.data.lr_stats(hr, k)                                   # CPU tensor: no fallback


def test_trainer_call_pattern_matches_manual_loop():
    """train_probunet_step / eval_probunet_model (train_prob_unet_model.py:105-210) against the hand-written
    elbo -> zero_grad -> backward -> step sequence on a twin model: same parameters afterwards, same epoch means."""
    torch.manual_seed(0)
    def twin():
        m, _, _ = _small_model("afcrps", H=32)
        m.dropout = 0.0
        return m
    batches = []
    for i in range(3):
        x, y = make_fields(2, 2, 1, 32, 32, seed=20 + i)
        batches.append({"inputs": x, "targets": y, "timestamps": torch.tensor([1.0, 2.0])})
    a, b = twin(), twin()
    oa = torch.optim.AdamW(a.parameters(), lr=1e-3); ob = pa.FlatAdamW(b, lr=1e-3)
    torch.manual_seed(5)
    rec, kls = [], []
    a.train()
    for bt in batches:
        loss, rl, kl = a.elbo(bt["inputs"].to(DEV), bt["targets"].to(DEV), bt["timestamps"].unsqueeze(1).to(DEV), M=3)
        oa.zero_grad(); loss.backward(); oa.step()
        rec.append(rl[0]); kls.append(kl.mean().item())
    torch.manual_seed(5)
    mr, mk = pa.trainer.train_probunet_step(b, batches, ob, 1, 2, DEV, ensemble_size=3)
    assert abs(mr - float(np.mean(rec))) < 1e-5 * (1 + abs(mr)) and abs(mk - float(np.mean(kls))) < 1e-5 * (1 + abs(mk))
    for (k, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        assert_close(q.detach().cpu(), p.detach().cpu(), rtol=1e-4, atol=1e-6, what=k)
    assert b.sync_scalars is True                       # restored
    torch.manual_seed(6); er, ek = pa.trainer.eval_probunet_model(b, batches, DEV, ensemble_size=3)
    assert not b.training and np.isfinite(er) and np.isfinite(ek)
    assert all(p.grad is None or True for p in b.parameters())
    # epoch loop with the beta schedule of main.py:126-155
    hist = pa.trainer.fit(b, batches[:1], batches[:1], ob, num_epochs=4, device=DEV, ensemble_size=2, warmup_epochs=2)
    assert len(hist[0]) == 4 and len(hist[2]) == 4 and b.beta_0 == 1.0 and abs(b.beta_1 - 0.5) < 1e-12
    hr = pa.trainer.sample_probunet_model(b, {"inputs": batches[0]["inputs"], "lrinterp": batches[0]["targets"]}, DEV, num_samples=3,
                                          residual_std=torch.ones(1, 32, 32))
    assert tuple(hr.shape) == (2, 3, 1, 32, 32)


def test_short_training_run_f16_tracks_f32_and_loss_decreases():
    """25 AdamW steps of the trainer pattern on a small config from the same seed-42 initialisation: the loss goes down and the
    f16 engine's loss curve stays within 5 % of the fp32 engine's (dropout off so both see the same objective)."""
    curves = {}
    for dtype in ("f32", "f16"):
        torch.manual_seed(42)
        m = pa.ProbabilisticUNet(2, 1, 4, [8, 16, 32], 8, [1, 2, 4], 1.0, 0.1, 0.0, dtype=dtype).to(DEV).train()
        m.dropout = 0.0
        opt = pa.FlatAdamW(m, lr=2e-3)
        x, y = make_fields(4, 2, 1, 32, 32, seed=31)
        x, y = x.to(DEV), y.to(DEV)
        eps = make_eps(3, 4, 4).to(DEV)
        losses = []
        for _ in range(25):
            loss, _, _ = m.elbo(x, y, None, M=3, eps=eps)
            opt.zero_grad(); loss.backward(); opt.step()
            losses.append(float(loss.detach()))
        curves[dtype] = losses
    a, b = np.array(curves["f32"]), np.array(curves["f16"])
    assert a[-1] < 0.8 * a[0], (a[0], a[-1])
    assert np.all(np.abs(a - b) <= 0.05 * np.abs(a) + 1e-3), np.max(np.abs(a - b) / np.abs(a))


def test_latent_space_attributes_after_fused_elbo():
    """prob_unet.py:241-242: elbo() leaves prior_latent_space / posterior_latent_space behind; here they are fetched lazily
    from the engine and must equal what the sub-modules return for the same inputs (and reproduce the returned KL)."""
    m, cfg, P = _small_model("afcrps", H=32)
    m.eval()
    x, y = make_fields(2, 2, 1, 32, 32, seed=8)
    x, y = x.to(DEV), y.to(DEV)
    assert m.prior_latent_space is None and m.posterior_latent_space is None
    with torch.no_grad():
        _, _, kl = m.elbo(x, y, None, M=2, eps=make_eps(2, 2, 4).to(DEV))
        p, q = m.prior_latent_space, m.posterior_latent_space
        p2, q2 = m.prior(x), m.posterior(x, y)
    assert_close(p.base_dist.loc.cpu(), p2.base_dist.loc.cpu(), what="prior loc"); assert_close(p.base_dist.scale.cpu(), p2.base_dist.scale.cpu(), what="prior scale")
    assert_close(q.base_dist.loc.cpu(), q2.base_dist.loc.cpu(), what="post loc"); assert_close(q.base_dist.scale.cpu(), q2.base_dist.scale.cpu(), what="post scale")
    assert_close(torch.distributions.kl.kl_divergence(q, p).cpu(), kl.cpu(), what="kl from the attributes")
    m.prior_latent_space = "user value"                      # plain attribute semantics are kept
    assert m.prior_latent_space == "user value"


@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_kl_gradients_stay_finite_for_extreme_prior_sigma(dtype):
    """A prior log-sigma of 50 (sigma_p ~ 5e21; reached by Adam in tools/train_demo.py) squares to inf in fp32: the direct KL
    derivative forms give inf * 0 = NaN there (torch's included).  The engine evaluates them in ratios to sigma_p:
    d KL / d log sigma_p -> 1 - (sigma_q/sigma_p)^2 - ((mu_q-mu_p)/sigma_p)^2 -> 1, so the bias gradient is beta_1 exactly."""
    m, cfg, P = _small_model("afcrps", dtype=dtype, H=32)
    m.train(); m.dropout = 0.0
    with torch.no_grad():
        m.prior.conv_log_sigma.bias.fill_(50.0)
    x, y = make_fields(2, 2, 1, 32, 32, seed=12)
    total, _, kl = m.elbo(x.to(DEV), y.to(DEV), None, M=2, eps=make_eps(2, 2, 4).to(DEV))
    total.backward()
    assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)
    assert float(kl.min()) > 4 * 45.0          # 4 latent dims x log(sigma_p / sigma_q) ~ 50 (inf if (sigma_q/sigma_p)^2 underflows, as in torch)
    g = m.prior.conv_log_sigma.bias.grad.cpu()
    assert_close(g, torch.full_like(g, 1.3), rtol=1e-3, atol=1e-4, what="d total / d prior log-sigma bias = beta_1")


@pytest.mark.parametrize("M", [15, 24])
def test_ensemble_size_15_as_in_main_py(M):
    """main.py:133 trains with ensemble_size = 15 (the fused afCRPS kernel keeps up to 16, or up to 32, members in registers):
    fp32 engine vs the oracle, loss / KL / parameter gradients."""
    m, cfg, P = _small_model("afcrps", H=32)
    m.train(); m.dropout = 0.0
    x, y = make_fields(2, 2, 1, 32, 32, seed=14); eps = make_eps(M, 2, 4)
    total, recon, kl = m.elbo(x.to(DEV), y.to(DEV), None, M=M, eps=eps.to(DEV))
    total.backward()
    r, og = O.elbo_with_grads(P, cfg, x, y, eps, beta0=0.7, beta1=1.3)
    assert_close(total.detach().cpu(), r["total"], what="total"); assert_close(recon[0], r["recon"], what="afcrps"); assert_close(kl.cpu(), r["kl"], what="kl")
    rels = []
    for k, p in m.named_parameters():
        v = og[k]
        if float(v.norm()) < 1e-7: continue
        rels.append(float((p.grad.detach().cpu().double() - v.double()).norm() / v.double().norm()))
    assert max(rels) < 2e-2 and float(np.median(rels)) < 2e-3, (max(rels), float(np.median(rels)))
    with pytest.raises(pa._lib.ProbUNetLibraryError):
        m.elbo(x.to(DEV), y.to(DEV), None, M=33)


def test_assume_static_parameters_opt_in():
    """Inference loops may skip the per-call weight re-pack; optimizers / load_state_dict still force it."""
    m, cfg, P = _small_model("afcrps", dtype="f16", H=32)
    m.eval()
    x, _ = make_fields(2, 2, 1, 32, 32, seed=15); x = x.to(DEV); eps = make_eps(2, 2, 4).to(DEV)
    a = m.sample(x, 2, eps=eps).clone()
    m.assume_static_parameters = True
    b = m.sample(x, 2, eps=eps).clone()
    assert torch.equal(a, b)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    sd["fcomb.layers.4.bias"] = sd["fcomb.layers.4.bias"] + 1.0
    m.load_state_dict(sd)                                        # forces the re-pack (fcomb reads fp32 params directly; unet weights too)
    sd["unet.out_conv.weight"] = sd["unet.out_conv.weight"] * 0.5
    m.load_state_dict(sd)
    c = m.sample(x, 2, eps=eps)
    assert not torch.allclose(c, b + 1.0, atol=1e-3)             # the halved out_conv weights were re-packed and changed the features
    m2, _, _ = _small_model("afcrps", dtype="f16", H=32); m2.eval(); m2.load_state_dict(sd)
    assert torch.equal(m2.sample(x, 2, eps=eps), c)


def test_f16_overflow_is_flagged_and_the_step_skipped_on_device():
    """A loss scale far too high overflows the fp16 activation gradients: the engine flags non-finite parameter gradients
    (PU_S_NONFINITE) and the flat AdamW leaves parameters and moments untouched, without a host sync; with the automatic scale
    the flag stays 0 and the step is taken."""
    x, y = make_fields(2, 2, 1, 32, 32, seed=16); eps = make_eps(2, 2, 4)
    for scale, expect in ((2.0 ** 40, 1.0), (0.0, 0.0)):
        m, cfg, P = _small_model("afcrps", dtype="f16", H=32)
        m.grad_scale = scale
        m.train(); m.dropout = 0.0
        opt = pa.FlatAdamW(m, lr=1e-2)
        before = torch.cat([p.detach().flatten() for p in m.parameters()]).clone()
        loss, _, _ = m.elbo(x.to(DEV), y.to(DEV), None, M=2, eps=eps.to(DEV))
        opt.zero_grad(); loss.backward(); opt.step()
        after = torch.cat([p.detach().flatten() for p in m.parameters()])
        assert float(m._last_scalars[L.PU_S_NONFINITE]) == expect
        if expect:
            assert torch.equal(before, after) and float(opt.exp_avg.abs().sum()) == 0.0
        else:
            assert not torch.equal(before, after) and torch.isfinite(after).all()


@pytest.mark.parametrize("H,W", [(48, 80), (16, 144)])
def test_mixed_kernel_paths_on_odd_geometries(H, W):
    """Sizes whose levels fall on different convolution paths (16x16-tile conv3 at the top, the LDS-staged 8x8-tile kernel
    below; GroupNorm statistics from the conv epilogue at one level, from the stand-alone pass at the other): fp32 engine at
    the parity tolerance, f16 engine by feature error and gradient cosine."""
    cfg = O.Config(2, 1, 4, [8, 16], 8, [1, 2])
    P = filled_params(cfg)
    x, y = make_fields(2, 2, 1, H, W, seed=17); eps = make_eps(2, 2, 4)
    r, og = O.elbo_with_grads(P, cfg, x, y, eps, beta0=0.7, beta1=1.3)
    gref = torch.cat([og[k].flatten() for k in sorted(og)]).double()
    for dtype in ("f32", "f16"):
        m = pa.ProbabilisticUNet(2, 1, 4, [8, 16], 8, [1, 2], 0.7, 1.3, 0.0, dtype=dtype, init=False)
        m.load_state_dict(P); m = m.to(DEV).train(); m.dropout = 0.0
        total, recon, kl = m.elbo(x.to(DEV), y.to(DEV), None, M=2, eps=eps.to(DEV))
        total.backward()
        g = torch.cat([dict(m.named_parameters())[k].grad.flatten().cpu() for k in sorted(og)]).double()
        cos = float(g @ gref / (g.norm() * gref.norm()))
        with torch.no_grad():
            feat = m.unet(x.to(DEV)).cpu()
        if dtype == "f32":
            assert_close(total.detach().cpu(), r["total"], what="total"); assert_close(feat, r["feat"], what="feat")
            assert cos > 0.99999, cos
        else:
            assert float((feat - r["feat"]).abs().max()) < 3e-2 * float(r["feat"].abs().max()) + 1e-2 and cos > 0.995, cos


def test_engine_regrowth_keeps_parameters_gradients_and_optimizer_state():
    """The static plan is sized for a maximum batch; a larger batch re-creates the engine.  Parameters, their flat aliasing and the
    optimizer moments must survive that (both optimizers), and the smaller batch must keep working afterwards."""
    for make_opt in (lambda m: pa.FlatAdamW(m, lr=1e-3), lambda m: torch.optim.AdamW(m.parameters(), lr=1e-3)):
        m, cfg, P = _small_model("afcrps", dtype="f16", H=32)
        m.train(); m.dropout = 0.0
        opt = make_opt(m)
        losses = []
        for B in (2, 2, 5, 3, 5):
            x, y = make_fields(B, 2, 1, 32, 32, seed=18)
            loss, _, _ = m.elbo(x.to(DEV), y.to(DEV), None, M=2, eps=make_eps(2, B, 4).to(DEV))
            opt.zero_grad(); loss.backward(); opt.step()
            losses.append(float(loss.detach()))
            assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
        assert all(np.isfinite(losses)), losses
        ref, _, _ = _small_model("afcrps", dtype="f16", H=32)
        moved = sum(float((p.detach() - q.detach()).abs().sum()) for p, q in zip(m.parameters(), ref.parameters()))
        assert moved > 0.0
        # every parameter is still a view of the engine's flat buffer
        base = m._flat.data_ptr(); end = base + m._flat.numel() * 4
        assert all(base <= p.data_ptr() < end for p in m.parameters())


def test_gradient_accumulation_across_backward_calls():
    """autograd semantics of p.grad: a second backward without zero_grad adds; (2 * loss).backward() doubles."""
    m, cfg, P = _small_model("afcrps", H=32)
    m.train(); m.dropout = 0.0
    x, y = make_fields(2, 2, 1, 32, 32, seed=19); x, y = x.to(DEV), y.to(DEV); eps = make_eps(2, 2, 4).to(DEV)
    def flat():
        return torch.cat([p.grad.flatten() for p in m.parameters()]).clone()
    def same(a, b):                                        # float-atomic parameter gradients: order noise at the 1e-6 level
        return float((a - b).abs().max()) <= 2e-5 * float(b.abs().max())
    l, _, _ = m.elbo(x, y, None, M=2, eps=eps); l.backward(); g1 = flat()
    l, _, _ = m.elbo(x, y, None, M=2, eps=eps); l.backward(); g2 = flat()
    assert same(g2, 2 * g1)
    m.zero_grad(set_to_none=True)
    l, _, _ = m.elbo(x, y, None, M=2, eps=eps); (2.0 * l).backward(); g3 = flat()
    assert same(g3, 2 * g1)
    m.zero_grad(set_to_none=False)                         # zeros in place: the next backward accumulates into them
    l, _, _ = m.elbo(x, y, None, M=2, eps=eps); l.backward()
    assert same(flat(), g1)
