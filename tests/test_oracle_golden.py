"""CPU: the oracle restatement vs golden vectors captured from the imported reference (tools/make_golden.py).
Tolerance = north_star's rtol 1e-3 / atol 1e-5 (fp32); gradients use atol scaled by the tensor's magnitude."""
import pytest, torch
from oracle import probunet_oracle as O
from tests.helpers import load_golden, oracle_cfg, filled_params, t, assert_close, GOLDEN_NAMES
from tests.filler import checksum


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_keys_and_shapes_match_reference(name):
    meta, _ = load_golden(name)
    cfg = oracle_cfg(meta["config"])
    mine = O.param_shapes(cfg)
    ref = [(k, tuple(s)) for k, s in meta["keys"]]
    assert [k for k, _ in ref] == list(mine.keys())
    for k, s in ref:
        if k == "posterior.encoder.0.weight":      # generalised (Cin+Cout)-plane stem vs reference 2*Cin
            assert s[1] == 2 * cfg.input_channels and mine[k][1] == cfg.input_channels + cfg.num_classes
            continue
        assert tuple(mine[k]) == s, k


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_filled_params_match_reference_fill(name):
    meta, _ = load_golden(name)
    P = filled_params(oracle_cfg(meta["config"]))
    for k, cs in meta["filled_checksums"].items():
        mine = checksum(P[k])
        assert abs(mine["sum"] - cs["sum"]) <= 1e-6 * max(1.0, cs["abssum"]), k


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_elbo_forward_and_grads(name):
    meta, g = load_golden(name)
    cfg = oracle_cfg(meta["config"])
    P = filled_params(cfg)
    r, grads = O.elbo_with_grads(P, cfg, t(g["x"]), t(g["y"]), t(g["eps"]),
                                 beta0=float(g["beta0"]), beta1=float(g["beta1"]), alpha=float(g["alpha"]))
    assert_close(r["feat"], g["feat"], what="feat")
    assert_close(r["mu_p"], g["mu_p"], what="mu_p")
    assert_close(O.sigma_of(r["ls_p"]), g["sig_p"], what="sig_p")
    assert_close(r["mu_q"], g["mu_q"], what="mu_q")
    assert_close(O.sigma_of(r["ls_q"]), g["sig_q"], what="sig_q")
    assert_close(r["preds"], g["preds"], what="preds")
    assert_close(r["kl"], g["kl"], what="kl")
    assert_close(r["recon"], g["crps"], what="crps")
    assert_close(r["total"], g["total"], what="total")
    for k, cs in meta["grad_checksums"].items():
        gr = grads[k]
        scale = cs["abssum"] / max(cs["n"], 1)
        key = "grad::" + k
        if key in g:
            assert_close(gr, g[key], rtol=1e-3, atol=1e-5 + 1e-3 * scale, what=key)
        mine = checksum(gr)
        assert abs(mine["abssum"] - cs["abssum"]) <= 2e-3 * cs["abssum"] + 1e-5 * cs["n"], k


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_l1_elbo_and_sampling(name):
    meta, g = load_golden(name)
    cfg = oracle_cfg(meta["config"])
    P = filled_params(cfg)
    with torch.no_grad():
        r = O.elbo(P, cfg, t(g["x"]), t(g["y"]), t(g["eps"])[:1], beta0=0.7, beta1=1.3, beta2=float(g["beta2"]), recon="l1")
        assert_close(r["recon"], g["l1"], what="l1")
        assert_close(r["kl2"], g["kl2"], what="kl2")
        assert_close(r["total"], g["total_l1"], what="total_l1")
        s = O.sample_forward(P, cfg, t(g["x"]), t(g["eps"])[:1])
        assert_close(s["out"][:, 0], g["sample_prior"], what="sample_prior")
        feat = O.unet_forward(P, cfg, t(g["x"]))
        K = g["grid_z"].shape[0]
        out = O.fcomb_forward(P, cfg, feat[:1].expand(K, -1, -1, -1), t(g["grid_z"]))
        assert_close(out, g["grid_out"], what="grid_out")


def test_afcrps_needs_two_members():
    meta, g = load_golden("tiny22")
    cfg = oracle_cfg(meta["config"])
    with pytest.raises(ValueError):
        O.elbo(filled_params(cfg), cfg, t(g["x"]), t(g["y"]), t(g["eps"])[:1], beta0=1, beta1=1)


def test_climex_transforms_match_reference():
    """Rows f3 / f4: tests/golden/climex.npz holds outputs of the imported src/climex_utils.py (tools/make_golden_climex.py):
    softplus / softplus_inv (:36-46), compute_stats (:255-264), the item transform (:197-250), invstand_residual and
    residual_to_hr (:270-285).  Pure fp32 arithmetic of a handful of ops: rtol 1e-6 on the statistics, exact elsewhere up to
    the last ulp of the division."""
    import os
    import numpy as np
    from tests.helpers import GOLDEN
    g = dict(np.load(os.path.join(GOLDEN, "climex.npz")))
    k = int(g["k"]); hr = t(g["hr"])
    assert_close(O.softplus_climex(t(g["sp_in"])), g["sp_out"], rtol=1e-6, atol=1e-7, what="softplus")
    assert_close(O.softplus_climex(t(g["sp_in"]), c=0.), g["sp_out_c0"], rtol=1e-6, atol=1e-7, what="softplus c=0")
    assert_close(O.softplus_inv_climex(t(g["spinv_in"])), g["spinv_out"], rtol=1e-6, atol=1e-7, what="softplus_inv")
    assert_close(O.softplus_inv_climex(t(g["spinv_in"]), c=0.), g["spinv_out_c0"], rtol=1e-6, atol=1e-7, what="softplus_inv c=0")
    (ml, sl), (mh, sh) = O.lr_stats(hr, k)
    for mine, key in ((ml, "mean_lr"), (sl, "std_lr"), (mh, "mean_hr"), (sh, "std_hr")):
        assert_close(mine, g[key], rtol=1e-6, atol=1e-7, what=key)
    it = O.lrinterp_to_residuals(hr[:3], k, mh, sh, 1e-10)
    for key in ("inputs", "targets", "lrinterp", "lr"):
        assert_close(it[key], g["item_" + key], rtol=1e-6, atol=1e-6, what="item " + key)
    assert_close(O.invstand_residual(t(g["item_targets"]), mh, sh), g["invstand_residual"], rtol=1e-6, atol=1e-7, what="invstand_residual")
    assert_close(O.residual_to_hr(t(g["item_targets"]), t(g["item_lrinterp"]), sh), g["residual_to_hr"], rtol=1e-6, atol=1e-6, what="residual_to_hr")
    assert_close(O.residual_to_hr(t(g["pred_residual"]), t(g["item_lrinterp"]), sh), g["pred_to_hr"], rtol=1e-6, atol=1e-6, what="residual_to_hr 5-D")
    assert_close(O.invstand_residual(t(g["hrkind_targets"]), mh, sh, kind="lrinterp_to_hr"), g["hrkind_invstand"], rtol=1e-6, atol=1e-6,
                 what="invstand (hr kind)")
    # the hr-kind item: inputs are the same standardised lrinterp, targets the standardised hr
    assert_close(it["inputs"][:2], g["hrkind_inputs"], rtol=1e-6, atol=1e-6, what="hr-kind inputs")
    assert_close(it["inputs"][:2] + it["targets"][:2], g["hrkind_targets"], rtol=1e-5, atol=1e-5, what="hr-kind targets")
