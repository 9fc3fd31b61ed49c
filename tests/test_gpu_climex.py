"""GPU: rows f3 / f4 against outputs of the IMPORTED reference (tests/golden/climex.npz, tools/make_golden_climex.py):
pu_lr_stats, pu_lrinterp_to_residuals, pu_softplus_transform, pu_destandardize and the fused residual_to_hr store of
pu_sample_hr vs src/climex_utils.py:36-46, :197-250, :255-285.  fp32 arithmetic of a handful of ops: rtol 1e-6 (statistics 1e-5:
the device keeps Welford sums in fp64, torch sums fp32)."""
import os
import numpy as np
import pytest, torch
import probunet_amd as pa
from tests.helpers import GOLDEN, assert_close, t

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def g():
    return dict(np.load(os.path.join(GOLDEN, "climex.npz")))


def test_statistics_and_item_transform_match_reference(g):
    k = int(g["k"]); hr = t(g["hr"]).to(DEV)
    (ml, sl), (mh, sh) = pa.data.lr_stats(hr, k)
    for mine, key in ((ml, "mean_lr"), (sl, "std_lr"), (mh, "mean_hr"), (sh, "std_hr")):
        assert_close(mine.cpu(), g[key], rtol=1e-5, atol=1e-6, what=key)
    mh_ref, sh_ref = t(g["mean_hr"]).to(DEV), t(g["std_hr"]).to(DEV)
    it = pa.data.lrinterp_to_residuals(hr[:3], k, mh_ref, sh_ref, 1e-10)
    for key in ("inputs", "targets", "lrinterp", "lr"):
        assert_close(it[key].cpu(), g["item_" + key], rtol=1e-6, atol=2e-6, what="item " + key)
    assert torch.equal(it["hr"].cpu(), t(g["item_hr"]))
    tr = pa.data.ClimExTransform(lowres_scale=k).fit(hr)
    item = tr(hr[:3], timestamps=t(g["item_timestamps"]).to(DEV))
    assert_close(item["targets"].cpu(), g["item_targets"], rtol=1e-5, atol=1e-5, what="transform targets")
    assert_close(tr.invstand_residual(t(g["item_targets"]).to(DEV)).cpu(), g["invstand_residual"], rtol=1e-5, atol=1e-6, what="invstand_residual")
    assert_close(tr.residual_to_hr(t(g["item_targets"]).to(DEV), t(g["item_lrinterp"]).to(DEV)).cpu(), g["residual_to_hr"], rtol=1e-5, atol=1e-5,
                 what="residual_to_hr")
    assert_close(tr.residual_to_hr(t(g["pred_residual"]).to(DEV), t(g["item_lrinterp"]).to(DEV)).cpu(), g["pred_to_hr"], rtol=1e-5, atol=1e-5,
                 what="residual_to_hr 5-D")
    tr2 = pa.data.ClimExTransform(lowres_scale=k, kind="lrinterp_to_hr").fit(hr)
    it2 = tr2(hr[:2])
    assert_close(it2["inputs"].cpu(), g["hrkind_inputs"], rtol=1e-5, atol=1e-5, what="hr-kind inputs")
    assert_close(it2["targets"].cpu(), g["hrkind_targets"], rtol=1e-5, atol=1e-5, what="hr-kind targets")
    assert_close(tr2.invstand_residual(t(g["hrkind_targets"]).to(DEV)).cpu(), g["hrkind_invstand"], rtol=1e-5, atol=1e-5, what="hr-kind invstand")


def test_softplus_pair_matches_reference(g):
    for c, suf in ((1e-7, ""), (0.0, "_c0")):
        v = t(g["sp_in"]).to(DEV).clone()
        assert pa.data.softplus(v, c=c) is v                      # in place, like the reference
        assert_close(v.cpu(), g["sp_out" + suf], rtol=1e-6, atol=1e-7, what="softplus" + suf)
        w = t(g["spinv_in"]).to(DEV).clone()
        pa.data.softplus_inv(w, c=c)
        assert_close(w.cpu(), g["spinv_out" + suf], rtol=2e-6, atol=2e-6, what="softplus_inv" + suf)
    # round trip: softplus(softplus_inv(x)) == x (the load-time transform and its inverse after sampling)
    x = torch.linspace(0.05, 30.0, 200, device=DEV)
    y = pa.data.softplus(pa.data.softplus_inv(x.clone()))
    assert_close(y.cpu(), x.cpu(), rtol=1e-5, atol=1e-5, what="round trip")
    with pytest.raises(pa._lib.ProbUNetLibraryError):
        pa.data.softplus(torch.zeros(3))                          # CPU tensor: no fallback


def test_fused_sample_hr_uses_reference_destandardisation(g):
    """pu_sample_hr's epilogue == ClimExDataset.residual_to_hr applied to pu_sample's residuals with the reference's own
    statistics (std_hr of the golden file)."""
    sh = t(g["std_hr"])[:1, :32, :32].contiguous()
    m = pa.ProbabilisticUNet(3, 1, 4, [8, 16], 8, [1, 2], 1.0, 1.0, 0.0, dtype="f32")
    from tests.filler import make_fields, make_eps
    x, _ = make_fields(2, 3, 1, 32, 32, seed=9); eps = make_eps(3, 2, 4)
    m = m.to(DEV).eval()
    with torch.no_grad():
        for p in m.parameters():
            if float(p.abs().sum()) == 0.0 and p.dim() == 4:
                p.copy_(0.05 * torch.sin(torch.arange(p.numel(), dtype=torch.float32)).view(p.shape))
    lrint = t(g["item_lrinterp"])[:2, :1].contiguous()
    res = m.sample(x.to(DEV), 3, eps=eps.to(DEV))
    want = pa.data.destandardize(res, sh.to(DEV), lrint.to(DEV))
    got = m.sample_hr(x.to(DEV), 3, lrint.to(DEV), sh.to(DEV), eps=eps.to(DEV))
    assert_close(got.cpu(), want.cpu(), rtol=1e-6, atol=1e-6, what="sample_hr vs destandardize(sample)")
    ref = lrint.unsqueeze(1) + res.cpu() * (sh + 1e-10)               # the reference's two lines (climex_utils.py:270-278)
    assert_close(got.cpu(), ref, rtol=1e-6, atol=1e-6, what="sample_hr vs reference formula")
