"""CPU (no GPU): host logic of the drop-in class and the C-ABI surface.
 - state_dict keys / shapes / order and seed-42 initialisation equal the reference's (golden init checksums)
 - libprobunet.so loads and exports every symbol include/probunet.h declares (no compute calls)
 - error behaviour without a device; load_state_dict of a reference-shaped (2*Cin plane) posterior stem."""
import ctypes, os, re
import numpy as np
import pytest, torch
import probunet_amd as pa
from probunet_amd import _lib as L
from tests.helpers import load_golden, GOLDEN_NAMES, oracle_cfg, filled_params
from tests.filler import checksum
from oracle import probunet_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def model_from(meta, **kw):
    c = meta["config"]
    return pa.ProbabilisticUNet(c["input_channels"], c["num_classes"], c["latent_dim"], c["num_filters"], c["model_channels"],
                                c["channel_mult"], 1.0, 1.0, 0.0, **kw)


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_state_dict_matches_reference(name):
    meta, _ = load_golden(name)
    torch.manual_seed(42)
    m = model_from(meta)
    sd = m.state_dict()
    assert list(sd.keys()) == [k for k, _ in meta["keys"]]
    cin = meta["config"]["input_channels"]
    for k, shp in meta["keys"]:
        if k == "posterior.encoder.0.weight":
            assert shp[1] == 2 * cin and sd[k].shape[1] == cin + meta["config"]["num_classes"]
        else:
            assert list(sd[k].shape) == shp, k
    # seed-exact initialisation replay (networks.py:21-26,243-297; prob_unet_utils.py:10-23)
    for k, cs in meta["init_seed42"].items():
        if k == "posterior.encoder.0.weight" and sd[k].shape[1] != 2 * cin:
            continue
        mine = checksum(sd[k])
        assert abs(mine["sum"] - cs["sum"]) <= 1e-6 * max(1.0, cs["abssum"]), k
        assert mine["sample"] == pytest.approx(cs["sample"], rel=1e-6, abs=1e-9), k
    # zero-initialised tensors and buffers
    assert float(sd["unet.out_conv.weight"].abs().max()) == 0.0
    assert torch.all(sd["unet.enc.64x64_down.conv0.resample_filter"] == 0.25)


def test_host_table_equals_oracle_layout():
    cfg = O.Config(4, 1, 12, [32, 64, 128, 256, 512], 32, [1, 2, 4, 8, 16])
    m = pa.ProbabilisticUNet(4, 1, 12, [32, 64, 128, 256, 512], 32, [1, 2, 4, 8, 16], 1, 1, 0, init=False)
    shapes = O.param_shapes(cfg)
    sd = m.state_dict()
    assert list(sd.keys()) == list(shapes.keys())
    assert all(tuple(sd[k].shape) == tuple(v) for k, v in shapes.items())
    n = sum(p.numel() for p in m.parameters())
    assert n == 75_677_489 - 864          # cfg3 total with the generalised 5-plane posterior stem
    assert len(sd) == 485


def test_module_surface_used_by_callers():
    meta, _ = load_golden("tiny22")
    m = model_from(meta, init=False)
    assert m.fcomb.layers[0].weight.shape == (8, 14, 1, 1)                 # latent_exploration_posterior.py:200
    assert m.prior.encoder[7].weight.shape[0] == 16
    t = m.fcomb.tile(torch.arange(6.).reshape(2, 3, 1), 2, 4)              # latent_exploration.py:524
    assert t.shape == (2, 3, 4) and torch.equal(t[..., 0], t[..., 3])
    m.beta_0 = 0.3; m.latent_dim                                          # main.py:117-123 mutate these attributes
    assert m.training and not m.eval().training
    hr = pa.ProbabilisticUNet.reconstruct(torch.ones(1, 1, 2, 2), torch.full((1, 1, 2, 2), 2.0), torch.full((1, 2, 2), 0.5))
    assert torch.allclose(hr, torch.full((1, 1, 2, 2), 2.5))               # climex_utils.py:277-285


def test_load_reference_shaped_checkpoint():
    meta, _ = load_golden("tiny41")
    m = model_from(meta, init=False)
    cfg = oracle_cfg(meta["config"])
    sd = filled_params(cfg)
    w = sd["posterior.encoder.0.weight"]
    ref_shaped = torch.cat([w, torch.randn(w.shape[0], 2 * cfg.input_channels - w.shape[1], 3, 3)], dim=1)   # 8 planes
    sd["posterior.encoder.0.weight"] = ref_shaped
    m.load_state_dict(sd)
    assert torch.equal(m.state_dict()["posterior.encoder.0.weight"], w)


def test_no_cpu_fallback():
    meta, g = load_golden("tiny22")
    m = model_from(meta)
    x = torch.zeros(1, 2, 32, 32)
    with pytest.raises(L.ProbUNetLibraryError):
        m.elbo(x, x, None, M=2)
    with pytest.raises(L.ProbUNetLibraryError):
        m.unet(x)
    with pytest.raises(ValueError):
        pa.ProbabilisticUNet(2, 2, 6, [8, 16], 8, [1, 2, 4], 1, 1, 0)


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(L.LIB_PATH)
    for header, least in (("probunet.h", 35), ("probunet_testing.h", 4)):       # product ABI; test / micro-benchmark hooks
        hdr = open(os.path.join(ROOT, "include", header)).read()
        names = sorted(set(re.findall(r"\b(pu_[a-z0-9_]+)\s*\(", hdr)))
        assert len(names) >= least, (header, len(names))
        for n in names:
            assert hasattr(lib, n), f"{n} declared in include/{header} but not exported"
    assert "pu_op_conv" not in open(os.path.join(ROOT, "include", "probunet.h")).read().replace("pu_op_*", "")
    assert L.lib().pu_abi_version() == 2
    assert ctypes.sizeof(L.PuConfig) == 4 * (4 + 8 + 1 + 8 + 2 + 2 + 1) + 4 + 4
    assert ctypes.sizeof(L.PuParamDesc) == 96 + 4 + 16 + 4 + 8 + 8 or ctypes.sizeof(L.PuParamDesc) % 8 == 0


# ------------------------------------------------------------------ rows f1-f4 (host logic + oracle pins, no GPU)
def test_wmse_half_of_live_loss_matches_reference_golden():
    """wmse_ms_ssim_loss (prob_unet_utils.py:270-305) with the absent third-party ms_ssim replaced by the same recording
    constant tools/make_golden.py used: WMSE, lam-combination, inferred data_range, gradient, and the ensemble-mean rule."""
    import json, os
    from tests.filler import make_fields, checksum
    from tests.helpers import GOLDEN
    g = json.load(open(os.path.join(GOLDEN, "wmse.json")))
    for tag, d in g.items():
        x, y = make_fields(d["B"], d["C"], d["C"], d["H"], d["H"], seed=77)
        pred = (0.8 * y + 0.3 * x).requires_grad_(True); tgt = 3.0 * y + 1.0
        seen = {}
        def fake(p, t_, data_range=None, win_size=None, size_average=True):
            seen.update(dr=data_range, win=win_size); return torch.tensor(0.25)
        loss, wmse, msl = O.wmse_ms_ssim_loss(pred, tgt, d["alpha"], d["beta"], d["lam"], ms_ssim_fn=fake)
        loss.backward()
        assert abs(float(loss) - d["loss"]) < 1e-6 * (1 + abs(d["loss"])) and abs(float(wmse) - d["wmse"]) < 1e-6 * (1 + d["wmse"])
        assert abs(seen["dr"] - d["data_range"]) < 1e-6 and seen["win"] == d["win_size"] == 7
        c = checksum(pred.grad)
        assert abs(c["sum"] - d["grad"]["sum"]) < 1e-6 and abs(c["abssum"] - d["grad"]["abssum"]) < 1e-6
        ens = torch.stack([pred.detach(), pred.detach() * 0.5], dim=1)
        _, w5, _ = O.wmse_ms_ssim_loss(ens, tgt, d["alpha"], d["beta"], d["lam"], ms_ssim_fn=fake)
        assert abs(float(w5) - d["wmse_ens_mean"]) < 1e-6 * (1 + d["wmse_ens_mean"])


def test_ms_ssim_restatement_known_answers():
    """PARITY UNPINNED arithmetic (pytorch-msssim 1.0.0 absent): properties of the published algorithm that do not need the
    package - identity, symmetry, constant images (cs = 1, ms = l^0.1333), the size assertion, window normalisation."""
    a = torch.rand(2, 1, 128, 112, generator=torch.Generator().manual_seed(1))
    b = torch.rand(2, 1, 128, 112, generator=torch.Generator().manual_seed(2))
    assert abs(float(O.ms_ssim(a, a, 1.0)) - 1.0) < 1e-6
    assert abs(float(O.ms_ssim(a, b, 1.0)) - float(O.ms_ssim(b, a, 1.0))) < 1e-6
    assert 0.0 <= float(O.ms_ssim(a, b, 1.0)) < 0.5
    c1, c2, R = 0.8, 0.5, 2.0
    v = float(O.ms_ssim(torch.full((1, 1, 128, 128), c1), torch.full((1, 1, 128, 128), c2), R))   # 128 -> 8: no padded (odd) level
    C1 = (0.01 * R) ** 2
    assert abs(v - ((2 * c1 * c2 + C1) / (c1 * c1 + c2 * c2 + C1)) ** 0.1333) < 1e-4      # fp32 E[x^2] - mu^2 noise against C2
    with pytest.raises(AssertionError):
        O.ms_ssim(a[..., :96, :96], b[..., :96, :96], 1.0)
    w = O.gauss_window()
    assert w.numel() == 7 and abs(float(w.sum()) - 1.0) < 1e-6 and float(w[3]) == float(w.max())


def test_beta_schedule_matches_main_loop():
    """main.py:108-155 replayed literally for num_epochs = 10: betas used for epochs 1..10."""
    import probunet_amd as pa
    used, b0, b1 = [], 1.0, 0.0
    for epoch in range(1, 11):
        used.append((b0, b1))
        if epoch <= 2: b0, b1 = 1.0, 0.0
        else: b0, b1 = 1.0, min((epoch - 2) / (10 - 2), 1.0) * 1.0
    mine, cur = [], pa.trainer.beta_schedule(0, 10)
    for epoch in range(1, 11):
        mine.append(cur); cur = pa.trainer.beta_schedule(epoch, 10)
    assert mine == used and pa.trainer.beta_schedule(10, 10) == (1.0, 1.0)


def test_climex_oracle_transforms_round_trip():
    hr = torch.randn(6, 2, 16, 24, generator=torch.Generator().manual_seed(3)) + 2.0
    (ml, sl), (mh, sh) = O.lr_stats(hr, 4)
    assert ml.shape == (2, 4, 6) and mh.shape == (2, 16, 24) and torch.equal(mh[:, ::4, ::4], ml)
    it = O.lrinterp_to_residuals(hr, 4, mh, sh)
    back = O.residual_to_hr(it["targets"], it["lrinterp"], sh)
    assert float((back - hr).abs().max()) < 1e-4
    assert torch.equal(O.softplus_climex(torch.tensor([25.0])), torch.tensor([25.0]))
    assert abs(float(O.softplus_climex(torch.tensor([0.0]))) - (np.log(2.0) - 1e-7)) < 1e-6


def test_data_module_has_no_cpu_fallback():
    import probunet_amd as pa
    from probunet_amd import _lib as L_
    with pytest.raises(L_.ProbUNetLibraryError):
        pa.data.lrinterp_to_residuals(torch.zeros(1, 1, 8, 8), 2, torch.zeros(1, 8, 8), torch.ones(1, 8, 8))
    with pytest.raises(ValueError):
        pa.ProbabilisticUNet(1, 1, 2, [8], 8, [1], 1.0, 1.0, 0.0, recon="ssim")


def test_leaf_module_metadata_used_by_the_exploration_scripts():
    """latent_exploration.py:296 / latent_exploration_posterior.py:200-202: `conv0 = model.fcomb.layers[0]; conv0.in_channels -
    model.latent_dim` must give the feature width."""
    import probunet_amd as pa
    m = pa.ProbabilisticUNet(4, 1, 6, [32, 64], 32, [1, 2], 1.0, 1.0, 0.0)
    conv0 = m.fcomb.layers[0]
    assert conv0.in_channels - m.latent_dim == 32 and conv0.out_channels == 32 and conv0.kernel_size == (1, 1)
    assert m.prior.encoder[0].in_channels == 4 and m.posterior.encoder[0].in_channels == 5 and m.prior.encoder[0].kernel_size == (3, 3)
    assert m.prior.conv_mu.out_channels == 6 and m.unet.out_conv.out_channels == 32
