"""CPU (no GPU): host logic of the drop-in class and the C-ABI surface.
 - state_dict keys / shapes / order and seed-42 initialisation equal the reference's (golden init checksums)
 - libprobunet.so loads and exports every symbol include/probunet.h declares (no compute calls)
 - error behaviour without a device; load_state_dict of a reference-shaped (2*Cin plane) posterior stem."""
import ctypes, os, re
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
    hdr = open(os.path.join(ROOT, "include", "probunet.h")).read()
    names = sorted(set(re.findall(r"\b(pu_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 20
    lib = ctypes.CDLL(L.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/probunet.h but not exported"
    assert L.lib().pu_abi_version() == 1
    assert ctypes.sizeof(L.PuConfig) == 4 * (4 + 8 + 1 + 8 + 2 + 2 + 1) + 4 + 4
    assert ctypes.sizeof(L.PuParamDesc) == 96 + 4 + 16 + 4 + 8 + 8 or ctypes.sizeof(L.PuParamDesc) % 8 == 0
