"""GPU: each hand-written HIP kernel family against a plain torch fp32 reference of the same op, through the
C ABI single-op entry points (include/probunet.h: pu_op_conv, pu_op_gnsilu).
fp32 path tolerance = north_star (rtol 1e-3, atol 1e-5 scaled); fp16/bf16 MFMA paths use dtype-appropriate bounds."""
import ctypes as C
import pytest, torch
import torch.nn.functional as F
import probunet_amd as pa
from probunet_amd import _lib as L

pytestmark = pytest.mark.gpu
TOL = {L.PU_F32: (1e-3, 1e-5), L.PU_F16: (1e-2, 4e-3), L.PU_BF16: (5e-2, 3e-2)}
DT = [L.PU_F32, L.PU_F16, L.PU_BF16]


def close(a, b, dt, what):
    rtol, atol = TOL[dt]
    a = a.double().cpu(); b = b.double().cpu()
    scale = max(1.0, float(b.abs().max()))
    err = (a - b).abs(); tol = atol * scale + rtol * b.abs()
    assert bool((err <= tol).all()), f"{what}: max err {float(err.max()):.3e} (scale {scale:.3g}), frac bad {float((err > tol).double().mean()):.3e}"


def q(t, dt):
    """Round to the kernel's storage dtype so that the reference sees the same inputs."""
    if dt == L.PU_F16: return t.half().float()
    if dt == L.PU_BF16: return t.bfloat16().float()
    return t


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("shape", [(2, 8, 16, 32, 32, 3), (1, 4, 32, 32, 64, 3), (2, 48, 24, 16, 16, 3), (3, 32, 64, 8, 8, 3),
                                   (2, 96, 32, 16, 32, 3), (2, 40, 16, 32, 32, 1), (1, 64, 128, 8, 8, 1), (1, 5, 8, 16, 16, 3),
                                   (1, 96, 128, 8, 32, 3), (1, 160, 96, 8, 64, 3),       # K > 64, Cout > 64, 32-wide: the dominant conv3 class
                                   (1, 512, 512, 16, 16, 3), (2, 256, 256, 32, 32, 3)])   # large dw: the all-taps units of the slab reduction (4 and 16 split lanes)
def test_conv_fwd_dgrad_wgrad(dt, shape):
    B, Cin, Cout, H, W, ks = shape
    dev = torch.device("cuda:0"); g = torch.Generator(device="cpu").manual_seed(B * 1000 + Cin * 10 + Cout + H)
    x = q(torch.randn(B, Cin, H, W, generator=g), dt).to(dev)
    w = q(torch.randn(Cout, Cin, ks, ks, generator=g) / (Cin * ks * ks) ** 0.5, dt).to(dev)
    bias = torch.randn(Cout, generator=g).to(dev)
    dy = q(torch.randn(B, Cout, H, W, generator=g), dt).to(dev)
    lib = L.lib(); st = L.current_stream()
    y = torch.empty(B, Cout, H, W, device=dev)
    for relu in (0, 1):
        L.check(lib.pu_op_conv(dt, 0, ks, relu, B, Cin, Cout, H, W, L.ptr(x), L.ptr(w), L.ptr(bias), None, L.ptr(y), st), None, "conv fwd")
        ref = F.conv2d(x, w, bias, padding=ks // 2)
        close(y, torch.relu(ref) if relu else ref, dt, f"conv fwd relu={relu}")
    dx = torch.empty(B, Cin, H, W, device=dev)
    L.check(lib.pu_op_conv(dt, 1, ks, 0, B, Cin, Cout, H, W, None, L.ptr(w), None, L.ptr(dy), L.ptr(dx), st), None, "dgrad")
    close(dx, torch.nn.grad.conv2d_input(x.shape, w, dy, padding=ks // 2), dt, "dgrad")
    dw = torch.empty(Cout, Cin, ks, ks, device=dev)
    L.check(lib.pu_op_conv(dt, 2, ks, 0, B, Cin, Cout, H, W, L.ptr(x), None, None, L.ptr(dy), L.ptr(dw), st), None, "wgrad")
    close(dw, torch.nn.grad.conv2d_weight(x, w.shape, dy, padding=ks // 2), dt, "wgrad")


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("rs", [0, 1, 2])
@pytest.mark.parametrize("shape", [(2, 32, 16, 16, True), (2, 48, 8, 8, False), (1, 192, 16, 16, True), (2, 8, 32, 32, True)])
def test_groupnorm_silu_fwd_bwd(dt, rs, shape):
    B, Cc, H, W, adaptive = shape
    dev = torch.device("cuda:0"); g = torch.Generator(device="cpu").manual_seed(Cc + H + rs)
    x = (q(torch.randn(B, Cc, H, W, generator=g) * 1.5 + 0.3, dt)).to(dev).requires_grad_(True)
    gamma = (1 + 0.2 * torch.randn(Cc, generator=g)).to(dev).requires_grad_(True)
    beta = (0.1 * torch.randn(Cc, generator=g)).to(dev).requires_grad_(True)
    ss = (0.2 * torch.randn(2 * Cc, generator=g)).to(dev).requires_grad_(True) if adaptive else None
    OH, OW = (H // 2, W // 2) if rs == 1 else ((H * 2, W * 2) if rs == 2 else (H, W))
    dy = q(torch.randn(B, Cc, OH, OW, generator=g), dt).to(dev)
    G = min(32, Cc // 4)
    h = F.group_norm(x, G, gamma, beta, 1e-5)
    if adaptive:
        h = h * (1 + ss[:Cc].view(1, -1, 1, 1)) + ss[Cc:].view(1, -1, 1, 1)
    h = F.silu(h)
    if rs == 1: h = F.avg_pool2d(h, 2)
    if rs == 2: h = h.repeat_interleave(2, 2).repeat_interleave(2, 3)
    h.backward(dy)
    lib = L.lib(); st = L.current_stream()
    y = torch.empty(B, Cc, OH, OW, device=dev); dx = torch.empty(B, Cc, H, W, device=dev)
    dg = torch.empty(Cc, device=dev); db = torch.empty(Cc, device=dev); dss = torch.empty(2 * Cc, device=dev)
    L.check(lib.pu_op_gnsilu(dt, rs, B, Cc, H, W, L.ptr(x.detach()), L.ptr(gamma.detach()), L.ptr(beta.detach()),
                             L.ptr(ss.detach()) if adaptive else None, L.ptr(y), L.ptr(dy), L.ptr(dx), L.ptr(dg), L.ptr(db), L.ptr(dss), 0.0, 0, st),
            None, "gnsilu")
    close(y, h.detach(), dt, "gn fwd")
    close(dx, x.grad, dt, "gn dx")
    close(dg, gamma.grad, dt, "gn dgamma")
    close(db, beta.grad, dt, "gn dbeta")
    if adaptive:
        close(dss, ss.grad, dt, "gn dscale/shift")


@pytest.mark.parametrize("dt", [L.PU_F32, L.PU_F16])
def test_dropout_mask_statistics_and_backward(dt):
    """UNetBlock dropout (networks.py:177): inverted dropout p=0.1 fused into GN+SiLU; the backward regenerates the
    same counter-hash mask.  Checked by injecting the observed mask into a torch reference (SURVEY §7 RNG parity)."""
    B, Cc, H, W, p = 2, 32, 32, 32, 0.1
    dev = torch.device("cuda:0"); g = torch.Generator(device="cpu").manual_seed(5)
    x = q(torch.randn(B, Cc, H, W, generator=g) + 2.0, dt).to(dev).requires_grad_(True)     # +2: silu output rarely exactly 0
    gamma = (1 + 0.2 * torch.randn(Cc, generator=g)).to(dev).requires_grad_(True)
    beta = (0.5 + 0.1 * torch.randn(Cc, generator=g)).to(dev).requires_grad_(True)
    dy = q(torch.randn(B, Cc, H, W, generator=g), dt).to(dev)
    lib = L.lib(); st = L.current_stream()
    outs = []
    for drop, seed in ((0.0, 0), (p, 123), (p, 124)):
        y = torch.empty(B, Cc, H, W, device=dev); dx = torch.empty_like(y)
        dg = torch.empty(Cc, device=dev); db = torch.empty(Cc, device=dev)
        L.check(lib.pu_op_gnsilu(dt, 0, B, Cc, H, W, L.ptr(x.detach()), L.ptr(gamma.detach()), L.ptr(beta.detach()), None, L.ptr(y), L.ptr(dy),
                                 L.ptr(dx), L.ptr(dg), L.ptr(db), None, drop, seed, st), None, "gnsilu")
        outs.append((y, dx, dg, db))
    y0, y1, y2 = outs[0][0], outs[1][0], outs[2][0]
    mask = (y1 != 0).float()
    frac = 1.0 - float(mask.mean())
    assert 0.08 < frac < 0.12, frac
    assert not torch.equal(y1 != 0, y2 != 0)                                   # different seed, different mask
    close(y1, y0 * mask / (1 - p), dt, "kept elements scaled by 1/(1-p)")
    h = F.silu(F.group_norm(x, 8, gamma, beta, 1e-5)) * mask / (1 - p)
    h.backward(dy)
    close(outs[1][1], x.grad, dt, "dropout dx"); close(outs[1][2], gamma.grad, dt, "dropout dgamma"); close(outs[1][3], beta.grad, dt, "dropout dbeta")
