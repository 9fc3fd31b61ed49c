"""GPU: gradient delivery without copies (round 3).  `p.grad` are views of the engine's flat gradient buffer; the engine owns two such
buffers and is re-bound to the other one (`pu_bind_grads`) while views of the first are still referenced, so both call orders of the
reference's trainer work and a gradient kept across the NEXT step stays intact:

  * `elbo -> zero_grad -> backward -> step` (train_prob_unet_model.py:133-140) and `zero_grad -> elbo -> backward -> step` give the same
    parameters, with FlatAdamW and with torch.optim.AdamW;
  * gradients still referenced when the next elbo() runs are not overwritten by it;
  * accumulation over two backward() calls without zero_grad() equals the sum of the two gradients;
  * the delivered values are the engine's (pu_scale_grads applies the upstream factor of `(2 * loss).backward()` once).
"""
import pytest, torch
import probunet_amd as pa
from oracle import probunet_oracle as O
from tests.helpers import filled_params
from tests.filler import make_fields, make_eps

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _small(dtype):
    cfg = O.Config(2, 1, 4, [8, 16], 8, [1, 2])
    m = pa.ProbabilisticUNet(2, 1, 4, [8, 16], 8, [1, 2], 0.7, 1.3, 0.0, dtype=dtype, init=False)
    m.load_state_dict(filled_params(cfg))
    m = m.to(DEV).train(); m.dropout = 0.0
    return m


def _batch(seed):
    x, y = make_fields(2, 2, 1, 32, 32, seed=seed)
    return x.to(DEV), y.to(DEV), make_eps(2, 2, 4).to(DEV)


def _flat(m):
    return torch.cat([p.detach().flatten() for p in m.parameters()]).clone()


def _grads(m):
    return torch.cat([p.grad.detach().flatten() for p in m.parameters()]).clone()


@pytest.mark.parametrize("dtype", ["f32", "f16"])
@pytest.mark.parametrize("optim", ["flat", "torch"])
def test_both_call_orders_of_the_trainer_give_the_same_parameters(dtype, optim):
    out = []
    for order in ("elbo_first", "zero_first"):
        m = _small(dtype)
        opt = pa.FlatAdamW(m, lr=1e-3) if optim == "flat" else torch.optim.AdamW(m.parameters(), lr=1e-3)
        for step in range(4):
            x, y, eps = _batch(10 + step)
            if order == "elbo_first":
                loss = m.elbo(x, y, None, M=2, eps=eps)[0]; opt.zero_grad(); loss.backward()
            else:
                opt.zero_grad(); loss = m.elbo(x, y, None, M=2, eps=eps)[0]; loss.backward()
            opt.step()
        torch.cuda.synchronize()
        out.append(_flat(m).cpu())
    assert torch.isfinite(out[0]).all()
    # same kernels, same inputs: the two orders differ only in which of the two gradient buffers a step writes
    # (float atomics in the GroupNorm / Fcomb parameter gradients leave order noise at the last bits)
    assert torch.allclose(out[0], out[1], rtol=2e-5, atol=1e-6), float((out[0] - out[1]).abs().max())
    assert float((out[0] - _flat(_small(dtype)).cpu()).abs().max()) > 1e-4        # and the parameters did move


@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_gradients_still_referenced_survive_the_next_elbo(dtype):
    m = _small(dtype)
    x, y, eps = _batch(3)
    m.elbo(x, y, None, M=2, eps=eps)[0].backward()
    held = [p.grad for p in m.parameters()]                      # views of engine buffer 0, kept alive
    snap = _grads(m)
    x2, y2, eps2 = _batch(4)
    loss2 = m.elbo(x2 * 0.5, y2, None, M=2, eps=eps2)[0]          # fused forward + backward: must write the OTHER buffer
    torch.cuda.synchronize()
    assert torch.equal(torch.cat([g.detach().flatten() for g in held]), snap)
    for p in m.parameters():
        p.grad = None
    loss2.backward()
    g2 = _grads(m)
    assert not torch.allclose(g2, snap)
    assert torch.equal(torch.cat([g.detach().flatten() for g in held]), snap)     # still intact after the second delivery


@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_accumulation_over_two_backward_calls_is_the_sum(dtype):
    m = _small(dtype)
    xa, ya, ea = _batch(5); xb, yb, eb = _batch(6)
    m.elbo(xa, ya, None, M=2, eps=ea)[0].backward(); ga = _grads(m); m.zero_grad(set_to_none=True)
    m.elbo(xb, yb, None, M=2, eps=eb)[0].backward(); gb = _grads(m); m.zero_grad(set_to_none=True)
    m.elbo(xa, ya, None, M=2, eps=ea)[0].backward()
    m.elbo(xb, yb, None, M=2, eps=eb)[0].backward()               # no zero_grad in between: p.grad += second gradient
    gs = _grads(m)
    tol = 1e-5 if dtype == "f32" else 2e-3
    assert torch.allclose(gs, ga + gb, rtol=tol, atol=tol * float((ga + gb).abs().max())), float((gs - ga - gb).abs().max())


def test_upstream_factor_is_applied_once():
    m = _small("f32")
    x, y, eps = _batch(7)
    m.elbo(x, y, None, M=2, eps=eps)[0].backward(); g1 = _grads(m); m.zero_grad(set_to_none=True)
    (2.0 * m.elbo(x, y, None, M=2, eps=eps)[0]).backward(); g2 = _grads(m)
    assert torch.allclose(g2, 2.0 * g1, rtol=1e-5, atol=1e-7 * float(g1.abs().max()) + 1e-12)
