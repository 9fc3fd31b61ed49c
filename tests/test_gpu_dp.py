"""GPU: the data-parallel MODEL path (enable_data_parallel -> broadcast, all-reduce inside loss.backward(), flat AdamW) with two
ranks that share the one GPU of the box (backend gloo: RCCL refuses two ranks on one device; the collective semantics - SUM
then 1/world folded into the delivery pass - are the same).  Checks: parameters are broadcast from rank 0, the delivered
gradient equals the single-process gradient of the global batch, both ranks hold identical parameters after the step."""
import os, sys, tempfile
import pytest, torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, initfile, outdir):
    sys.path.insert(0, ROOT)
    import probunet_amd as pa
    from tests.filler import make_fields, make_eps
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"file://{initfile}", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)                                  # different initialisations: the broadcast must fix that
    m = pa.ProbabilisticUNet(2, 1, 4, [8, 16], 8, [1, 2], 0.7, 1.3, 0.0, dtype="f32").to(dev).train()
    m.dropout = 0.0
    m.enable_data_parallel()
    x, y = make_fields(4, 2, 1, 32, 32, seed=40); eps = make_eps(2, 4, 4)
    xs, ys = pa.dp.shard_batch(x, rank, world).contiguous().to(dev), pa.dp.shard_batch(y, rank, world).contiguous().to(dev)
    es = eps[:, rank * 2:(rank + 1) * 2].contiguous().to(dev)
    opt = pa.FlatAdamW(m, lr=1e-3)
    loss, _, _ = m.elbo(xs, ys, None, M=2, eps=es)
    opt.zero_grad(); loss.backward()
    g = torch.cat([p.grad.flatten() for p in m.parameters()]).cpu()
    p0 = torch.cat([p.detach().flatten() for p in m.parameters()]).cpu()
    opt.step()
    p1 = torch.cat([p.detach().flatten() for p in m.parameters()]).cpu()
    # accumulate path of _deliver (p.grad already set): a second backward doubles the gradient, still averaged over ranks
    loss2, _, _ = m.elbo(xs, ys, None, M=2, eps=es)
    loss2.backward()
    torch.save(dict(g=g, p0=p0, p1=p1), os.path.join(outdir, f"r{rank}.pt"))
    if rank == 0:                                                  # single-process reference: same start parameters, global batch
        dist.barrier()
        ref = pa.ProbabilisticUNet(2, 1, 4, [8, 16], 8, [1, 2], 0.7, 1.3, 0.0, dtype="f32", init=False).to(dev).train()
        ref.dropout = 0.0
        with torch.no_grad():
            off = 0
            for p in ref.parameters():
                p.copy_(p0[off:off + p.numel()].view(p.shape).to(dev)); off += p.numel()
        l, _, _ = ref.elbo(x.to(dev), y.to(dev), None, M=2, eps=eps.to(dev))
        l.backward()
        torch.save(dict(g=torch.cat([p.grad.flatten() for p in ref.parameters()]).cpu()), os.path.join(outdir, "ref.pt"))
    else:
        dist.barrier()
    dist.destroy_process_group()


def _worker_overflow(rank, world, initfile, outdir):
    sys.path.insert(0, ROOT)
    import probunet_amd as pa
    from probunet_amd import _lib as L
    from tests.filler import make_fields, make_eps
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"file://{initfile}", rank=rank, world_size=world)
    torch.manual_seed(7)
    m = pa.ProbabilisticUNet(2, 1, 4, [8, 16], 8, [1, 2], 0.7, 1.3, 0.0, dtype="f16", grad_scale=(2.0 ** 40 if rank == 1 else 0.0)).to(dev).train()
    m.dropout = 0.0
    m.enable_data_parallel()
    x, y = make_fields(4, 2, 1, 32, 32, seed=41); eps = make_eps(2, 4, 4)
    xs, ys = pa.dp.shard_batch(x, rank, world).contiguous().to(dev), pa.dp.shard_batch(y, rank, world).contiguous().to(dev)
    opt = pa.FlatAdamW(m, lr=1e-2)
    p0 = torch.cat([p.detach().flatten() for p in m.parameters()]).cpu()
    loss, _, _ = m.elbo(xs, ys, None, M=2, eps=eps[:, rank * 2:(rank + 1) * 2].contiguous().to(dev))
    opt.zero_grad(); loss.backward(); opt.step()
    p1 = torch.cat([p.detach().flatten() for p in m.parameters()]).cpu()
    torch.save(dict(p0=p0, p1=p1, flag=float(m._last_scalars[L.PU_S_NONFINITE])), os.path.join(outdir, f"o{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_overflow_on_one_rank_makes_every_rank_skip():
    """f16 engine, rank 1 overflows (absurd loss scale), rank 0 does not: the flag is MAX-reduced with the gradients, so both
    ranks leave their (identical) parameters untouched instead of rank 0 applying a NaN-poisoned average."""
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker_overflow, args=(2, os.path.join(d, "init"), d), nprocs=2, join=True)
        o0, o1 = torch.load(os.path.join(d, "o0.pt")), torch.load(os.path.join(d, "o1.pt"))
    assert o0["flag"] == 1.0 and o1["flag"] == 1.0
    assert torch.equal(o0["p0"], o0["p1"]) and torch.equal(o1["p0"], o1["p1"]) and torch.equal(o0["p1"], o1["p1"])
    assert torch.isfinite(o0["p1"]).all()


def test_two_rank_model_path_on_one_gpu():
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, os.path.join(d, "init"), d), nprocs=2, join=True)
        r0, r1, ref = torch.load(os.path.join(d, "r0.pt")), torch.load(os.path.join(d, "r1.pt")), torch.load(os.path.join(d, "ref.pt"))
    assert torch.equal(r0["p0"], r1["p0"])                          # broadcast from rank 0
    assert torch.equal(r0["g"], r1["g"])                            # both ranks hold the same averaged gradient
    err = float((r0["g"] - ref["g"]).abs().max()); scale = float(ref["g"].abs().max())
    assert err <= 2e-4 * scale + 1e-7, (err, scale)                 # == gradient of the global batch (fp32 engine, summation order only)
    assert torch.equal(r0["p1"], r1["p1"]) and not torch.equal(r0["p1"], r0["p0"])
