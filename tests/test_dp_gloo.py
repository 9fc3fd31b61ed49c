"""CPU, world_size 2, gloo: the data-parallel exchange used by the ELBO path (one all-reduce of the flat gradient
buffer; rank-offset seeds; contiguous batch shards) and the property it rests on: the mean of equal-sized shard
gradients equals the global-batch gradient (checked with the oracle on a tiny model)."""
import os, sys, tempfile
import pytest, torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, initfile, out):
    sys.path.insert(0, ROOT)
    import probunet_amd as pa
    from oracle import probunet_oracle as O
    from tests.helpers import filled_params
    from tests.filler import make_fields, make_eps
    dist.init_process_group("gloo", init_method=f"file://{initfile}", rank=rank, world_size=world)
    torch.set_num_threads(2)
    cfg = O.Config(2, 2, 6, [8, 16, 32], 8, [1, 2, 4])
    P = filled_params(cfg)
    x, y = make_fields(4, 2, 2, 32, 32); eps = make_eps(2, 4, 6)
    xs, ys = pa.dp.shard_batch(x, rank, world), pa.dp.shard_batch(y, rank, world)
    es = eps[:, rank * 2:(rank + 1) * 2]
    _, g = O.elbo_with_grads(P, cfg, xs, ys, es, beta0=0.7, beta1=1.3)
    names = sorted(g)
    flat = torch.cat([g[k].reshape(-1) for k in names])
    pa.dp.allreduce_mean_(flat, None, bucket_elems=10_000)          # bucketed path
    flat2 = torch.cat([g[k].reshape(-1) for k in names])
    pa.dp.allreduce_mean_(flat2)                                    # single collective
    assert torch.allclose(flat, flat2, rtol=0, atol=0)
    flat3 = torch.cat([g[k].reshape(-1) for k in names])
    pa.dp.allreduce_mean_(flat3, None, bucket_elems=10_000, wire_dtype=torch.bfloat16)     # optional bf16 wire compression
    assert flat3.dtype == torch.float32
    assert float((flat3 - flat).abs().max()) <= 1e-2 * float(flat.abs().max()) and not torch.equal(flat3, flat)
    assert pa.dp.rank_seed(7, rank) != pa.dp.rank_seed(7, 1 - rank)
    if rank == 0:
        _, gg = O.elbo_with_grads(P, cfg, x, y, eps, beta0=0.7, beta1=1.3)
        ref = torch.cat([gg[k].reshape(-1) for k in names])
        torch.save(dict(err=float((flat - ref).abs().max()), scale=float(ref.abs().max())), out)
    dist.destroy_process_group()


def test_shard_gradient_mean_equals_global_batch_gradient():
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "r.pt")
        mp.spawn(_worker, args=(2, os.path.join(d, "init"), out), nprocs=2, join=True)
        r = torch.load(out)
        assert r["err"] <= 1e-5 * max(1.0, r["scale"]), r
