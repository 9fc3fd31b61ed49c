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


def _worker(rank, world, initfile, outdir, buckets=4, wire=None):
    sys.path.insert(0, ROOT)
    import probunet_amd as pa
    from tests.filler import make_fields, make_eps
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"file://{initfile}", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)                                  # different initialisations: the broadcast must fix that
    m = pa.ProbabilisticUNet(2, 1, 4, [8, 16], 8, [1, 2], 0.7, 1.3, 0.0, dtype="f32").to(dev).train()
    m.dropout = 0.0
    m.dp_overlap_buckets = buckets                                 # > 0: bucketed all-reduce issued by elbo() behind the engine's events
    m.dp_wire_dtype = wire                                         # "bf16": gradients cross the wire as bfloat16
    m.enable_data_parallel()
    x, y = make_fields(4, 2, 1, 32, 32, seed=40); eps = make_eps(2, 4, 4)
    xs, ys = pa.dp.shard_batch(x, rank, world).contiguous().to(dev), pa.dp.shard_batch(y, rank, world).contiguous().to(dev)
    es = eps[:, rank * 2:(rank + 1) * 2].contiguous().to(dev)
    opt = pa.FlatAdamW(m, lr=1e-3)
    loss, _, _ = m.elbo(xs, ys, None, M=2, eps=es)
    ranges = m._dp_bucket_ranges()
    if buckets > 0:                                                # the buckets partition the flat buffer
        assert m._dp_works is not None and len(m._dp_works) == len(ranges) >= 2
        cover = sorted(ranges)
        assert cover[0][0] == 0 and cover[-1][1] == m._nparams and all(a[1] == b[0] for a, b in zip(cover, cover[1:]))
    else:
        assert m._dp_works is None and ranges == []
    opt.zero_grad(); loss.backward()
    assert m._dp_works is None
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
    torch.save(dict(p0=p0, p1=p1, flag=float(opt._flag), steps=opt.step_count), os.path.join(outdir, f"o{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_overflow_on_one_rank_makes_every_rank_skip():
    """f16 engine, rank 1 overflows (absurd loss scale), rank 0 does not: the SUM is non-finite on every rank, and the optimizer derives
    its skip flag from the averaged buffer it reads, so both ranks leave their (identical) parameters untouched instead of rank 0
    applying a NaN-poisoned average; the device step counter does not advance."""
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker_overflow, args=(2, os.path.join(d, "init"), d), nprocs=2, join=True)
        o0, o1 = torch.load(os.path.join(d, "o0.pt")), torch.load(os.path.join(d, "o1.pt"))
    assert o0["flag"] == 1.0 and o1["flag"] == 1.0 and o0["steps"] == 0 and o1["steps"] == 0
    assert torch.equal(o0["p0"], o0["p1"]) and torch.equal(o1["p0"], o1["p1"]) and torch.equal(o0["p1"], o1["p1"])
    assert torch.isfinite(o0["p1"]).all()


@pytest.mark.parametrize("buckets,wire", [(4, None), (0, None), (4, "bf16"), (0, "bf16")])
def test_two_rank_model_path_on_one_gpu(buckets, wire):
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, os.path.join(d, "init"), d, buckets, wire), nprocs=2, join=True)
        r0, r1, ref = torch.load(os.path.join(d, "r0.pt")), torch.load(os.path.join(d, "r1.pt")), torch.load(os.path.join(d, "ref.pt"))
    assert torch.equal(r0["p0"], r1["p0"])                          # broadcast from rank 0
    assert torch.equal(r0["g"], r1["g"])                            # both ranks hold the same averaged gradient
    err = float((r0["g"] - ref["g"]).abs().max()); scale = float(ref["g"].abs().max())
    tol = 2e-4 if wire is None else 1e-2                           # bf16 on the wire: 8 significant bits per rank contribution
    assert err <= tol * scale + 1e-7, (err, scale)                  # == gradient of the global batch (fp32 engine, summation order only)
    assert torch.equal(r0["p1"], r1["p1"]) and not torch.equal(r0["p1"], r0["p0"])


def test_bench_self_launch_two_rank_rehearsal():
    """`python bench.py --gpus 2` with NO launcher around it: the script spawns torch.distributed.run on itself before touching the GPU.
    PU_BENCH_REHEARSAL=1 puts both ranks on the one GPU of this box with gloo (RCCL refuses two ranks per device); a reduced batch keeps
    the two replicas small.  Checks the JSON contract of the multi-rank line."""
    import json, subprocess
    env = dict(os.environ, PU_BENCH_REHEARSAL="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2",
                        "--members", "2", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["config"]["world_size_seen_by_backend"] == 2 and out["config"]["backend"] == "gloo"
    assert out["config"]["dp_gradient_buckets"] >= 2 and out["config"]["loss_finite"] is True
    assert out["value"] > 0 and out["scaling"] == "weak" and out["config"]["global_batch"] == 4
    d = out["dp_diagnostics"]                                       # what makes the first real multi-GPU run diagnosable
    assert len(d["per_rank_ms_per_step"]) == 2 and all(v > 0 for v in d["per_rank_ms_per_step"])
    for k in ("ms_per_step_buckets_4", "ms_per_step_buckets_0", "ms_per_step_no_allreduce", "exposed_comm_ms_buckets_4", "exposed_comm_ms_buckets_0"):
        assert k in d, (k, d)


def test_bench_single_rank_rccl_path():
    """The data-parallel path against the REAL backend on a one-GPU box: a process group of one rank on 'nccl' (= RCCL), the flat
    broadcast, the bucketed all-reduce behind the engine's bucket events and the averaged delivery all run (PU_BENCH_FORCE_DIST=1).
    A group of one makes the collectives identities, so the loss must stay finite and the line must report backend nccl."""
    import json, socket, subprocess
    with socket.socket() as sk:                                    # a free rendezvous port
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    env = dict(os.environ, PU_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--batch", "4", "--members", "2",
                        "--no-cpu-baseline", "--no-secondary"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["config"]["backend"] == "nccl" and out["config"]["world_size_seen_by_backend"] == 1
    assert out["config"]["dp_gradient_buckets"] >= 2 and out["config"]["loss_finite"] is True and out["value"] > 0
    assert out["dp_diagnostics"]["ms_per_step_no_allreduce"] > 0 and len(out["dp_diagnostics"]["per_rank_ms_per_step"]) == 1


def _worker_wmse(rank, world, initfile, outdir):
    sys.path.insert(0, ROOT)
    import probunet_amd as pa
    from tests.filler import make_fields, make_eps
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"file://{initfile}", rank=rank, world_size=world)
    torch.manual_seed(3)
    m = pa.ProbabilisticUNet(2, 1, 4, [8, 16], 8, [1, 2], 1.0, 0.5, 0.0, dtype="f32", recon="wmse_msssim").to(dev).train()
    m.dropout = 0.0
    m.enable_data_parallel()
    x, y = make_fields(4, 2, 1, 128, 128, seed=43); eps = make_eps(1, 4, 4)
    y = y.clone(); y[2:] *= 3.0                                    # the two shards see different value ranges
    xs, ys = pa.dp.shard_batch(x, rank, world).contiguous().to(dev), pa.dp.shard_batch(y, rank, world).contiguous().to(dev)
    es = eps[:, rank * 2:(rank + 1) * 2].contiguous().to(dev)
    out = m.elbo(xs, ys, None, M=1, eps=es)
    out[0].backward()
    g = torch.cat([p.grad.flatten() for p in m.parameters()]).cpu()
    p0 = torch.cat([p.detach().flatten() for p in m.parameters()]).cpu()
    # the same step with the shard-local range (what rank-local inference would give): must differ, or the test proves nothing
    m.dp_global_data_range = False
    for p in m.parameters():
        p.grad = None
    m.elbo(xs, ys, None, M=1, eps=es)[0].backward()
    g_local = torch.cat([p.grad.flatten() for p in m.parameters()]).cpu()
    torch.save(dict(g=g, g_local=g_local), os.path.join(outdir, f"w{rank}.pt"))
    dist.barrier()
    if rank == 0:
        ref = pa.ProbabilisticUNet(2, 1, 4, [8, 16], 8, [1, 2], 1.0, 0.5, 0.0, dtype="f32", init=False, recon="wmse_msssim").to(dev).train()
        ref.dropout = 0.0
        with torch.no_grad():
            off = 0
            for p in ref.parameters():
                p.copy_(p0[off:off + p.numel()].view(p.shape).to(dev)); off += p.numel()
        ref.elbo(x.to(dev), y.to(dev), None, M=1, eps=eps.to(dev))[0].backward()
        torch.save(dict(g=torch.cat([p.grad.flatten() for p in ref.parameters()]).cpu()), os.path.join(outdir, "wref.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_wmse_msssim_uses_the_global_batch_range():
    """WMSE-MS-SSIM with data_range=None infers max(target) - min(target) over the batch (prob_unet_utils.py:288-289).  Under data
    parallelism the ranks exchange their minima / maxima, so the averaged gradient equals the single-process gradient of the global batch;
    with rank-local ranges (dp_global_data_range = False) it does not."""
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker_wmse, args=(2, os.path.join(d, "init"), d), nprocs=2, join=True)
        w0, w1, ref = torch.load(os.path.join(d, "w0.pt")), torch.load(os.path.join(d, "w1.pt")), torch.load(os.path.join(d, "wref.pt"))
    assert torch.equal(w0["g"], w1["g"])
    scale = float(ref["g"].abs().max())
    err = float((w0["g"] - ref["g"]).abs().max())
    err_local = float((w0["g_local"] - ref["g"]).abs().max())
    assert err <= 5e-4 * scale + 1e-7, (err, scale)
    assert err_local > 10 * max(err, 1e-7 * scale), (err_local, err, scale)
