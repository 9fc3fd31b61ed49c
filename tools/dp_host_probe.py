"""Host-side time of the three calls of a training step (elbo, backward, optimizer) with the data-parallel path off and on (a process
group of ONE rank on RCCL).  No device synchronisation inside the timed calls: a call that takes as long as the GPU step is blocking the
host.  usage: python tools/dp_host_probe.py [0|1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import probunet_amd as pa
force = len(sys.argv) > 1 and sys.argv[1] == "1"
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
if force:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
cfg = dict(bench.CFG3)
m = bench.build_model(cfg, "f16", dev)
m.sync_scalars = False
if force:
    m.enable_data_parallel(single_rank_ok=True)
opt = pa.FlatAdamW(m, lr=1e-4)
x, y = bench.synthetic_fields(cfg["batch"], cfg["input_channels"], cfg["num_classes"], cfg["H"], cfg["W"], 1, dev)
t = torch.zeros(cfg["batch"], 1, device=dev)
acc = [0.0, 0.0, 0.0, 0.0]
evs = []
def step(rec):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    e[0].record()
    t0 = time.perf_counter(); loss = m.elbo(x, y, t, M=cfg["M"])[0]
    e[1].record()
    t1 = time.perf_counter(); opt.zero_grad(); loss.backward()
    e[2].record()
    t2 = time.perf_counter(); opt.step()
    e[3].record()
    t3 = time.perf_counter()
    if rec:
        acc[0] += t1 - t0; acc[1] += t2 - t1; acc[2] += t3 - t2; evs.append(e)
for _ in range(5): step(False)
torch.cuda.synchronize()
N = 20
w0 = time.perf_counter()
for _ in range(N): step(True)
h1 = time.perf_counter()
torch.cuda.synchronize()
w1 = time.perf_counter()
print(f"force_dist={int(force)}: host ms per step: elbo {1e3*acc[0]/N:.2f}  backward {1e3*acc[1]/N:.2f}  optimizer {1e3*acc[2]/N:.2f}  "
      f"| host loop {1e3*(h1-w0)/N:.2f}  wall {1e3*(w1-w0)/N:.2f}")
g = [sum(e[i].elapsed_time(e[i + 1]) for e in evs) / len(evs) for i in range(3)]
gap = sum(a[3].elapsed_time(b[0]) for a, b in zip(evs, evs[1:])) / (len(evs) - 1)
print(f"   GPU ms per step on the main stream: elbo {g[0]:.2f}  backward {g[1]:.2f}  optimizer {g[2]:.2f}  between steps {gap:.3f}")
if force:
    dist.destroy_process_group()
