"""Per-shape event timing of the GroupNorm kernels inside one serial cfg3 step (PU_PROF_GN=1 makes launch_gn_fwd / launch_gn_bwd
record HIP events per launch, tagged with kernel, resample mode, tensor shape, strides and flags).  Use it for the POPULATION of GroupNorm
launches (which shapes, how many, with which flags); the durations are inflated by the per-launch event creation on the host (14.2 ms
summed here against 9.7 ms in the rocprof trace) - time a shape with tools/gn_microbench.py.  usage: PU_PROF_GN=1 python tools/gn_profile.py"""
import os, sys
os.environ.setdefault("PU_PROF_GN", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from probunet_amd import _lib as L
import probunet_amd as pa

cfg = dict(bench.CFG3) if hasattr(bench, "CFG3") else None
dev = torch.device("cuda", 0)
m = bench.build_model(cfg, "f16", dev)
x, y = bench.synthetic_fields(cfg["batch"], cfg["input_channels"], cfg["num_classes"], cfg["H"], cfg["W"], 1, dev)
opt = pa.FlatAdamW(m, lr=1e-4)
lib = L.lib()
lib.pu_set_overlap(m._ctx, 0)
def step():
    loss, _, _ = m.elbo(x, y, None, M=cfg["M"])
    opt.zero_grad(); loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
lib.pu_profile_enable(1); step(); torch.cuda.synchronize(); lib.pu_profile_enable(0)
ents = (L.PuProfEntry * 512)()
n = lib.pu_profile_collect(ents, 512)
rows = [(ents[i].name.decode(), ents[i].launches, ents[i].ms, ents[i].bytes) for i in range(n) if ents[i].name.decode().startswith("gn_")]
rows.sort(key=lambda r: -r[2])
tot = sum(r[2] for r in rows)
print(f"GroupNorm kernels (without finalize): {tot:.3f} ms per step")
for name, k, ms, by in rows:
    print(f"{name:46s} x{k:2d} {1e3 * ms / k:7.1f} us/launch {ms:6.3f} ms  {by / (ms * 1e-3) / 1e12:5.2f} TB/s algorithmic")
