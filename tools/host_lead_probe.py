"""Host time per un-synchronised training step() call (profiles/r3_experiments.txt [C]): how far the host runs ahead of the GPU."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import probunet_amd as pa
cfg = dict(bench.CFG3)
dev = "cuda:0"
model = bench.build_model(cfg, "f16", dev)
model.sync_scalars = False
bench.perturb_zero_init(model)
opt = pa.FlatAdamW(model, lr=1e-4)
x, y = bench.synthetic_fields(cfg["batch"], cfg["input_channels"], cfg["num_classes"], cfg["H"], cfg["W"], 1234, dev)
ts = torch.zeros(cfg["batch"], 1, device=dev)
def step():
    loss = model.elbo(x, y, ts, M=cfg["M"])[0]
    opt.zero_grad(); loss.backward(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter(); marks = []
for i in range(12):
    a = time.perf_counter(); step(); marks.append((time.perf_counter() - a) * 1e3)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host ms per step() call:", " ".join(f"{m:.1f}" for m in marks))
print(f"host loop {1e3*(t1-t0):.1f} ms, drain after loop {1e3*(t2-t1):.1f} ms, total {1e3*(t2-t0)/12:.2f} ms/step")
