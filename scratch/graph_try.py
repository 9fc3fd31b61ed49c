import sys, time, torch
sys.path.insert(0, '.')
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
cfg = dict(bench.CFG3)
model = bench.build_model(cfg, "f16", dev); model.sync_scalars = False
opt = torch.optim.AdamW(model.parameters(), lr=1e-4, capturable=True, fused=True)
x, y = bench.synthetic_fields(cfg["batch"], 4, 1, 256, 256, 1234, dev)
def step():
    loss, rl, kl = model.elbo(x, y, None, M=5)
    opt.zero_grad(); loss.backward(); opt.step()
    return loss
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=True)
with torch.cuda.graph(g):
    loss = step()
torch.cuda.synchronize()
for _ in range(3): g.replay()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): g.replay()
torch.cuda.synchronize(); el = time.perf_counter() - t0
print("graph replay ms/step", 1e3 * el / 10, "loss", float(loss))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize(); el = time.perf_counter() - t0
print("eager ms/step", 1e3 * el / 10)
