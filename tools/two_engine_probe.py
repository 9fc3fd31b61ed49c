"""Two engines of 16 field pairs on two streams against one engine of 32 (profiles/r3_experiments.txt [B]): does running half-batches side by
side - the HBM-bound GroupNorm kernels of one under the MFMA-bound convolutions of the other - beat one full batch?  It does not (-21 %)."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import probunet_amd as pa
dev = "cuda:0"
def make(Bn):
    cfg = dict(bench.CFG3, batch=Bn)
    m = bench.build_model(cfg, "f16", dev); m.sync_scalars = False
    opt = pa.FlatAdamW(m, lr=1e-4)
    x, y = bench.synthetic_fields(Bn, cfg["input_channels"], cfg["num_classes"], cfg["H"], cfg["W"], 1234, dev)
    ts = torch.zeros(Bn, 1, device=dev)
    def step():
        loss = m.elbo(x, y, ts, M=cfg["M"])[0]
        opt.zero_grad(); loss.backward(); opt.step()
    return step
def run(steps_fns, streams, n=12, warm=4):
    for _ in range(warm):
        for f, s in zip(steps_fns, streams):
            with torch.cuda.stream(s): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        for f, s in zip(steps_fns, streams):
            with torch.cuda.stream(s): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
one = make(32)
t = run([one], [torch.cuda.current_stream()])
print(f"one engine B=32: {t*1e3:.2f} ms/step  {32/t:.1f} pairs/s")
del one; torch.cuda.empty_cache()
a, b = make(16), make(16)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
t = run([a, b], [s1, s2])
print(f"two engines B=16 on two streams: {t*1e3:.2f} ms per pair of steps  {32/t:.1f} pairs/s")
t = run([a, b], [s1, s1])
print(f"two engines B=16 on ONE stream: {t*1e3:.2f} ms per pair of steps  {32/t:.1f} pairs/s")
