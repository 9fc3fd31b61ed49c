"""Probe: does running the ELBO step as TWO half-batch pipelines on two HIP streams fill the per-kernel bubbles (prologue / epilogue /
kernel boundary) of the single-stream chain?  Two engine contexts (B = 16 each) on two torch streams vs one context at B = 32;
fwd + bwd only (no optimizer).  usage: python tools/dual_stream_probe.py [f16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dtype = sys.argv[1] if len(sys.argv) > 1 else "f16"
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)


def make(B):
    cfg = dict(bench.CFG3, batch=B)
    m = bench.build_model(cfg, dtype, dev)
    m.sync_scalars = False
    x, y = bench.synthetic_fields(B, 4, 1, 256, 256, 1234, dev)
    return m, x, y


def run(models, streams, iters):
    def one():
        for (m, x, y), s in zip(models, streams):
            with torch.cuda.stream(s):
                loss = m.elbo(x, y, None, M=5)[0]
                m.zero_grad(set_to_none=True)
                loss.backward()
    for _ in range(4):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        one()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


single = [make(32)]
t1 = run(single, [torch.cuda.current_stream()], 15)
print(f"one context, B=32, one stream : {1e3 * t1:7.2f} ms / 32 pairs  -> {32 / t1:7.1f} pairs/s (fwd+bwd, no optimizer)")
del single
torch.cuda.empty_cache()
duo = [make(16), make(16)]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
t2 = run(duo, [s1, s2], 15)
print(f"two contexts, B=16, two streams: {1e3 * t2:7.2f} ms / 32 pairs  -> {32 / t2:7.1f} pairs/s")
t3 = run(duo, [s1, s1], 15)
print(f"two contexts, B=16, ONE stream : {1e3 * t3:7.2f} ms / 32 pairs  -> {32 / t3:7.1f} pairs/s (control: half-batch kernels, no overlap)")
quad = duo + [make(8), make(8)]
