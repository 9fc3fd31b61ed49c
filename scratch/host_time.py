import sys, time, torch
sys.path.insert(0, '/root/repo')
import bench
dev = torch.device('cuda', 0)
cfg = dict(bench.CFG3)
m = bench.build_model(cfg, 'f16', dev); m.sync_scalars = False
x, y = bench.synthetic_fields(cfg['batch'], 4, 1, 256, 256, 1, dev)
for _ in range(3):
    l = m.elbo(x, y, None, M=5)[0]; l.backward()
torch.cuda.synchronize()
hs, ts = [], []
for _ in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    l = m.elbo(x, y, None, M=5)[0]
    t1 = time.perf_counter()
    l.backward()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    hs.append((t1 - t0) * 1e3); ts.append((t2 - t0) * 1e3)
print('host return ms', ['%.1f' % v for v in hs]); print('synced ms', ['%.1f' % v for v in ts])
