import sys; sys.path.insert(0, '/root/repo')
import torch, bench
from probunet_amd import _lib as L
dev = torch.device('cuda', 0)
for B, M, tag in ((32, 5, 'cfg3 B=32 M=5'), (64, 5, 'B=64 M=5'), (8, 64, 'cfg5 B=8 n=64')):
    cfg = dict(bench.CFG3, batch=B, M=M)
    m = bench.build_model(cfg, 'f16', dev); m.sync_scalars = False
    x, y = bench.synthetic_fields(B, 4, 1, 256, 256, 1, dev)
    with torch.no_grad():
        if M == 64: m.sample(x, 64)
        else: m.elbo(x, y, None, M=M)
    torch.cuda.synchronize()
    print(tag, 'arena GiB %.2f' % (L.lib().pu_workspace_bytes(m._ctx) / 2**30), 'torch GiB %.2f' % (torch.cuda.memory_allocated() / 2**30), 'free GiB %.1f' % (torch.cuda.mem_get_info()[0] / 2**30))
    del m; torch.cuda.empty_cache()
