"""GroupNorm+SiLU kernel micro-benchmark at the cfg3 tensor shapes (pu_bench_gn: random NHWC f16 data, back-to-back launches of ONE
kernel between two HIP events).  Prints us per launch and algorithmic TB/s.  usage: python tools/gn_microbench.py [shape filter]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from probunet_amd import _lib as L
lib = L.lib(); st = L.current_stream()
B = 32
# (H, C, drop_p, flags) - the cfg3 GroupNorm sites that dominate tools/gn_profile.py
shapes = [(256, 32, 0.0, 0), (256, 32, 0.1, 0), (256, 32, 0.1, 2), (256, 64, 0.0, 0), (256, 96, 0.0, 0), (128, 64, 0.1, 0), (128, 64, 0.0, 0),
          (128, 128, 0.0, 0), (128, 192, 0.0, 0), (64, 128, 0.1, 0), (64, 256, 0.0, 0), (64, 384, 0.0, 0), (32, 256, 0.1, 0), (32, 512, 0.0, 0),
          (16, 512, 0.1, 0), (16, 1024, 0.0, 0)]
# resampled sites (rs 1 = avg-pool down fused into apply, rs 2 = nearest up): (rs, H of x, C)
rs_shapes = [(1, 256, 32), (1, 128, 64), (1, 64, 128), (1, 32, 256), (2, 128, 64), (2, 64, 128), (2, 32, 256), (2, 16, 512)]
flt = sys.argv[1] if len(sys.argv) > 1 else ""
out = (C.c_float * 4)()
print(f"{'shape':28s} {'apply':>16s} {'pass1':>16s} {'pass2':>16s} {'small':>16s}   (us, TB/s)")
for H, Cc, dp, fl in shapes:
    name = f"{B}x{H}x{H}x{Cc} p={dp} f={fl}"
    if flt and flt not in name:
        continue
    iters = 20 if H >= 128 else 50
    rc = lib.pu_bench_gn(1, 0, B, Cc, H, H, dp, fl, iters, out, st)
    torch.cuda.synchronize()
    t = B * H * H * Cc * 2
    by = [2 * t, 2 * t, (3 + (fl & 1) + ((fl >> 1) & 1)) * t, (3 + (fl & 1) + ((fl >> 1) & 1)) * t]
    cells = [f"{out[k]:8.1f} {by[k] / (out[k] * 1e-6) / 1e12:5.2f}" if out[k] > 0 else f"{'-':>14s}" for k in range(4)]
    print(f"{name:28s} " + "   ".join(cells) + (f"  rc={rc}" if rc else ""), flush=True)
for rs, H, Cc in rs_shapes:
    name = f"rs{rs} {B}x{H}x{H}x{Cc}"
    if flt and flt not in name:
        continue
    rc = lib.pu_bench_gn(1, rs, B, Cc, H, H, 0.0, 0, 20 if H >= 128 else 50, out, st)
    torch.cuda.synchronize()
    t = B * H * H * Cc * 2
    ty = t // 4 if rs == 1 else t * 4
    by = [t + ty, t + ty, 2 * t + ty, 0]
    cells = [f"{out[k]:8.1f} {by[k] / (out[k] * 1e-6) / 1e12:5.2f}" if out[k] > 0 else f"{'-':>14s}" for k in range(4)]
    print(f"{name:28s} " + "   ".join(cells) + (f"  rc={rc}" if rc else ""), flush=True)
