"""Per-shape micro-benchmark of the MFMA convolution kernels at the cfg3 layer shapes (B=32). GPU only."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import probunet_amd as pa
from probunet_amd import _lib as L
lib = L.lib(); st = L.current_stream()
SHAPES = [(32, 32, 256), (64, 64, 128), (128, 128, 64), (256, 256, 32), (512, 512, 16), (96, 32, 256), (192, 64, 128), (1024, 512, 16), (8, 32, 256)]
dt = {"f16": 1, "bf16": 2, "f32": 0}[sys.argv[1] if len(sys.argv) > 1 else "f16"]
B = 32
only = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != "all" else None
if len(sys.argv) > 3:                       # e.g. 256,256,32 : one shape only
    SHAPES = [tuple(int(v) for v in sys.argv[3].split(","))]
for mode, nm in ((0, "fwd"), (1, "dgrad"), (2, "wgrad")):
    if only and nm != only:
        continue
    for cin, cout, hw in SHAPES:
        us = C.c_float(0)
        rc = lib.pu_bench_conv(dt, mode, 3, B, cin, cout, hw, hw, 20, C.byref(us), st)
        fl = 2.0 * B * hw * hw * cin * cout * 9
        by = B * hw * hw * (cin + cout) * 2
        print(f"{nm:5s} {cin:4d}->{cout:4d} @{hw:3d}: rc {rc} {us.value:8.1f} us  {fl / us.value / 1e6:7.1f} TF/s  {by / us.value / 1e3:7.1f} GB/s algo")
