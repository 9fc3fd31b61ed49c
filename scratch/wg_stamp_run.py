"""scratch: per-phase cycle shares of the instrumented weight-gradient kernel (scratch/wg_stamp.so)"""
import ctypes as C, sys, os
os.environ["PU_LIB_PATH"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "wg_stamp.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import probunet_amd as pa
from probunet_amd import _lib as L
lib = L.lib(); st = L.current_stream()
lib.pu_debug_stamps.argtypes = [C.POINTER(C.c_ulonglong)]
names = ["gload issue", "K loop (reads + MFMA)", "bias sums", "wait vmcnt(0)", "lstore", "barrier"]
for cin, cout, hw in [(128, 128, 64), (256, 256, 32), (64, 64, 128), (32, 32, 256)]:
    us = C.c_float(0); buf = (C.c_ulonglong * 8)()
    lib.pu_debug_stamps(buf)
    rc = lib.pu_bench_conv(1, 2, 3, 32, cin, cout, hw, hw, 10, C.byref(us), st)
    lib.pu_debug_stamps(buf)
    v = list(buf); nw = max(v[7], 1)
    tot = sum(v[:6])
    print(f"wgrad {cin}->{cout}@{hw}: {us.value:.1f} us, waves {nw}, tile-loop cycles/wave {v[6]/nw:.0f}")
    for n, x in zip(names, v[:6]):
        print(f"    {n:24s} {x/nw:10.0f} cyc/wave  {100*x/tot:5.1f} %")
