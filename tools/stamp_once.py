"""One launch of the conv3 kernel at a cfg3 layer shape through a -DPU_ABLATE=32 build (tools/ablate_conv.sh): per-block start / end
times, placement and cycle split.  usage: PU_LIB_PATH=.../libprobunet_ab32.so python tools/stamp_once.py 256 256 32"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from probunet_amd import _lib as L
lib = L.lib(); st = L.current_stream()
cin, cout, hw = (int(v) for v in sys.argv[1:4])
us = C.c_float(0)
rc = lib.pu_bench_conv(1, 0, 3, 32, cin, cout, hw, hw, 1, C.byref(us), st)      # 3 warm-up launches + 1 timed
torch.cuda.synchronize()
print("rc", rc, "timed launch", us.value, "us")
