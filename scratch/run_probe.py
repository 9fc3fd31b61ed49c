import ctypes, torch, os, sys
lib = ctypes.CDLL(os.path.join(os.path.dirname(__file__), "libprobe.so"))
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
x = torch.randn(1000, device=dev); y = torch.zeros(1000, device=dev)
rc = lib.probe_axpy(ctypes.c_void_p(y.data_ptr()), ctypes.c_void_p(x.data_ptr()), ctypes.c_float(2.0), 1000, ctypes.c_void_p(st))
torch.cuda.synchronize(); print("axpy rc", rc, "err", (y - 2 * x).abs().max().item())
def mf(which, dt, M, K):
    A = torch.randint(-3, 4, (M, K), device=dev).to(dt); B = torch.randint(-3, 4, (K, M), device=dev).to(dt)
    Bt = B.t().contiguous(); D = torch.zeros(M, M, device=dev)
    rc = lib.probe_mfma(which, ctypes.c_void_p(A.data_ptr()), ctypes.c_void_p(Bt.data_ptr()), ctypes.c_void_p(D.data_ptr()), ctypes.c_void_p(st))
    torch.cuda.synchronize(); ref = A.float() @ B.float()
    print("mfma", which, dt, "rc", rc, "maxerr", (D - ref).abs().max().item())
mf(0, torch.float16, 32, 16); mf(1, torch.bfloat16, 32, 16); mf(2, torch.float32, 32, 2); mf(3, torch.float16, 16, 32)
out = torch.zeros(128, dtype=torch.int32, device=dev)
rc = lib.probe_trread(ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(st)); torch.cuda.synchronize()
o = out.cpu().tolist()
for l in range(64):
    lo, hi = o[2*l] & 0xffffffff, o[2*l+1] & 0xffffffff
    vals = [lo & 0xffff, lo >> 16, hi & 0xffff, hi >> 16]
    print("lane", l, [(v >> 8, v & 255) for v in vals])
print(torch.cuda.get_device_name(0), torch.version.hip)
