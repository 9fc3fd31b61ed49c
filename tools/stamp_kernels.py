#!/usr/bin/env python3
"""In-kernel phase stamps (s_memtime) of the two hot MFMA kernels, taken in a DIAGNOSTIC build: the product sources carry no
instrumentation (round 2's PU_ABLATE / PU_VARIANT switches were removed), this script text-patches a copy of a kernel source,
compiles it with the product objects into scratch/<name>.so and runs one micro-benchmark launch through it (PU_LIB_PATH).

  python tools/stamp_kernels.py build            # here (hipcc cross-compiles): scratch/stamp_wgrad.so, scratch/stamp_conv3.so
  python tools/stamp_kernels.py run wgrad        # on the GPU box: cycles per tile of the phased weight-gradient loop, per tap group
  python tools/stamp_kernels.py run conv3        # per-wave phase split + per-CU block residency of conv3_kernel<f16,3,4,32,1,4,16>

Rules followed (cdna_hip_programming.md "In-kernel stamps"): a stamp is ONE asm statement `s_memtime ; s_waitcnt lgkmcnt(0)`; stamps sit
only at phase boundaries (never inside a section with counted lgkmcnt waits); every wave writes its sums to ITS OWN slot of a device array
- the first version added them with atomics to eight shared addresses and the contended atomics themselves took 47 us per launch and
slowed every load that shared their L2 channel (a 40 us kernel measured 440 us).  Stamped builds run ~8 % slower than the product.
The patches are anchored on source text: when a kernel changes, a failed `assert old in s` says which anchor to refresh.
What these stamps showed in round 3: profiles/r3_kernel_stamps.txt."""
import ctypes as C, collections, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "prob-unet-climate-downscaling_amd", "csrc")
SC = os.path.join(ROOT, "scratch")
HIPCC = "/opt/rocm/bin/hipcc"
STAMP_FN = ('__device__ __forceinline__ unsigned long long cstamp() { unsigned long long t; '
            'asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }\n')
READER = '''}  // namespace pu
extern "C" __attribute__((visibility("default"))) int pu_debug_stamps(unsigned long long* out) {
  static unsigned long long h[65536 * 8];
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(pu::g_stamps), sizeof h);
  for (size_t i = 0; i < sizeof h / sizeof h[0]; ++i) { out[i] = h[i]; h[i] = 0; }
  (void)hipMemcpyToSymbol(HIP_SYMBOL(pu::g_stamps), h, sizeof h);
  return 0;
}'''


def sub(s, old, new):
    assert old in s, "anchor not found:\n" + old[:200]
    return s.replace(old, new, 1)


def patch_wgrad(s):
    s = sub(s, '#include "pu_kernels.h"', '#include "%s/pu_kernels.h"' % CS)
    s = sub(s, "namespace pu {\n", "namespace pu {\n__device__ unsigned long long g_stamps[65536 * 8];\n" + STAMP_FN)
    s = sub(s, "    if (grp == 0) {\n      for (; tile < ntiles; tile += G) {\n        ktile_grp(cur);",
            "    unsigned long long ph[4] = {0, 0, 0, 0};\n    if (grp == 0) {\n      for (; tile < ntiles; tile += G) {\n"
            "        const unsigned long long t0 = cstamp();\n        ktile_grp(cur);")
    s = sub(s, "        ktile_grp(cur);\n        __syncthreads();\n        if (tile + G < ntiles) lstore(cur ^ 1);\n        if (tile + 2 * G < ntiles) gload(tile + 2 * G);\n        __syncthreads();\n        cur ^= 1;\n      }",
            "        ktile_grp(cur);\n        const unsigned long long t1 = cstamp();\n        __syncthreads();\n        const unsigned long long t2 = cstamp();\n"
            "        if (tile + G < ntiles) lstore(cur ^ 1);\n        if (tile + 2 * G < ntiles) gload(tile + 2 * G);\n        const unsigned long long t3 = cstamp();\n"
            "        __syncthreads();\n        const unsigned long long t4 = cstamp();\n        ph[0] += t1 - t0; ph[1] += t2 - t1; ph[2] += t3 - t2; ph[3] += t4 - t3;\n        cur ^= 1;\n      }")
    s = sub(s, "      for (; tile < ntiles; tile += G) {\n        if (tile + G < ntiles) lstore(cur ^ 1);\n        if (tile + 2 * G < ntiles) gload(tile + 2 * G);\n        __syncthreads();\n        ktile_grp(cur);\n        __syncthreads();\n        cur ^= 1;\n      }",
            "      for (; tile < ntiles; tile += G) {\n        const unsigned long long t0 = cstamp();\n        if (tile + G < ntiles) lstore(cur ^ 1);\n        if (tile + 2 * G < ntiles) gload(tile + 2 * G);\n"
            "        const unsigned long long t1 = cstamp();\n        __syncthreads();\n        const unsigned long long t2 = cstamp();\n        ktile_grp(cur);\n        const unsigned long long t3 = cstamp();\n"
            "        __syncthreads();\n        const unsigned long long t4 = cstamp();\n        ph[0] += t1 - t0; ph[1] += t2 - t1; ph[2] += t3 - t2; ph[3] += t4 - t3;\n        cur ^= 1;\n      }\n"
            "    }\n    if (l == 0 && KS == 3 && TH == 4 && TW == 32) {\n      const unsigned wid = (((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * NW + wave) & 65535u;\n"
            "      for (int i = 0; i < 4; ++i) g_stamps[wid * 8 + i] = ph[i];\n      g_stamps[wid * 8 + 4] = 1 + grp; g_stamps[wid * 8 + 5] = (ntiles - (int)blockIdx.x + G - 1) / G;\n    }\n    if (false) {")
    return sub(s, "}  // namespace pu", READER)


def patch_conv3(s):
    s = sub(s, '#include "pu_kernels.h"', '#include "%s/pu_kernels.h"' % CS)
    s = sub(s, "namespace pu {\n", "namespace pu {\n__device__ unsigned long long g_stamps[65536 * 8];\n" + STAMP_FN)
    s = sub(s, "  gload(0);\n  // the prologue issues the weight-fragment loads in EXACTLY the order",
            "  constexpr bool STAMP = MS == 16 && KS == 3 && TH == 4 && TW == 32 && WM == 1 && WN == 4 && !GNB;\n"
            "  unsigned long long ph[5] = {0, 0, 0, 0, 0}; unsigned long long tk0 = 0; if (STAMP) tk0 = cstamp();\n"
            "  gload(0);\n  // the prologue issues the weight-fragment loads in EXACTLY the order")
    s = sub(s, "  int cur = 0;\n  for (int c = 0; c < nch; ++c) {\n    const bool more = c + 1 < nch;\n    const int cn = more ? c + 1 : c;               // the last chunk harmlessly re-reads its own fragments\n    gload(cn * KC);",
            "  int cur = 0;\n  if (STAMP) ph[0] = cstamp() - tk0;\n  for (int c = 0; c < nch; ++c) {\n    const bool more = c + 1 < nch;\n    const int cn = more ? c + 1 : c;\n"
            "    unsigned long long ta = 0, tb = 0, tc = 0, td = 0; if (STAMP) ta = cstamp();\n    gload(cn * KC);\n    if (STAMP) tb = cstamp();")
    s = sub(s, "    if (more) lstore(cur ^ 1, (c + 1) * KC);\n    __syncthreads();\n    cur ^= 1;\n  }\n\n  // ---- epilogue: D[row = cout][col = pixel]",
            "    if (STAMP) tc = cstamp();\n    if (more) lstore(cur ^ 1, (c + 1) * KC);\n    if (STAMP) td = cstamp();\n    __syncthreads();\n"
            "    if (STAMP) { ph[1] += tb - ta; ph[2] += tc - tb; ph[3] += td - tc; ph[4] += cstamp() - td; }\n    cur ^= 1;\n  }\n\n  // ---- epilogue: D[row = cout][col = pixel]")
    s = sub(s, "                                                tiles_x * tiles_y, lpix);\n}\n\n// ------------------------------------------------------------------ conv3p",
            "                                                tiles_x * tiles_y, lpix);\n  if (STAMP) {\n    __builtin_amdgcn_s_waitcnt(0);\n    if (l == 0) {\n"
            "      const unsigned wid = ((blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) & 65535u;\n      for (int i = 0; i < 5; ++i) g_stamps[wid * 8 + i] = ph[i];\n"
            "      unsigned hw, xcc; asm volatile(\"s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\" : \"=s\"(hw)); asm volatile(\"s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)\" : \"=s\"(xcc));\n"
            "      g_stamps[wid * 8 + 5] = tk0; g_stamps[wid * 8 + 6] = cstamp(); g_stamps[wid * 8 + 7] = ((unsigned long long)xcc << 32) | hw;\n    }\n  }\n}\n\n"
            "// ------------------------------------------------------------------ conv3p")
    return sub(s, "}  // namespace pu", READER)


def build():
    os.makedirs(SC, exist_ok=True)
    subprocess.check_call(["make", "-C", CS, "-j4"], stdout=subprocess.DEVNULL)
    objs = {k: os.path.join(CS, "build", k + ".o") for k in ("engine", "kernels_conv", "kernels_wgrad", "kernels_elem", "kernels_fcomb", "kernels_msssim")}
    for name, src, fn in (("stamp_wgrad", "kernels_wgrad", patch_wgrad), ("stamp_conv3", "kernels_conv", patch_conv3)):
        hip = os.path.join(SC, name + ".hip"); obj = os.path.join(SC, name + ".o")
        open(hip, "w").write(fn(open(os.path.join(CS, src + ".hip")).read()))
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-fvisibility=hidden", "-I", CS, "-c", hip, "-o", obj])
        link = [o for k, o in objs.items() if k != src] + [obj]
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(SC, name + ".so")] + link)
        os.remove(obj)
        print("built", os.path.join(SC, name + ".so"))


def run(which):
    os.environ["PU_LIB_PATH"] = os.path.join(SC, "stamp_wgrad.so" if which == "wgrad" else "stamp_conv3.so")
    sys.path.insert(0, ROOT)
    import torch                                  # noqa: F401  (device context)
    from probunet_amd import _lib as L
    lib = L.lib(); st = L.current_stream()
    Buf = C.c_ulonglong * (65536 * 8)
    lib.pu_debug_stamps.argtypes = [C.POINTER(C.c_ulonglong)]
    shapes = [(128, 128, 64), (256, 256, 32)]
    for cin, cout, hw in shapes:
        us = C.c_float(0); buf = Buf()
        lib.pu_debug_stamps(buf)
        lib.pu_bench_conv(1, 2 if which == "wgrad" else 0, 3, 32, cin, cout, hw, hw, 1, C.byref(us), st)   # 3 warm-up launches + 1 timed: slots hold the last one
        lib.pu_debug_stamps(buf)
        rows = [tuple(buf[w * 8 + i] for i in range(8)) for w in range(65536)]
        if which == "wgrad":
            print(f"wgrad {cin}->{cout}@{hw}: {us.value:.1f} us (stamped build)")
            for g, nm in ((1, "group A: K-loop | barrier | staging | barrier"), (2, "group B: staging | barrier | K-loop | barrier")):
                rs = [r for r in rows if r[4] == g]
                if rs:
                    nt = sum(r[5] for r in rs)
                    print(f"   {nm}:  " + "  ".join(f"{sum(r[i] for r in rs) / nt:7.0f}" for i in range(4)) + f"   cycles per tile ({len(rs)} waves)")
        else:
            rs = [r for r in rows if r[6]]
            nch = cin // 32
            names = ["prologue", "gload/chunk", "mfma/chunk", "lstore/chunk", "barrier/chunk"]; div = [1, nch, nch, nch, nch]
            print(f"conv3 {cin}->{cout}@{hw}: {us.value:.1f} us (stamped build, one launch incl. launch latency), {len(rs)} waves, {nch} chunks")
            print("   per wave: " + "  ".join(f"{n} {sum(r[i] for r in rs) / len(rs) / d:.0f}" for i, (n, d) in enumerate(zip(names, div)))
                  + f"   wave lifetime {sum(r[6] - r[5] for r in rs) / len(rs):.0f} cycles")
            percu = collections.defaultdict(list)          # s_memtime bases differ between XCDs: spans are taken per CU
            for r in rs:
                hwid = r[7] & 0xffffffff
                percu[((r[7] >> 32) & 0xf, (hwid >> 13) & 7, (hwid >> 12) & 1, (hwid >> 8) & 0xf)].append((r[5], r[6]))
            spans = [max(b for _, b in v) - min(a for a, _ in v) for v in percu.values()]
            conc = [sum(b - a for a, b in v) / sp / 4 for v, sp in zip(percu.values(), spans)]
            print(f"   {len(percu)} CUs; busy span per CU {sum(spans) / len(spans):.0f} cycles (min {min(spans)}, max {max(spans)}); resident blocks per CU over its span: {sum(conc) / len(conc):.2f}")
            k = sorted(percu)[0]
            b0 = min(a for a, _ in percu[k])
            print("   waves of CU", k, "(start..end, cycles from the CU's first wave):", " ".join(f"[{a - b0}..{b - b0}]" for a, b in sorted(set(percu[k]))[:16]))


if __name__ == "__main__":
    if len(sys.argv) >= 2 and sys.argv[1] == "build":
        build()
    elif len(sys.argv) >= 3 and sys.argv[1] == "run":
        run(sys.argv[2])
    else:
        print(__doc__)
