#!/bin/bash
# Ablated copies of the library for the weight-gradient main kernel (kernels_wgrad.hip with -DPU_WG_ABLATE=<bits>, see the comment there):
#   prob-unet-climate-downscaling_amd/libprobunet_wgab<bits>.so   (git-ignored; results of ablated kernels are WRONG by design)
# "run": times the kernel at the cfg3 layer shapes with each of them (tools/conv_microbench.py ... wgrad).
set -e
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/prob-unet-climate-downscaling_amd/csrc; HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
BITS="${ABLATE_BITS:-1 2 4 8 16 3 7}"
if [ "$1" != "run" ]; then
  make -C $C -j8 > /dev/null
  for b in $BITS; do
    $HIPCC --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DPU_WG_ABLATE=$b -c $C/kernels_wgrad.hip -o $C/build/kernels_wgrad_ab${b}.o &
  done; wait
  for b in $BITS; do
    $HIPCC --offload-arch=gfx950 -shared -fPIC -o $R/prob-unet-climate-downscaling_amd/libprobunet_wgab${b}.so $C/build/engine.o $C/build/kernels_conv.o \
      $C/build/kernels_wgrad_ab${b}.o $C/build/kernels_elem.o $C/build/kernels_fcomb.o $C/build/kernels_msssim.o
  done
  ls -la $R/prob-unet-climate-downscaling_amd/*.so
else
  echo "== baseline"; python3 $R/tools/conv_microbench.py f16 wgrad
  for b in $BITS; do echo "== ablate $b"; PU_LIB_PATH=$R/prob-unet-climate-downscaling_amd/libprobunet_wgab${b}.so python3 $R/tools/conv_microbench.py f16 wgrad; done
fi
