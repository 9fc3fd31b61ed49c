#!/bin/bash
# Builds ablated copies of the library (kernels_conv.hip with -DPU_ABLATE=<bits>, see the comment there) next to the real one:
#   prob-unet-climate-downscaling_amd/libprobunet_ab<bits>.so      (git-ignored; results of ablated kernels are WRONG by design)
# and, with "run", times the conv3 kernel at the cfg3 layer shapes with each of them (tools/conv_microbench.py).
set -e
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/prob-unet-climate-downscaling_amd/csrc; HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
BITS="${ABLATE_BITS:-32 47 63}"
if [ "$1" != "run" ]; then
  make -C $C -j8 > /dev/null
  for b in $BITS; do
    $HIPCC --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DPU_ABLATE=$b -DPU_VARIANT=${VARIANT:-0} -c $C/kernels_conv.hip -o $C/build/kernels_conv_ab${b}v${VARIANT:-0}.o &
  done; wait
  for b in $BITS; do
    $HIPCC --offload-arch=gfx950 -shared -fPIC -o $R/prob-unet-climate-downscaling_amd/libprobunet_ab${b}v${VARIANT:-0}.so $C/build/engine.o $C/build/kernels_conv_ab${b}v${VARIANT:-0}.o \
      $C/build/kernels_wgrad.o $C/build/kernels_elem.o $C/build/kernels_fcomb.o $C/build/kernels_msssim.o
  done
  ls -la $R/prob-unet-climate-downscaling_amd/*.so
else
  echo "== baseline"; python3 $R/tools/conv_microbench.py f16 fwd
  for b in $BITS; do echo "== ablate $b"; PU_LIB_PATH=$R/prob-unet-climate-downscaling_amd/libprobunet_ab${b}v${VARIANT:-0}.so python3 $R/tools/conv_microbench.py f16 fwd; done
fi
