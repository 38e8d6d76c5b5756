#!/bin/bash
# tools/ablate_wide.sh <bits> [<bits> ...] -- builds ablation variants of the large-M 32x32 kernel (PETIT_ABLATE bits, gemm_wide.hpp: 1 no A-tile DMA, 2 no W
# refills, 4 no unpack VALU, 8 fragments read once, 16 no MFMA, 32 (group-ahead form) no step barrier) as separate libraries under tools/ablate/wide/ (the bf16 x NVFP4 part-4 TU only -- tiled, wide32
# and shared kernels; every other object is the shipped one).  Run on the GPU box: PETIT_AMD_LIB=<lib> python tools/power_probe.py --nv-only (results are garbage,
# only time / power count), or PETIT_AMD_LIB=<lib> python tools/tune.py --no-check --kinds 12 ...
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/tools/ablate/wide
cd $R/petit-kernel_amd
for abl in "$@"; do
  hipcc -O3 -std=c++20 -fPIC --offload-arch=gfx950 -Wno-unused-variable -Wno-unused-function -fno-gpu-rdc -DNDEBUG -mllvm -amdgpu-kernarg-preload-count=16 \
     -I../include -DPETIT_ABLATE=$abl -c csrc/gemm_nv_bf16_p4.hip -o $R/tools/ablate/wide/nv_bf16_p4_$abl.o &
done
wait
for abl in "$@"; do
  OBJS=$(ls build/*.o | grep -v gemm_nv_bf16_p4.o)
  hipcc -shared -fPIC --offload-arch=gfx950 -o $R/tools/ablate/wide/libpetit_abl_$abl.so $R/tools/ablate/wide/nv_bf16_p4_$abl.o $OBJS
done
ls -la $R/tools/ablate/wide/*.so
