#!/bin/bash
# tools/ablate_wide.sh -- builds ablation variants of the large-M 32x32 kernel (PETIT_ABLATE bits, gemm_wide.hpp) as
# separate libraries under tools/ablate/wide/ (bf16 x NVFP4 TU only; the other objects are the shipped ones).
# Run on the GPU box: for each lib, PETIT_AMD_LIB=<lib> python tools/tune.py --no-check --kinds 12 ...
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/tools/ablate/wide
cd $R/petit-kernel_amd
for abl in "$@"; do
  hipcc -O3 -std=c++20 -fPIC --offload-arch=gfx950 -Wno-unused-variable -Wno-unused-function -fno-gpu-rdc -DNDEBUG -DPETIT_ABLATE=$abl \
     -c csrc/gemm_nv_bf16.hip -o $R/tools/ablate/wide/nv_bf16_$abl.o &
done
wait
for abl in "$@"; do
  hipcc -shared -fPIC --offload-arch=gfx950 -o $R/tools/ablate/wide/libpetit_abl_$abl.so $R/tools/ablate/wide/nv_bf16_$abl.o \
     build/api.o build/gemm_nv_f16.o build/gemm_mx_bf16.o build/gemm_mx_f16.o build/hal.o build/repack.o
done
ls -la $R/tools/ablate/wide/*.so
