#!/bin/bash
# tools/ablate_native32.sh <bits> ... -- builds ablation variants of the native 32x32x64 kernel (PETIT_ABLATE_N32 bits,
# gemm_native32.hpp) as separate libraries under tools/ablate/n32/ (bf16 x MXFP4 TU only; the other objects are the shipped ones).
# On the GPU box: for each lib, PETIT_AMD_LIB=<lib> python tools/tune.py --native --no-check --kinds 13 ...
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/tools/ablate/n32
cd $R/petit-kernel_amd
for abl in "$@"; do
  hipcc -O3 -std=c++20 -fPIC --offload-arch=gfx950 -Wno-unused-variable -Wno-unused-function -fno-gpu-rdc -DNDEBUG -DPETIT_ABLATE_N32=$abl \
     -mllvm -amdgpu-kernarg-preload-count=16 -I../include -c csrc/gemm_mx_bf16.hip -o $R/tools/ablate/n32/mx_bf16_$abl.o &
done
wait
for abl in "$@"; do
  hipcc -shared -fPIC --offload-arch=gfx950 -o $R/tools/ablate/n32/libpetit_abl_$abl.so $R/tools/ablate/n32/mx_bf16_$abl.o \
     build/api.o build/tune.o build/gemm_nv_f16.o build/gemm_nv_bf16.o build/gemm_mx_f16.o build/hal.o build/repack.o build/dequant.o
done
ls -la $R/tools/ablate/n32/*.so
