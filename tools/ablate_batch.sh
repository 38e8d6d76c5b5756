#!/bin/bash
# tools/ablate_batch.sh <bits> [<bits> ...] -- ablation variants of the batched-decode kernel (PETIT_ABLATE_BATCH bits, gemm_batch.hpp: 1 no activation DMA after
# the prologue, 2 no unpack VALU, 4 no MFMA, 8 no LDS fragment reads, 16 no weight refills) as separate libraries under tools/ablate/batch/ (the bf16 x NVFP4 part-6
# TU only; every other object is the shipped one).  On the GPU box: PETIT_AMD_LIB=<lib> python tools/time_ids.py ...  (results are garbage, only time counts).
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/tools/ablate/batch
cd $R/petit-kernel_amd
for abl in "$@"; do
  hipcc -O3 -std=c++20 -fPIC --offload-arch=gfx950 -Wno-everything -fno-gpu-rdc -DNDEBUG -mllvm -amdgpu-kernarg-preload-count=16 \
     -I../include -DPETIT_ABLATE_BATCH=$abl -c csrc/gemm_nv_bf16_p6.hip -o $R/tools/ablate/batch/nv_bf16_p6_$abl.o &
done
wait
for abl in "$@"; do
  OBJS=$(ls build/*.o | grep -v gemm_nv_bf16_p6.o)
  hipcc -shared -fPIC --offload-arch=gfx950 -o $R/tools/ablate/batch/libpetit_abl_$abl.so $R/tools/ablate/batch/nv_bf16_p6_$abl.o $OBJS
done
ls -la $R/tools/ablate/batch/*.so
