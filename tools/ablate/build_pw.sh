#!/bin/bash
# tools/ablate/build_pw.sh -- experiment builds of the native 32x32x64 kernel's loader-wave variants (bf16 x MXFP4 TU only, the other objects are the
# shipped ones; run `python petit-kernel_amd/build.py` first):
#   <ahead>_<share>  PETIT_N32_PW=<ahead> PETIT_N32_PWSHARE=<share>: a PREFETCH wave touches the workgroup's weight lines <ahead> k-tiles early, every
#                    <share>-th line per row block of the panel (share = 4: the four row blocks of a panel split the work)
#   early            PETIT_N32_EARLY_REFILL=1: a tile's weight refills go out right after its MFMAs instead of after the stage barrier
#   pf<N>            PETIT_N32_LWPF=<N>: the loader wave keeps N activation stages in flight (up to what vmcnt can count)
# then on the GPU box: bash tools/ablate/run_pw.sh base 4_4 8_1 8_4 16_4 pf4 pf7
R=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p $R/tools/ablate/pw
cd $R/petit-kernel_amd
build() { # name, defines
  hipcc -O3 -std=c++20 -fPIC --offload-arch=gfx950 -Wno-unused-variable -Wno-unused-function -fno-gpu-rdc -DNDEBUG $2 \
     -mllvm -amdgpu-kernarg-preload-count=16 -I../include -c csrc/gemm_mx_bf16.hip -o $R/tools/ablate/pw/mx_bf16_$1.o &&
  hipcc -shared -fPIC --offload-arch=gfx950 -o $R/tools/ablate/pw/libpetit_pw_$1.so $R/tools/ablate/pw/mx_bf16_$1.o \
     build/api.o build/tune.o build/gemm_nv_f16.o build/gemm_nv_bf16.o build/gemm_mx_f16.o build/hal.o build/repack.o build/dequant.o
}
for v in "${@:-4_4 8_1 8_4 16_4 pf4 pf7}"; do
  case $v in
    pf*) build $v "-DPETIT_N32_LWPF=${v#pf}" & ;;
    early*) build $v "-DPETIT_N32_EARLY_REFILL=${v#early}" & ;;
    *)   build $v "-DPETIT_N32_PW=${v%_*} -DPETIT_N32_PWSHARE=${v#*_}" & ;;
  esac
done
wait
ls -la $R/tools/ablate/pw/*.so
