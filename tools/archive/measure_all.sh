#!/bin/bash
# tools/measure_all.sh [round-tag] -- on the MI355X box: the sweeps the arch table and the numbers in
# DESIGN.md come from.  Outputs land in gpurun_out/<tag>_*; copy the ones to keep into profiles/.
TAG=${1:-r01}
mkdir -p gpurun_out
SH="sq8192,sq4096,qkv,gate_up,down"
for fam in "nv bf16" "nv f16" "mx bf16" "mx f16"; do
  set -- $fam
  python tools/tune.py --shapes $SH --ms 1,2,4,8,16 --fmt $1 --dtype $2 --out gpurun_out/${TAG}_tune_$1_$2.json > gpurun_out/${TAG}_tune_$1_$2.log 2>&1
  python tools/tune.py --shapes sq8192,qkv,gate_up,down --ms 32,64,128,256 --fmt $1 --dtype $2 --rotate-mb 640 --out gpurun_out/${TAG}_tune_midm_$1_$2.json > gpurun_out/${TAG}_tune_midm_$1_$2.log 2>&1
done
# very wide N at M = 16 / 32: the 16- and 32-row tiled shapes against the streaming kernel
for fam in "nv bf16" "nv f16" "mx bf16" "mx f16"; do
  set -- $fam
  python tools/tune.py --shapes gate_up --ms 16,32 --fmt $1 --dtype $2 --out gpurun_out/${TAG}_tune_wideN_$1_$2.json > gpurun_out/${TAG}_tune_wideN_$1_$2.log 2>&1
done
# the reference's headline claim (README.md:27, "1.2x-2.2x over hipBLASLt bf16 for batch < 16"): the default pick next to the
# vendor 16-bit GEMM on a dense weight of the same shape
python tools/tune.py --shapes $SH --ms 1,4,8,16 --fmt nv --dtype bf16 --only-default --compare-dense --out gpurun_out/${TAG}_tune_vs_dense_nv_bf16.json > gpurun_out/${TAG}_tune_vs_dense_nv_bf16.log 2>&1
# M = 512 (BASELINE config 5): dequant kernels and the vendor 16-bit GEMM on the same shapes
python tools/tune.py --shapes sq8192,qkv,gate_up,down --ms 512 --fmt nv --dtype bf16 --compare-dense --rotate-mb 640 --out gpurun_out/${TAG}_tune_bigm_nv_bf16.json > gpurun_out/${TAG}_tune_bigm_nv_bf16.log 2>&1
python tools/tune.py --shapes sq8192,qkv,gate_up,down --ms 512 --fmt nv --dtype f16 --rotate-mb 640 --out gpurun_out/${TAG}_tune_bigm_nv_f16.json > gpurun_out/${TAG}_tune_bigm_nv_f16.log 2>&1
python tools/tune.py --shapes sq8192,qkv,gate_up,down --ms 128,256,512 --fmt mx --dtype f16 --rotate-mb 640 --out gpurun_out/${TAG}_tune_bigm_mx_f16.json > gpurun_out/${TAG}_tune_bigm_mx_f16.log 2>&1
python tools/tune.py --shapes sq8192,qkv,gate_up,down --ms 512 --fmt mx --dtype bf16 --compare-dense --rotate-mb 640 --out gpurun_out/${TAG}_tune_bigm_mx_bf16.json > gpurun_out/${TAG}_tune_bigm_mx_bf16.log 2>&1
# the opt-in native-FP4 kernels (MXFP4 weights x MXFP8-quantised activations) next to them
python tools/tune.py --shapes sq8192,qkv,gate_up,down --ms 512,2048 --fmt mx --dtype bf16 --native --compare-dense --rotate-mb 640 --out gpurun_out/${TAG}_tune_native_mx_bf16.json > gpurun_out/${TAG}_tune_native_mx_bf16.log 2>&1
for f in gpurun_out/${TAG}_tune_*.log; do echo "== $f"; grep -v amdgpu.ids $f | grep best | cut -c1-200; done
