#!/bin/bash
# tools/measure_decode_r02.sh [tag] -- on the MI355X box: the M <= 4 sweep with the scale-after-MFMA decode kernels
# (gemm_decode.hpp, kinds 4 / 14 / 15) next to the streaming kernels (kinds 0-3, 5-7); NVFP4 only.
TAG=${1:-r02}
O=gpurun_out/${TAG}_sweeps
mkdir -p $O
SH="sq8192,sq4096,qkv,gate_up,down"
for dt in bf16 f16; do
  timeout 900 python tools/tune.py --shapes $SH --ms 1,2,3,4 --fmt nv --dtype $dt --kinds 0,1,2,3,4,5,6,7,14,15 --reps 5 --out $O/decode2_nv_$dt.json > $O/decode2_nv_$dt.log 2>&1
done
# Llama-3-8B linears and the TP = 8 shards of Llama-3-70B
SH2="6144x4096,4096x4096,28672x4096,4096x14336,1280x8192,8192x1024,7168x8192,8192x3584"
for dt in bf16 f16; do
  timeout 900 python tools/tune.py --shapes $SH2 --ms 1,2,4 --fmt nv --dtype $dt --kinds 0,1,2,3,4,5,6,7,14,15 --reps 3 --out $O/decode2_more_nv_$dt.json > $O/decode2_more_nv_$dt.log 2>&1
done
for f in $O/decode2_*.log; do echo "== $f"; grep -v amdgpu.ids $f | grep "best\|DROPPED" | cut -c1-220; done
