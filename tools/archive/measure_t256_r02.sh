#!/bin/bash
# tools/measure_t256_r02.sh [tag] -- on the MI355X box: the 128 x 256 tiled shape (half the LDS fragment traffic per flop of
# 128 x 128, the same unpack per flop) with K splits, M = 128 .. 512, all four families, next to the existing tiled shapes.
TAG=${1:-r02l}
O=gpurun_out/${TAG}_sweeps
mkdir -p $O
for fam in "nv bf16" "nv f16" "mx bf16" "mx f16"; do
  set -- $fam
  timeout 900 python tools/tune.py --shapes sq8192,qkv,gate_up,down --ms 128,256,512 --kinds 8,12 --splitk 1,2,4,8 --splitk-kinds tiled --fmt $1 --dtype $2 --rotate-mb 640 --reps 3 --out $O/t256_$1_$2.json > $O/t256_$1_$2.log 2>&1
done
for f in $O/t256_*.log; do echo "== $f"; grep -v amdgpu.ids $f | grep "best\|DROPPED" | cut -c1-200; done
