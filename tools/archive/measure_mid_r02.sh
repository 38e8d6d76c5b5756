#!/bin/bash
# tools/measure_mid_r02.sh [tag] -- on the MI355X box: M = 8 / 16 with the shared-activation-tile kernels (gemm_mid.hpp, ids with
# warp_partition_m = 2) next to the staged streaming kernels (kinds 10 / 11) and the tiled ones (8).
TAG=${1:-r02}
O=gpurun_out/${TAG}_sweeps
mkdir -p $O
SH="sq8192,sq4096,qkv,gate_up,down"
for fam in "nv bf16" "nv f16" "mx bf16"; do
  set -- $fam
  timeout 900 python tools/tune.py --shapes $SH --ms 8,16 --fmt $1 --dtype $2 --kinds 0,8,10,11 --reps 5 --out $O/mid_$1_$2.json > $O/mid_$1_$2.log 2>&1
done
for f in $O/mid_*.log; do echo "== $f"; grep -v amdgpu.ids $f | grep "best\|DROPPED" | cut -c1-230; done
