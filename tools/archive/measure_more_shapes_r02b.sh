#!/bin/bash
# tools/measure_more_shapes_r02b.sh -- Llama-3-8B linears and the TP = 8 shards of Llama-3-70B re-swept with the kernels added late
# in round 2: decode (M <= 4, NVFP4), shared-tile (M = 8 / 16), 64x320 / 128x256 tiled shapes (M >= 128).
O=gpurun_out/r02_more2
mkdir -p $O
SH="6144x4096,4096x4096,28672x4096,4096x14336,1280x8192,8192x1024,7168x8192,8192x3584"
for fam in "nv bf16" "nv f16" "mx bf16" "mx f16"; do
  set -- $fam
  timeout 900 python tools/tune.py --shapes $SH --ms 1,2,4,8,16 --fmt $1 --dtype $2 --reps 3 --out $O/dec_$1_$2.json > $O/dec_$1_$2.log 2>&1
  timeout 900 python tools/tune.py --shapes $SH --ms 128,256,512 --kinds 8,12 --splitk 1,2,4,8 --splitk-kinds tiled --fmt $1 --dtype $2 --rotate-mb 640 --reps 3 --out $O/big_$1_$2.json > $O/big_$1_$2.log 2>&1
done
grep -c best $O/*.log; grep -h DROPPED $O/*.log | cut -c 1-200 | head
