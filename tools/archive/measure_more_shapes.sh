#!/bin/bash
# tools/measure_more_shapes.sh [tag] -- arch-table rows beyond the BASELINE shapes: Llama-3-8B linears and the TP = 8 shards of
# Llama-3-70B (tools/benchmarks/matmul.py:18-33,96-100 of the reference list the same families).
TAG=${1:-r01}
mkdir -p gpurun_out
SH="6144x4096,4096x4096,28672x4096,4096x14336,1280x8192,8192x1024,7168x8192,8192x3584"
for fam in "nv bf16" "nv f16" "mx bf16" "mx f16"; do
  set -- $fam
  python tools/tune.py --shapes $SH --ms 1,2,4,8,16,32,64,128,256,512 --fmt $1 --dtype $2 --rotate-mb 640 --reps 3 --out gpurun_out/${TAG}_tune_more_$1_$2.json > gpurun_out/${TAG}_tune_more_$1_$2.log 2>&1
done
for f in gpurun_out/${TAG}_tune_more_*.log; do echo "== $f"; grep -c best $f; done
