#!/bin/bash
# tools/measure_qkv320_r02.sh [tag] -- on the MI355X box: M = 512 on qkv (N = 10240) with the 64 x 320 tiles
# (256 workgroups = one round on 256 CUs, where 128 x 128 needs 1.25), tiled and native kernels, hipBLASLt next to them.
TAG=${1:-r02e}
O=gpurun_out/${TAG}_sweeps
mkdir -p $O
for fam in "nv bf16" "mx bf16" "nv f16"; do
  set -- $fam
  timeout 600 python tools/tune.py --shapes qkv,sq8192 --ms 512 --kinds 8,12 --fmt $1 --dtype $2 --rotate-mb 640 --reps 5 --compare-dense --out $O/qkv320_$1_$2.json > $O/qkv320_$1_$2.log 2>&1
done
timeout 600 python tools/tune.py --shapes qkv --ms 512 --fmt mx --dtype bf16 --native --kinds 9,13 --no-check --rotate-mb 640 --reps 5 --out $O/qkv320_native.json > $O/qkv320_native.log 2>&1
for f in $O/qkv320*.log; do echo "== $f"; grep -v amdgpu.ids $f | grep "best\|DROPPED\|320\|hipblaslt\|dense" | cut -c1-220; done
