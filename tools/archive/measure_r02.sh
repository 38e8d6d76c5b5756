#!/bin/bash
# tools/measure_r02.sh [tag] -- on the MI355X box: the round-2 sweeps behind tuned_gfx950.inc and DESIGN.md.
# Every candidate's output is checked before it is timed (tools/tune.py); results: compact CSV + JSON summary + tune rows.
TAG=${1:-r02}
O=gpurun_out/${TAG}_sweeps
mkdir -p $O
SH="sq8192,sq4096,qkv,gate_up,down"
for fam in "nv bf16" "nv f16" "mx bf16" "mx f16"; do
  set -- $fam
  # decode regime: every streaming kernel
  timeout 900 python tools/tune.py --shapes $SH --ms 1,2,4,8,16 --fmt $1 --dtype $2 --reps 5 --out $O/decode_$1_$2.json > $O/decode_$1_$2.log 2>&1
  # M = 32 .. 256: streaming (MT 1/2/4), tiled (16x16x32) and wide (32x32x16) kernels
  timeout 900 python tools/tune.py --shapes sq8192,qkv,gate_up,down --ms 32,64,128,256 --fmt $1 --dtype $2 --rotate-mb 640 --reps 3 --out $O/midm_$1_$2.json > $O/midm_$1_$2.log 2>&1
  # M = 512 (BASELINE config 5): tiled / wide kernels, with and without a K split across workgroups; hipBLASLt next to them
  timeout 600 python tools/tune.py --shapes sq8192,qkv,gate_up,down --ms 512 --kinds 0,8,12 --splitk 1,2 --splitk-kinds tiled --fmt $1 --dtype $2 --rotate-mb 640 --reps 5 --compare-dense --out $O/bigm_$1_$2.json > $O/bigm_$1_$2.log 2>&1
done
# the opt-in native-FP4 kernels (MXFP4 weights; activations quantised to MXFP8 / MXFP4), M = 512 and 2048
timeout 600 python tools/tune.py --shapes sq8192,qkv,gate_up,down --ms 512,2048 --fmt mx --dtype bf16 --native --kinds 9,13 --splitk 1,2 --no-check --rotate-mb 640 --reps 5 --out $O/native_mx_bf16.json > $O/native_mx_bf16.log 2>&1
for f in $O/*.log; do echo "== $f"; grep -v amdgpu.ids $f | grep "best\|DROPPED" | cut -c1-220; done
