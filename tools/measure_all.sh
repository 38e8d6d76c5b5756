#!/bin/bash
# tools/measure_all.sh [round-tag] -- on the MI355X box: the sweeps the arch table and the numbers in
# DESIGN.md come from.  Outputs land in gpurun_out/<tag>_*; copy the ones to keep into profiles/.
TAG=${1:-r01}
mkdir -p gpurun_out
SH="sq8192,sq4096,qkv,gate_up,down"
python tools/tune.py --shapes $SH --ms 1,2,4,8,16 --fmt nv --dtype bf16 --out gpurun_out/${TAG}_tune_nv_bf16.json > gpurun_out/${TAG}_tune_nv_bf16.log 2>&1
python tools/tune.py --shapes $SH --ms 1,2,4,8,16 --fmt nv --dtype f16  --out gpurun_out/${TAG}_tune_nv_f16.json  > gpurun_out/${TAG}_tune_nv_f16.log 2>&1
python tools/tune.py --shapes $SH --ms 1,2,4,8,16 --fmt mx --dtype bf16 --out gpurun_out/${TAG}_tune_mx_bf16.json > gpurun_out/${TAG}_tune_mx_bf16.log 2>&1
python tools/tune.py --shapes $SH --ms 1,2,4,8,16 --fmt mx --dtype f16  --out gpurun_out/${TAG}_tune_mx_f16.json  > gpurun_out/${TAG}_tune_mx_f16.log 2>&1
python tools/tune.py --shapes sq8192,qkv,gate_up,down --ms 64,128,512 --fmt nv --dtype bf16 --compare-dense --rotate-mb 640 --out gpurun_out/${TAG}_tune_bigm_nv_bf16.json > gpurun_out/${TAG}_tune_bigm_nv_bf16.log 2>&1
python tools/tune.py --shapes sq8192,gate_up --ms 128,512 --fmt mx --dtype bf16 --compare-dense --rotate-mb 640 --out gpurun_out/${TAG}_tune_bigm_mx_bf16.json > gpurun_out/${TAG}_tune_bigm_mx_bf16.log 2>&1
for f in nv_bf16 nv_f16 mx_bf16 mx_f16 bigm_nv_bf16 bigm_mx_bf16; do echo "== $f"; grep -v amdgpu.ids gpurun_out/${TAG}_tune_$f.log | grep best | cut -c1-175; done
