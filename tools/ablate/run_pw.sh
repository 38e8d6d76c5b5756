#!/bin/bash
# tools/ablate/run_pw.sh -- prefetch-wave experiment (PETIT_N32_PW / PETIT_N32_PWSHARE builds of the bf16 x MXFP4 TU under tools/ablate/pw/)
mkdir -p gpurun_out/pw
for v in ${@:-base 4_4 8_1 8_4 16_4}; do
  lib=$PWD/tools/ablate/pw/libpetit_pw_$v.so
  [ $v = base ] && lib=$PWD/petit-kernel_amd/lib/libpetit_amd.so
  PETIT_AMD_LIB=$lib timeout 600 python tools/tune.py --shapes o,qkv,down,gate_up --ms 512 --fmt mx --dtype bf16 --native --kinds 13 --splitk 1 --rotate-mb 640 --reps 3 --out gpurun_out/pw/$v.json > gpurun_out/pw/$v.log 2>&1
  python - <<P
import csv
best = {}
for r in csv.DictReader(open('gpurun_out/pw/$v.csv')):
    if '+loader' not in r['desc'] and '$ALL' != '1': continue
    if r['checked'] != 'ok': print("NOT OK", r['solution'], r['checked'])
    key = (r['shape'], 'fp8' if 'mxfp8' in r['desc'] else 'fp4')
    us = float(r['us_median'])
    if key not in best or us < best[key][0]: best[key] = (us, r['solution'])
print("$v", {k: v for k, v in sorted(best.items())})
P
  grep -c -i "mismatch\|FAIL\|dropped [1-9]" gpurun_out/pw/$v.log
done
