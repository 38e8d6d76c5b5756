#!/bin/bash
# tools/ablate/run_n32_r03.sh -- ablation of three native 32x32x64 kernel structures on `o` (8192 x 8192, M = 512, FP4 x FP4): plain (four waves), two K
# groups, loader wave; libraries from `bash tools/ablate_native32.sh 0 1 2 3 8 16` (PETIT_ABLATE_N32 bits: 1 no activation DMA, 2 no W refills, 8 no
# MFMAs, 16 no stores).  us per call incl. the quantiser launch.
mkdir -p gpurun_out/r03_n32
for abl in 0 1 2 3 8 16; do
  PETIT_AMD_LIB=$PWD/tools/ablate/n32/libpetit_abl_$abl.so timeout 300 python tools/tune.py --shapes o --ms 512 --fmt mx --dtype bf16 --native --kinds 13 --no-check --rotate-mb 640 --reps 3 --out gpurun_out/r03_n32/abl_$abl.json > gpurun_out/r03_n32/abl_$abl.log 2>&1
  python - <<P
import csv
want = {"0x142da41623300804": "plain kt2 pf2 d4", "0x142d643623300804": "two K groups kt1 pf2 d4", "0x142da44623300804": "loader kt2 pf2 d4", "0x142d644623300804": "loader kt1 pf2 d4"}
for r in csv.DictReader(open('gpurun_out/r03_n32/abl_$abl.csv')):
    if r['solution'] in want:
        print("abl $abl", want[r['solution']], r['us_median'])
P
done
