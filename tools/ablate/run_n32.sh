mkdir -p gpurun_out/r02i_n32
for abl in 0 1 2 4 8 16 7; do
  PETIT_AMD_LIB=$PWD/tools/ablate/n32/libpetit_abl_$abl.so timeout 300 python tools/tune.py --shapes sq8192,gate_up --ms 512 --fmt mx --dtype bf16 --native --kinds 13 --no-check --rotate-mb 640 --reps 3 --out gpurun_out/r02i_n32/abl_$abl.json > gpurun_out/r02i_n32/abl_$abl.log 2>&1
  echo "== abl $abl"; python - <<P
import csv
for r in csv.DictReader(open('gpurun_out/r02i_n32/abl_$abl.csv')):
    d=r['desc']
    if 'mxfp4)' in d and ('mb4 np1 waves1x4 d4 kt2 pf2' in d or 'mb4 np2 waves1x4 d4 kt2 pf2' in d) and 'splitk1' in d:
        print(r['shape'], r['us_median'], d[44:90])
P
done
