mkdir -p gpurun_out
timeout 900 python tools/tune.py --shapes sq8192,gate_up --ms 1 --out gpurun_out/tune_f.json > gpurun_out/tune_f.log 2>&1
python - <<'PY'
import json
d=json.load(open('gpurun_out/tune_f.json'))
for e in d['results']:
    print(e['shape'], 'M=',e['m'])
    for r in e['results']:
        if 'us_median' in r: print('   %7.2f us  %5.0f GB/s  %s'%(r['us_median'], r['gbs'], r['desc'][18:56]))
PY
