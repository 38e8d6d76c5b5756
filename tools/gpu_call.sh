mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3) > gpurun_out/r1_tests.log
tail -3 gpurun_out/r1_tests.log
timeout 900 python tools/tune.py --shapes sq8192,gate_up,down --ms 8,16 --out gpurun_out/tune_pa.json > gpurun_out/tune_pa.log 2>&1
python - <<'PY'
import json
d=json.load(open('gpurun_out/tune_pa.json'))
for e in d['results']:
    print(e['shape'], 'M=',e['m'])
    for r in e['results'][:7]:
        if 'us_median' in r: print('   %7.2f us  %5.0f GB/s  %s %s'%(r['us_median'], r['gbs'], r['desc'][18:48]+r['desc'].split(')')[-1], '*' if r['is_default'] else ''))
PY
