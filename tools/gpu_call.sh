mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4) > gpurun_out/r1_tests.log
tail -4 gpurun_out/r1_tests.log
timeout 900 python tools/tune.py --shapes sq8192,gate_up,down --ms 1,2,4 --out gpurun_out/tune_bfp.json > gpurun_out/tune_bfp.log 2>&1
python - <<'PY'
import json
d=json.load(open('gpurun_out/tune_bfp.json'))
for e in d['results']:
    print(e['shape'], 'M=',e['m'])
    for r in e['results'][:6]:
        if 'us_median' in r: print('   %7.2f us  %5.0f GB/s  %s %s'%(r['us_median'], r['gbs'], r['desc'][18:70], '*' if r['is_default'] else ''))
PY
