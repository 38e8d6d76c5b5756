mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3) > gpurun_out/r1_tests.log
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench_x1.json 2> gpurun_out/bench_x1.err
timeout 300 python tools/tune.py --shapes sq8192 --ms 1 --only-default --out gpurun_out/tune_x.json 2>&1 | grep -v amdgpu | tail -2
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench_x2.json 2> gpurun_out/bench_x2.err
timeout 300 python tools/dbg_tmp.py 2>&1 | grep rotating
tail -2 gpurun_out/r1_tests.log; python -c "
import json
for f in ('gpurun_out/bench_x1.json','gpurun_out/bench_x2.json'):
    d=json.load(open(f)); print(f, d['ms_per_step']*1e3, 'us', d['value'])"
