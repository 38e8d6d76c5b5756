for i in 1 2 3; do timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench copies',d['config']['weights_rotated_over_copies'],'steps',d['steps'],'warmup',d['warmup'],'us/step',round(d['ms_per_step']*1e3,2), d['config']['solution'][:20])"; done
timeout 300 python bench.py --no-cpu-baseline --steps 400 --warmup 400 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench steps',d['steps'],'warmup',d['warmup'],'us/step',round(d['ms_per_step']*1e3,2))"
timeout 300 python tools/tune.py --shapes sq8192 --ms 1 --only-default --out gpurun_out/tune_x.json 2>&1 | grep "best" | cut -c1-60
