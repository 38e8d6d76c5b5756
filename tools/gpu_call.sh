for mb in 40 160 320 640 1280 2560; do echo "rotate $mb MB"; timeout 300 python tools/tune.py --shapes sq8192 --ms 1 --only-default --rotate-mb $mb --out gpurun_out/tune_x.json 2>&1 | grep "best" | cut -c1-60; done
for mb in 320 1280; do timeout 300 python bench.py --no-cpu-baseline --rotate-mb $mb 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench rotate',d['config']['weights_rotated_over_copies'],'us/step',round(d['ms_per_step']*1e3,2))"; done
