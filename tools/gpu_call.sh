for k in 20 50 100 250 400 500 800 1000 2000 4000; do timeout 300 python bench.py --no-cpu-baseline --steps $k --warmup 2000 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench steps',d['steps'],'warmup',d['warmup'],'us/step',round(d['ms_per_step']*1e3,2),'wall',round(d['wall_ms_per_step']*1e3,2))"; done
