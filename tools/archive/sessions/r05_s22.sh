#!/bin/bash
# round 5, GPU session 22: the full GPU suite on the final table, the bench line both ways
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05v; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1
tail -3 $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
cp gpurun_out/bench_cells_full.json $O/bench_cells_full.json 2>/dev/null
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err
cp gpurun_out/bench_cells_full.json $O/bench_cells_full_steps20.json 2>/dev/null
tail -c 300 $O/bench_steps20.json
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
