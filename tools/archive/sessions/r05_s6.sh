#!/bin/bash
# round 5, GPU session 6: full GPU suite, the bench (long + driver form), held-out shapes for the heuristic check, profiles of the headline
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05f; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log
cp gpurun_out/mlp_accuracy_budget*.json gpurun_out/stacked_mlp_accuracy_budget*.json $O/ 2>/dev/null
timeout 900 python bench.py --verbose > $O/bench.json 2> $O/bench.err
cp gpurun_out/bench_cells_full.json $O/bench_cells_full.json 2>/dev/null
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err
tail -c 1200 $O/bench_steps20.json
timeout 1500 python tools/build_table.py --part heldout --ms 1,2,4,8,16,32,64,128,256,512,1024,2048,8192 --out-dir $O/heldout > $O/heldout.log 2>&1
tail -2 $O/heldout.log
gzip -f $O/heldout/candidates_heldout.csv
du -sh $O
