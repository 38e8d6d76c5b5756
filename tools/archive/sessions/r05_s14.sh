#!/bin/bash
# round 5, GPU session 14: the record, second take (after the M = 512 / 1024 rows, the ragged-M rule and the FP8 group bound)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05n; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1
tail -3 $O/pytest_gpu.log
cp gpurun_out/mlp_accuracy_budget*.json gpurun_out/stacked_mlp_accuracy_budget*.json $O/ 2>/dev/null
timeout 1800 bash tools/collect_profiles.sh r05 > $O/collect_profiles.log 2>&1
cp gpurun_out/bench_cells_full.json $O/bench_cells_full.json 2>/dev/null
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err
cp gpurun_out/bench_cells_full.json $O/bench_cells_full_steps20.json 2>/dev/null
tail -c 900 $O/bench_steps20.json
for seed in 61 62 63; do
  echo "# tools/fuzz_parity.py $seed 420" >> $O/fuzz.txt
  timeout 700 python tools/fuzz_parity.py $seed 420 >> $O/fuzz.txt 2>&1
done
grep -c FAIL $O/fuzz.txt; grep "^ok" $O/fuzz.txt
find gpurun_out -name "*.db" -delete 2>/dev/null
du -sh $O
