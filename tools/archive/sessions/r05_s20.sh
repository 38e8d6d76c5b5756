#!/bin/bash
# round 5, GPU session 20: the record, third take (band raster, bulk + tail for ragged prefill M, fp16 x MXFP4 rows re-measured, neighbours ranked by grid fit)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05t; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1
tail -3 $O/pytest_gpu.log
cp gpurun_out/mlp_accuracy_budget*.json gpurun_out/stacked_mlp_accuracy_budget*.json $O/ 2>/dev/null
timeout 1800 bash tools/collect_profiles.sh r05 > $O/collect_profiles.log 2>&1
cp gpurun_out/bench_cells_full.json $O/bench_cells_full.json 2>/dev/null
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err
cp gpurun_out/bench_cells_full.json $O/bench_cells_full_steps20.json 2>/dev/null
tail -c 600 $O/bench_steps20.json
timeout 1500 python tools/build_table.py --part heldout --ms 1,2,4,8,16,32,64,128,256,512,1024,2048,8192 --out-dir $O/heldout > $O/heldout.log 2>&1
tail -1 $O/heldout.log
gzip -f $O/heldout/candidates_heldout.csv
for seed in 71 72; do
  echo "# tools/fuzz_parity.py $seed 420" >> $O/fuzz.txt
  timeout 700 python tools/fuzz_parity.py $seed 420 >> $O/fuzz.txt 2>&1
done
grep -c FAIL $O/fuzz.txt; grep "^ok" $O/fuzz.txt
find gpurun_out -name "*.db" -delete 2>/dev/null
du -sh $O
