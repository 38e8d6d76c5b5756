#!/bin/bash
# round 5, GPU session 13: M = 512 / 1024 re-tuned on every table shape with the batched-decode kernels offered (128 x 128 form)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05m; mkdir -p $O
timeout 2400 python tools/build_table.py --ms 512,1024 --samples 3 --out-dir $O/table > $O/table.log 2>&1
tail -2 $O/table.log
gzip -f $O/table/candidates_table.csv
du -sh $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "native" > $O/pytest_native.log 2>&1
tail -3 $O/pytest_native.log
for seed in 51 52; do
  echo "# tools/fuzz_parity.py $seed 300" >> $O/fuzz.txt
  timeout 600 python tools/fuzz_parity.py $seed 300 >> $O/fuzz.txt 2>&1
done
grep -c FAIL $O/fuzz.txt; grep "^ok" $O/fuzz.txt
