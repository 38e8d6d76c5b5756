#!/bin/bash
# round 5, GPU session 23: fuzz of the bulk + tail launches, two more seeds of the general fuzz
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05w; mkdir -p $O
for seed in 1 2; do
  echo "# tools/fuzz_row_split.py $seed 300" >> $O/fuzz_row_split.txt
  timeout 500 python tools/fuzz_row_split.py $seed 300 >> $O/fuzz_row_split.txt 2>&1
done
grep -c "^FAIL" $O/fuzz_row_split.txt; grep "^ok [0-9]" $O/fuzz_row_split.txt
for seed in 73 74; do
  echo "# tools/fuzz_parity.py $seed 300" >> $O/fuzz.txt
  timeout 500 python tools/fuzz_parity.py $seed 300 >> $O/fuzz.txt 2>&1
done
grep -c FAIL $O/fuzz.txt; grep "^ok" $O/fuzz.txt
