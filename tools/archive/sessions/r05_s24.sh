#!/bin/bash
# round 5, GPU session 24: the 48-row batched-decode tiles (probe), the bulk + tail fuzz again (its SiLU-mul inputs overflowed fp16)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05x; mkdir -p $O
timeout 600 python tools/mt3_probe.py > $O/mt3_probe.jsonl 2> $O/mt3_probe.err
cat $O/mt3_probe.jsonl | cut -c1-260
echo "# tools/fuzz_row_split.py 3 240" >> $O/fuzz_row_split.txt
timeout 400 python tools/fuzz_row_split.py 3 240 >> $O/fuzz_row_split.txt 2>&1
grep -c "^FAIL" $O/fuzz_row_split.txt; grep "^ok [0-9]" $O/fuzz_row_split.txt
timeout 600 python -m pytest tests -q -m gpu -x -k "batch_kernels" > $O/pytest_batch.log 2>&1; tail -2 $O/pytest_batch.log
