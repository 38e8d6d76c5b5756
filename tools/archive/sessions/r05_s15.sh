#!/bin/bash
# round 5, GPU session 15: 224- / 112-column batch instances (WN = 7) on the wide-N shapes at M = 32 ... 256; parity of the new instances
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05o; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "batch_kernels" > $O/pytest_sel.log 2>&1
tail -2 $O/pytest_sel.log
timeout 1800 python tools/build_table.py --only "gate_up,down" --ms 32,64,128,256 --out-dir $O/table > $O/table.log 2>&1
tail -2 $O/table.log
gzip -f $O/table/candidates_table.csv
