#!/bin/bash
# round 5, GPU session 7: M = 8 / 16 (loader-wave shared-tile kernels, MT = 1) and M = 256 (batch kernels reach it) re-tuned on every table shape; batch parity
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "batch_kernels or every_solution or repeat_launch" > $O/pytest_sel.log 2>&1
tail -3 $O/pytest_sel.log
timeout 2400 python tools/build_table.py --ms 8,16,256 --families nv:bf16,nv:f16,mx:bf16 --out-dir $O/table > $O/table.log 2>&1
tail -2 $O/table.log
gzip -f $O/table/candidates_table.csv
du -sh $O
