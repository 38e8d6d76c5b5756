#!/bin/bash
# round 5, GPU session 8: the prefill buckets (and 257-512) measured on EVERY table shape; fp16 x MXFP4 batch kernels vs the oracle; mid-M rows of fp16 x MXFP4
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "batch_kernels or fp16_mxfp4_every_solution or repeat_launch" > $O/pytest_sel.log 2>&1
tail -3 $O/pytest_sel.log
timeout 1200 python tools/build_table.py --ms 32,64,128,256 --families mx:f16 --out-dir $O/table_mxf16 > $O/table_mxf16.log 2>&1
tail -1 $O/table_mxf16.log
timeout 3000 python tools/build_table.py --ms 512,1024,2048,8192 --samples 3 --out-dir $O/table > $O/table.log 2>&1
tail -2 $O/table.log
gzip -f $O/table/candidates_table.csv $O/table_mxf16/candidates_table.csv
du -sh $O
