#!/bin/bash
# session 19: fp16 x MXFP4 rows re-measured with the saturating output check (33 prefill rows named the streaming reference), then the full GPU suite
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05s; mkdir -p $O
timeout 1500 python tools/build_table.py --families mx:f16 --ms 128,256,512,1024,2048,8192 --samples 3 --out-dir $O/table > $O/table.log 2>&1
tail -1 $O/table.log
gzip -f $O/table/candidates_table.csv
timeout 2400 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1
tail -3 $O/pytest_gpu.log
