#!/bin/bash
# round 5, GPU session 13: M = 512 / 1024 re-tuned on every table shape with the batched-decode kernels offered (128 x 128 form)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05m; mkdir -p $O
timeout 2400 python tools/build_table.py --ms 512,1024 --samples 3 --out-dir $O/table > $O/table.log 2>&1
tail -2 $O/table.log
gzip -f $O/table/candidates_table.csv
du -sh $O
