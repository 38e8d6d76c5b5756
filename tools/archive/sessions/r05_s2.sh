#!/bin/bash
# round 5, GPU session 2: the batch kernels against the oracle, mid-M counters of the round-4 picks, the mid-M buckets re-tuned with the batch kernels in
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05b; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "batch_kernels or every_solution or fused_silu or fused_bias or reference_gtest or repeat_launch" > $O/pytest_sel.log 2>&1
tail -15 $O/pytest_sel.log
PETIT_AMD_NO_TUNED= timeout 900 bash tools/collect_midm_evidence.sh $O/midm
timeout 1500 python tools/build_table.py --only llama3-70b,llama3-8b,r01-r03 --ms 32,64,128 --out-dir $O/table > $O/table.log 2>&1
tail -3 $O/table.log
find $O/midm -name "*.db" -delete 2>/dev/null; rm -rf $O/midm/*_p[0-9]/*/*.json 2>/dev/null; du -sh $O
