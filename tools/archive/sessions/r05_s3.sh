#!/bin/bash
# round 5, GPU session 3: batch kernels (both forms) vs the oracle; the scaled-MFMA alignment probe; checkpoint-like accuracy budgets; mid-M re-tune
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05c; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "batch_kernels or checkpoint_like or every_solution or fused_silu" > $O/pytest_sel.log 2>&1
tail -12 $O/pytest_sel.log
cp gpurun_out/mlp_accuracy_budget_checkpoint_like.json gpurun_out/stacked_mlp_accuracy_budget_checkpoint_like.json $O/ 2>/dev/null
(cd tools/probes && hipcc -O2 --offload-arch=gfx950 mfma_scale_align.hip -o mfma_scale_align && timeout 300 ./mfma_scale_align) > $O/mfma_scale_align.txt 2>&1
grep SUMMARY $O/mfma_scale_align.txt
timeout 1500 python tools/build_table.py --only llama3-70b,r01-r03 --ms 32,64,128 --families nv:bf16,mx:bf16 --out-dir $O/table > $O/table.log 2>&1
tail -2 $O/table.log
du -sh $O
