#!/bin/bash
# round 5, GPU session 4: ablation + counters of the batched-decode kernels (what binds at M = 32 / 64 / 128 on `o`), the tests session 3 did not reach
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05d; mkdir -p $O
ID32=1810422113500202; ID64=1420422113500402; ID128=1420222113500404; ID64B=1810242113500402
for abl in 0 1 2 4 8 16 6 14 30 31; do
  for spec in "32 $ID32" "64 $ID64" "64 $ID64B" "128 $ID128"; do
    set -- $spec
    PETIT_AMD_LIB=$PWD/tools/ablate/batch/libpetit_abl_$abl.so timeout 120 python tools/time_ids.py --m $1 --n 8192 --k 8192 --tag abl$abl $2 >> $O/ablate_o.jsonl 2>> $O/ablate.err
  done
done
cat $O/ablate_o.jsonl | cut -c1-200
SOL_o_m32=$ID32 SOL_o_m64=$ID64 SOL_qkv_m32=auto SOL_qkv_m64=1810432113300304 EXTRA_SPECS="o_m128:128:8192:8192" timeout 900 bash tools/collect_midm_evidence.sh $O/midm
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "fused_silu or checkpoint_like or native_silu_mul_with" > $O/pytest_sel.log 2>&1
tail -5 $O/pytest_sel.log
cp gpurun_out/mlp_accuracy_budget_checkpoint_like.json gpurun_out/stacked_mlp_accuracy_budget_checkpoint_like.json $O/ 2>/dev/null
find $O/midm -name "*.db" -delete 2>/dev/null; rm -f $O/midm/*_p[0-9]/*/*.json $O/midm/*_p[0-9]/*agent_info.csv 2>/dev/null; du -sh $O
