#!/bin/bash
# round 5, GPU session 1: parity of the new cells, the bench with mid-M / prefill cells (baseline of the round), mid-M counters, prefill tuning
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05a; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "bench_cells_parity or native_silu_mul_with or fused_silu or autotune_on_first_sight or default_solution_on_unseen" > $O/pytest_sel.log 2>&1
tail -5 $O/pytest_sel.log
timeout 900 python bench.py --verbose > $O/bench.json 2> $O/bench.err
cp gpurun_out/bench_cells_full.json $O/ 2>/dev/null
tail -c 1500 $O/bench.json
timeout 900 bash tools/collect_midm_evidence.sh $O/midm
timeout 1500 python tools/build_table.py --only llama3-70b,llama3-8b,r01-r03 --ms 1024,2048,8192 --samples 3 --out-dir $O/table > $O/table.log 2>&1
tail -3 $O/table.log
# keep the merged output small: drop the raw rocprof directories' big files
find $O/midm -name "*.db" -delete 2>/dev/null; du -sh $O
