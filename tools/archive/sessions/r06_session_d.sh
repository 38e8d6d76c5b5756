# round 6, GPU session D: the tests that failed / changed, the sibling-vs-own-split A/B, fuzz with NVFP4 native
python -m pytest tests -m gpu -q -k "bench_cells_parity or autotune or examples or tp8 or table_rows_sampled or auto_row_split" > gpurun_out/r06_gputest_d.log 2>&1; tail -4 gpurun_out/r06_gputest_d.log
for fmt in nv mx; do for sh in "8192 8192" "10240 8192" "57344 8192" "8192 28672"; do set -- $sh
  for m in 2084 4314; do
    python tools/time_ids.py --m $m --n $1 --k $2 --fmt $fmt --tag split_on auto 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06_sibling_vs_split.jsonl | cut -c1-170
    PETIT_AMD_NO_ROW_SPLIT=1 python tools/time_ids.py --m $m --n $1 --k $2 --fmt $fmt --tag split_off auto 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06_sibling_vs_split.jsonl | cut -c1-170
  done; done; done
python tools/fuzz_parity.py 6 240 2>&1 | tail -6 | cut -c1-400
