# round 6, GPU session E6: the record on the final tree (29 rows for K % 1024 != 0 shapes and their four kernel instances in)
python -m pytest tests -m gpu -q --durations=5 > gpurun_out/r06_gputest_e6.log 2>&1; tail -10 gpurun_out/r06_gputest_e6.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_smoke.log 2>&1; tail -1 gpurun_out/r06_smoke.log
bash tools/collect_profiles.sh r06 > gpurun_out/r06_collect_profiles.log 2>&1; tail -c 300 gpurun_out/r06_collect_profiles.log
cp gpurun_out/bench_cells_full.json gpurun_out/r06_bench_cells_full.json
python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_steps20.json 2> gpurun_out/r06_bench_steps20.err; tail -1 gpurun_out/r06_bench_steps20.err
cp gpurun_out/bench_cells_full.json gpurun_out/r06_bench_cells_full_steps20.json
rm -f gpurun_out/r06_reference_list.jsonl gpurun_out/r06_reference_list_bf16.jsonl
python tools/reference_list_sweep.py --atype fp16 --btype nv --native --out gpurun_out/r06_reference_list.jsonl > /dev/null 2>&1
python tools/reference_list_sweep.py --atype bf16 --btype nv --out gpurun_out/r06_reference_list_bf16.jsonl > /dev/null 2>&1
wc -l gpurun_out/r06_reference_list*.jsonl
