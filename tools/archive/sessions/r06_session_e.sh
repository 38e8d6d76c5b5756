# round 6, GPU session E: the record on the final tree -- whole suite (log kept), smoke, bench default + rocprofv3 stats + PMC passes (collect_profiles),
# the bench as the driver runs it, then the reference's benchmark list (tools/reference_list_sweep.py)
python -m pytest tests -m gpu -q > gpurun_out/r06_gputest_e.log 2>&1; tail -4 gpurun_out/r06_gputest_e.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_smoke.log 2>&1; tail -2 gpurun_out/r06_smoke.log
bash tools/collect_profiles.sh r06 > gpurun_out/r06_collect_profiles.log 2>&1; tail -c 600 gpurun_out/r06_collect_profiles.log
cp gpurun_out/bench_cells_full.json gpurun_out/r06_bench_cells_full.json
python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_steps20.json 2> gpurun_out/r06_bench_steps20.err; tail -2 gpurun_out/r06_bench_steps20.err
cp gpurun_out/bench_cells_full.json gpurun_out/r06_bench_cells_full_steps20.json
python tools/reference_list_sweep.py --atype fp16 --btype nv --native --out gpurun_out/r06_reference_list.jsonl > gpurun_out/r06_reference_list.log 2>&1; tail -2 gpurun_out/r06_reference_list.log | cut -c1-300
