# round 6, GPU session C: the whole suite (log kept, no -x), then the bench as the driver runs it
python -m pytest tests -m gpu -q > gpurun_out/r06_gputest_c.log 2>&1; tail -5 gpurun_out/r06_gputest_c.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r06c_bench_steps20.json 2> gpurun_out/r06c_bench_steps20.err; tail -c 1500 gpurun_out/r06c_bench_steps20.json; tail -3 gpurun_out/r06c_bench_steps20.err
cp gpurun_out/bench_cells_full.json gpurun_out/r06c_bench_cells_full_steps20.json
