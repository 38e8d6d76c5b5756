# round 6, GPU session K: the committed tree after the last rebuild (comment-only and instance-list round trips since session E): whole suite, smoke, the driver's bench form
python -m pytest tests -m gpu -q > gpurun_out/r06_gputest_k.log 2>&1; tail -3 gpurun_out/r06_gputest_k.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 > gpurun_out/r06k_bench_steps20.json 2> gpurun_out/r06k_bench_steps20.err; tail -1 gpurun_out/r06k_bench_steps20.err; head -c 400 gpurun_out/r06k_bench_steps20.json
