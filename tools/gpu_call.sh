mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3) > gpurun_out/r1_tests.log
timeout 300 python bench.py > gpurun_out/bench_r01.json 2> gpurun_out/bench_r01.err
timeout 120 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -o bench -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_bench.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -o p -- python3 $R/bench.py --no-cpu-baseline --no-graph --steps 100 > $R/gpurun_out/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -o p -- python3 $R/bench.py --no-cpu-baseline --no-graph --steps 100 > $R/gpurun_out/pmc_write.log 2>&1
cd $R
tail -2 gpurun_out/r1_tests.log; cat gpurun_out/bench_r01.json; tail -2 gpurun_out/smoke.log; ls gpurun_out/prof_bench gpurun_out/pmc_fetch
