set -x
mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -40) > gpurun_out/r1_tests.log
timeout 120 python tools/probes/run_probes.py > gpurun_out/probes.json 2> gpurun_out/probes.err
timeout 600 python tools/tune.py --shapes sq8192,sq4096 --ms 1,16 --out gpurun_out/tune_a.json > gpurun_out/tune_a.log 2>&1
timeout 300 python bench.py > gpurun_out/bench_a.json 2> gpurun_out/bench_a.err
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_a -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_a.log 2>&1
cd $GRAFT_REPO_ROOT; tail -5 gpurun_out/r1_tests.log; cat gpurun_out/bench_a.json; tail -12 gpurun_out/tune_a.log
