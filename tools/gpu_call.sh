mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -12) > gpurun_out/r1_tests.log
tail -12 gpurun_out/r1_tests.log
