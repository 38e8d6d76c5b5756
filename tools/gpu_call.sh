mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5) > gpurun_out/r1_tests.log
timeout 900 python tools/tune.py --shapes sq8192,sq4096,gate_up,down --ms 1,4 --out gpurun_out/tune_e.json > gpurun_out/tune_e.log 2>&1
timeout 300 python tools/ablate/run_ablate.py 1 > gpurun_out/ablate_e.log 2>&1
tail -2 gpurun_out/r1_tests.log; grep -v amdgpu.ids gpurun_out/tune_e.log | tail -9; grep -v amdgpu.ids gpurun_out/ablate_e.log | grep "variant 0"
