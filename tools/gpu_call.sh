set -x
mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > gpurun_out/r1_tests.log
timeout 600 python tools/tune.py --shapes sq8192,sq4096 --ms 1,4,16 --out gpurun_out/tune_b.json > gpurun_out/tune_b.log 2>&1
timeout 300 python tools/ablate/run_ablate.py 1,4 > gpurun_out/ablate_b.log 2>&1
tail -5 gpurun_out/r1_tests.log; grep -v amdgpu.ids gpurun_out/tune_b.log | tail; grep -v amdgpu.ids gpurun_out/ablate_b.log | grep "us$" 
