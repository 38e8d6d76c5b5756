set -x
mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > gpurun_out/r1_tests.log
timeout 900 python tools/tune.py --shapes sq8192,sq4096,qkv,gate_up,down --ms 1,4,16 --out gpurun_out/tune_c.json > gpurun_out/tune_c.log 2>&1
timeout 300 python tools/ablate/run_ablate.py 1 > gpurun_out/ablate_c.log 2>&1
tail -3 gpurun_out/r1_tests.log; grep -v amdgpu.ids gpurun_out/tune_c.log | tail -20; grep -v amdgpu.ids gpurun_out/ablate_c.log | grep "us$" 
