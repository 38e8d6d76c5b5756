mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5) > gpurun_out/r1_tests.log
timeout 900 python tools/tune.py --shapes sq8192,sq4096,gate_up --ms 1,4,16 --out gpurun_out/tune_d.json > gpurun_out/tune_d.log 2>&1
tail -3 gpurun_out/r1_tests.log; grep -v amdgpu.ids gpurun_out/tune_d.log | tail -12
