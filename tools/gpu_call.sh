mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4) > gpurun_out/r1_tests.log
tail -4 gpurun_out/r1_tests.log
timeout 900 python tools/tune.py --shapes sq8192,gate_up,down --ms 64,128,512 --compare-dense --rotate-mb 320 --out gpurun_out/tune_bigm.json 2>&1 | grep -v amdgpu | cut -c1-260 | tail -12
