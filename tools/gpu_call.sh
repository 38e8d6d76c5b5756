mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3) > gpurun_out/r1_tests.log
timeout 1500 python tools/tune.py --shapes sq8192,sq4096,qkv,gate_up,down --ms 1,2,4,8,16 --fmt nv --dtype bf16 --out gpurun_out/tune_nv_bf16.json > gpurun_out/tune_nv_bf16.log 2>&1
timeout 1500 python tools/tune.py --shapes sq8192,sq4096,qkv,gate_up,down --ms 1,2,4,8,16 --fmt nv --dtype f16 --out gpurun_out/tune_nv_f16.json > gpurun_out/tune_nv_f16.log 2>&1
timeout 1500 python tools/tune.py --shapes sq8192,sq4096,qkv,gate_up,down --ms 1,2,4,8,16 --fmt mx --dtype bf16 --out gpurun_out/tune_mx_bf16.json > gpurun_out/tune_mx_bf16.log 2>&1
tail -2 gpurun_out/r1_tests.log; for f in nv_bf16 nv_f16 mx_bf16; do echo == $f; grep -v amdgpu.ids gpurun_out/tune_$f.log | tail -27 | cut -c1-150; done
