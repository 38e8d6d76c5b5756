# round 6, GPU session U1: every table shape at M = 256 / 512 / 1024 / 2048 with the table's own row among the candidates (110 rows name batched-decode kernels beyond the tuner's
# eight-m-block cap -- N from 576 to 9216 -- and had not been timed since the cap moved above the push); second session: u2
python tools/build_table.py --ms 256,512,1024,2048 --out-dir gpurun_out/r06_incumbent_s1 --samples 3 2>&1 | tail -1
