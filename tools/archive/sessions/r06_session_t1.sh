# round 6, GPU session T1: the narrow table shapes (N <= 3072) at M = 256 / 512 / 1024 / 2048 with the table's own row among the candidates (110 of their rows name batched-decode kernels beyond the
# tuner's eight-m-block cap: never timed since the cap moved above the push, hence never challenged by the round's new forms); second session: t2
python tools/build_table.py --n-max 3072 --ms 256,512,1024,2048 --out-dir gpurun_out/r06_narrow_s1 --samples 3 2>&1 | tail -1
