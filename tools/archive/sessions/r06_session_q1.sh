# round 6, GPU session Q1: the prefill buckets (measured at M = 1024 / 2048 / 8192) on EVERY table shape with the 128 x 320 form among the candidates (m1 / m2 covered N % 320 = 0 only; at M = 512 the
# form won 27 rows, most of them on N it does not divide); second session: q2
python tools/build_table.py --ms 1024,2048,8192 --out-dir gpurun_out/r06_prefill320_s1 --samples 3 2>&1 | tail -1
