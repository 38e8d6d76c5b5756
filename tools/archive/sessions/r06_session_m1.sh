# round 6, GPU session M1: the 128 x 320 tiled instance (N = 10240 / 5120 / 1280 ... in whole column tiles): parity of the tests that enumerate kernels, then a tuner session over the
# 18 table shapes with N % 320 == 0 at the prefill M of the table's buckets (second session: m2)
python -m pytest tests -m gpu -q -k "every_solution or gtest_problem_list or m512_full_size" > gpurun_out/r06_gputest_m1.log 2>&1; tail -3 gpurun_out/r06_gputest_m1.log
python tools/build_table.py --n-multiple 320 --ms 512,1024,2048,8192 --out-dir gpurun_out/r06_t320_s1 --samples 3 2>&1 | tail -1
