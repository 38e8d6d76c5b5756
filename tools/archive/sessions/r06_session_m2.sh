# round 6, GPU session M2: the 128 x 320 tiled instance (N = 10240 / 5120 / 1280 ... in whole column tiles): parity of the tests that enumerate kernels, then a tuner session over the
# 18 table shapes with N % 320 == 0 at the prefill M of the table's buckets (second session: m2)
python tools/build_table.py --n-multiple 320 --ms 512,1024,2048,8192 --out-dir gpurun_out/r06_t320_s2 --samples 3 2>&1 | tail -1
