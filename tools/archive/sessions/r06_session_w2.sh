# round 6, GPU session W2: the 128 x 256 / 128 x 320 tiled forms for K % 1024 != 0 (KS = 4 / 2: T(4 | 2, 8, 4 | 5, 4, 2)): parity of the tests that enumerate kernels, then a tuner session
# over the 15 table shapes with such a K at M = 512 / 1024 / 2048 / 8192; second session: w2
python tools/build_table.py --k-not-multiple 1024 --ms 512,1024,2048,8192 --out-dir gpurun_out/r06_ks4_s2 --samples 3 2>&1 | tail -1
