# round 6, GPU session P1: the 257-512 bucket re-measured on the final library (its rows predate the group-ahead forms of gemm_wide; the 128 x 320 form was measured there for N % 320 = 0 only); second session: p2
python tools/build_table.py --ms 512 --out-dir gpurun_out/r06_m512_s1 --samples 3 2>&1 | tail -1
