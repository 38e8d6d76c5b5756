# round 6, GPU session N: the ten held-out shapes tuned on the final library (every kernel of round 6 in: MT = 3 batch forms, group-ahead 256 x 256, 128 x 320) -- what an unseen shape gets,
# replayed afterwards without a GPU by tools/check_heuristic.py --heldout --mode nearest --by-m
python tools/build_table.py --part heldout --ms 1,2,4,8,16,32,48,64,128,256,512,1024,2048,8192 --out-dir gpurun_out/r06_heldout --samples 3 2>&1 | tail -1
