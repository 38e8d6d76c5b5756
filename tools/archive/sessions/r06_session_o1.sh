# round 6, GPU session O1: the 65-128 and 129-256 buckets re-measured on the final library (their rows predate the MT = 3 batched-decode instances, which win 8 / 36 and 4 / 36 held-out
# problems at M = 128 / 256: profiles/r06_heldout_candidates.csv.gz); second session: o2
python tools/build_table.py --ms 128,256 --out-dir gpurun_out/r06_m128_256_s1 --samples 3 2>&1 | tail -1
