#!/usr/bin/env python3
"""Launch the hipBLASLt bf16 comparator (tools/comparators/hipblaslt_gemm.cc) a few times, eagerly, for profiling under rocprofv3."""
import argparse
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
import torch

import benchlib as BL

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=57344)
ap.add_argument("--k", type=int, default=8192)
ap.add_argument("--m", type=int, default=512)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--algo", default="first", help="first (what a plain caller runs), best (the fastest of the heuristic's results, timed here first), or an index")
ap.add_argument("--fp8", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda", 0)
hb = BL.HipblasLtGemm(a.m, a.n, a.k, torch.float8_e4m3fn if a.fp8 else torch.bfloat16, dev, rotate_mb=640)
hb.check()
torch.cuda.synchronize()
if a.algo != "first":
    stream = torch.cuda.current_stream()
    idx = hb.time_best(stream, hb.time(stream, reps=3))["algo_index"] if a.algo == "best" else int(a.algo)
    BL.HipblasLtGemm._lib.hbl_select(hb.h, idx)
    print("algo index", idx, "of", BL.HipblasLtGemm._lib.hbl_count(hb.h))
for i in range(a.iters):
    hb.launch(i)
torch.cuda.synchronize()
print("done")
