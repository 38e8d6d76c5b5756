#!/usr/bin/env python3
"""tools/check_inlib_tune.py [out.json] -- does petit_kernel.tune (csrc/tune.hip) find the best kernel?

For shapes no built-in row names (the four "unseen" shapes of DESIGN.md section 3.5) and M in {1, 4, 16, 128}:
  heuristic : what solution_id = -1 runs before tuning (api.hip heuristic), timed with tools/benchlib.py;
  tuned     : what it runs after ONE petit_kernel.tune_tensors call, timed the same way;
  best      : the fastest of ALL enumerated kernels x K splits, timed the same way (the offline sweep of tools/tune.py).
Done = tuned within 3 % of best.  Also reports how long the tune call itself took.
"""
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "petit-kernel_amd"))
sys.path.insert(0, str(ROOT / "tools"))
import torch

import benchlib as BL
import petit_kernel as pk
from petit_kernel import _lib

SHAPES = [(12288, 4096), (5120, 8192), (16384, 5120), (5120, 13824)]
MS = (1, 4, 16, 128)


def main():
    out = Path(sys.argv[1]) if len(sys.argv) > 1 else ROOT / "gpurun_out" / "inlib_tune.json"
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(dev)
    rows = []
    for (n, k) in SHAPES:
        w = BL.Weights("nv", n, k, 1280, dev)
        for m in MS:
            g = BL.Gemm(w, m, torch.bfloat16, dev)
            heur = g.resolve(_lib.PETIT_SOLUTION_AUTO)
            t_heur = g.time(heur, stream, reps=5)["us"]
            t0 = time.time()
            with torch.cuda.stream(stream):
                tuned, us_lib = pk.tune_tensors(g.a, w.packed, g.gs, m, n, k, "nvfp4")
            tune_s = time.time() - t0
            assert g.resolve(_lib.PETIT_SOLUTION_AUTO) == tuned
            t_tuned = g.time(tuned, stream, reps=5)["us"]
            best, t_best = None, 1e30
            for sid in g.solutions():
                for sk in (1, 2, 4, 8):
                    cand = (sid & ~(0xF << 60)) | (sk << 60)
                    try:
                        t = g.time(cand, stream, reps=3)["us"]
                    except Exception:  # noqa: BLE001 -- a kind that takes no K split
                        continue
                    if t < t_best:
                        best, t_best = cand, t
            t_best = min(t_best, g.time(best, stream, reps=5)["us"])
            row = {"n": n, "k": k, "m": m, "heuristic": f"{heur:x}", "us_heuristic": round(t_heur, 2), "tuned": f"{tuned:x}",
                   "us_tuned": round(t_tuned, 2), "us_tuned_as_timed_by_the_library": round(us_lib, 2), "best": f"{best:x}",
                   "us_best": round(t_best, 2), "tuned_over_best": round(t_tuned / t_best, 3), "heuristic_over_best": round(t_heur / t_best, 3),
                   "tune_call_seconds": round(tune_s, 2)}
            rows.append(row)
            print(json.dumps(row), flush=True)
        del w
        torch.cuda.empty_cache()
    worst = max(r["tuned_over_best"] for r in rows)
    out.parent.mkdir(parents=True, exist_ok=True)
    out.write_text(json.dumps({"cells": rows, "worst_tuned_over_best": worst,
                               "worst_heuristic_over_best": max(r["heuristic_over_best"] for r in rows),
                               "method": "tools/benchlib.py timing for all three columns; tune = petit_kernel.tune_tensors (csrc/tune.hip)"}, indent=1))
    print(f"worst tuned/best = {worst}")


if __name__ == "__main__":
    main()
