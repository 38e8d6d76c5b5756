#!/usr/bin/env python3
"""tools/mt3_probe.py -- are 48-row (MT = 3) batched-decode tiles worth a sub-bucket 33-48?  Times the MT = 3 instances (every K split that runs) against the default pick at M = 36 / 44 / 48
on the four Llama-70B shapes, bf16 x NVFP4 / MXFP4 (tools/benchlib.py timing: HIP-graph replay over rotating weights, median)."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
import torch

import benchlib as BL
from petit_kernel import _lib

dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(dev)
for fmt in ("nv", "mx"):
    for shape, (n, k) in BL.LLAMA70B.items():
        w = BL.Weights(fmt, n, k, 1280, dev)
        for m in (36, 44, 48):
            g = BL.Gemm(w, m, torch.bfloat16, dev)
            auto = g.time(_lib.PETIT_SOLUTION_AUTO, stream, reps=5)["us"]
            best = None
            for sid in g.solutions():
                if (sid >> 48) & 0xF == 0 and (sid >> 36) & 0xF == 2 and (sid & 0xFF) == 3:
                    for sk in (1, 2, 4):
                        s2 = (sid & ~(0xF << 60)) | (sk << 60)
                        try:
                            us = g.time(s2, stream, reps=5)["us"]
                        except Exception:  # noqa: BLE001
                            continue
                        if best is None or us < best[0]:
                            best = (us, s2)
            print(json.dumps({"fmt": fmt, "shape": shape, "m": m, "auto_us": round(auto, 2), "auto": _lib.describe_solution(g.resolve(_lib.PETIT_SOLUTION_AUTO)).split("  (")[0],
                              "mt3_us": round(best[0], 2) if best else None, "mt3": _lib.describe_solution(best[1]).split("  (")[0] if best else None}), flush=True)
            del g
        del w
        torch.cuda.empty_cache()
