#!/usr/bin/env python3
"""tools/time_ids.py --m M --n N --k K [--fmt nv|mx] [--dtype bf16|f16] [--no-check] ID [ID ...] -- time explicit kernel ids (hex; "auto" = solution_id -1) the way
bench.py times its cells (tools/benchlib.py: HIP-graph replay, rotating weights, median), one JSON line per id.  $PETIT_AMD_LIB selects the library (ablation
builds: tools/ablate_batch.sh, tools/ablate_wide.sh)."""
import argparse
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
import torch

import benchlib as BL
from petit_kernel import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, required=True)
ap.add_argument("--n", type=int, required=True)
ap.add_argument("--k", type=int, required=True)
ap.add_argument("--fmt", default="nv")
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--tag", default="")
ap.add_argument("ids", nargs="+")
a = ap.parse_args()
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(dev)
w = BL.Weights(a.fmt, a.n, a.k, 1280, dev)
g = BL.Gemm(w, a.m, torch.bfloat16 if a.dtype == "bf16" else torch.float16, dev)
for s in a.ids:
    sid = _lib.PETIT_SOLUTION_AUTO if s == "auto" else int(s, 16)
    real = g.resolve(sid)
    try:
        r = g.time(sid, stream, reps=7)
        print(json.dumps({"tag": a.tag, "lib": os.path.basename(os.environ.get("PETIT_AMD_LIB", "shipped")), "m": a.m, "n": a.n, "k": a.k, "id": f"{real:x}", "us": round(r["us"], 2),
                          "us_min": round(r["us_min"], 2), "kernel": _lib.describe_solution(real).split("  (")[0]}), flush=True)
    except Exception as exc:  # noqa: BLE001
        print(json.dumps({"tag": a.tag, "id": s, "error": str(exc)}), flush=True)
