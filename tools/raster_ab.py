#!/usr/bin/env python3
"""tools/raster_ab.py [--ms 4314,16375] [--shapes o,gate_up] [--native] -- time the default picks (solution_id -1, and the native classes with --native)
of the prefill cells under the raster band the environment selects ($PETIT_AMD_RASTER_BAND: unset = launch_flags()'s choice, 0 = whole columns, n = bands of
n m-tiles; csrc/device_common.hpp tile_of_block), one JSON line per cell.  One process per band value: the library reads the variable once."""
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

SHAPES = {"qkv": (10240, 8192), "o": (8192, 8192), "gate_up": (57344, 8192), "down": (8192, 28672)}
ap = argparse.ArgumentParser()
ap.add_argument("--ms", default="4314,16375")
ap.add_argument("--shapes", default="o,gate_up")
ap.add_argument("--fmts", default="nv,mx")
ap.add_argument("--native", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(dev)
band = os.environ.get("PETIT_AB_TAG") or os.environ.get("PETIT_AMD_RASTER_BAND", "default")   # (any A/B the environment selects: $PETIT_AB_TAG names it)
for fmt in a.fmts.split(","):
    for shape in a.shapes.split(","):
        n, k = SHAPES[shape]
        w = BL.Weights(fmt, n, k, 1280, dev)
        for m in (int(x) for x in a.ms.split(",")):
            g = BL.Gemm(w, m, torch.bfloat16, dev)
            ids = [("exact", _lib.PETIT_SOLUTION_AUTO)]
            if a.native and fmt == "mx":
                ids += [("native_mxfp8", _lib.PETIT_SOLUTION_AUTO_NATIVE_MXFP8), ("native_mxfp6", _lib.PETIT_SOLUTION_AUTO_NATIVE_MXFP6),
                        ("native_mxfp4", _lib.PETIT_SOLUTION_AUTO_NATIVE_MXFP4)]
            for name, sid in ids:
                r = g.time(sid, stream, reps=5)
                print(json.dumps({"band": band, "fmt": fmt, "shape": shape, "m": m, "klass": name, "us": round(r["us"], 1), "tflops": round(r["tflops"], 1),
                                  "kernel": _lib.describe_solution(g.resolve(sid)).split("  (")[0]}), flush=True)
            del g
        del w
        torch.cuda.empty_cache()
