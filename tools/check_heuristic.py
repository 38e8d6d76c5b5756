#!/usr/bin/env python3
"""tools/check_heuristic.py -- how good is the heuristic of csrc/api.hip where the arch table has no row?

For every (dtype, shape, M) of the committed sweeps (profiles/r01_tune_*.json) ask the library for its pick with the
table disabled ($PETIT_AMD_NO_TUNED=1, no GPU needed) and look that solution up in the sweep's timings: prints the
slowdown of the heuristic pick against the best measured solution.  Shapes outside the table get this quality."""
import ctypes as C
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
os.environ["PETIT_AMD_NO_TUNED"] = "1"
sys.path.insert(0, str(ROOT / "petit-kernel_amd"))
from petit_kernel import _lib  # noqa: E402

worst, rows = [], 0
for f in sorted((ROOT / "profiles").glob("r01_tune_*.json")):
    if "native" in f.name or "vs_dense" in f.name:
        continue
    d = json.loads(f.read_text())
    at = _lib.CXX_DTYPE_BF16 if d["dtype"] == "bf16" else _lib.CXX_DTYPE_FP16
    bt = _lib.CXX_DTYPE_FP4_E2M1 if d["fmt"] == "nv" else _lib.CXX_DTYPE_MXFP4_E2M1
    hints = _lib.SolutionHints(at, bt, at, 0)
    for e in d["results"]:
        ok = {int(r["solution"], 16): r["us_median"] for r in e["results"] if "us_median" in r}
        if not ok:
            continue
        pick = _lib.lib.petit_gemm_default_solution(C.byref(hints), e["m"], e["n"], e["k"])
        best = min(ok.values())
        rows += 1
        if pick in ok:
            worst.append((ok[pick] / best, f"{d['dtype']}x{d['fmt']}", e["shape"], e["m"], _lib.describe_solution(pick).split("  (")[0]))
        else:
            worst.append((float("nan"), f"{d['dtype']}x{d['fmt']}", e["shape"], e["m"], "not timed: " + _lib.describe_solution(pick).split("  (")[0]))
timed = sorted(w for w in worst if w[0] == w[0])
print(f"{rows} cases, {len(timed)} heuristic picks found in the sweeps")
import statistics
print(f"slowdown vs best: median {statistics.median(w[0] for w in timed):.3f}, p90 {timed[int(0.9 * len(timed))][0]:.3f}, max {timed[-1][0]:.3f}")
for w in timed[-12:]:
    print(f"  {w[0]:.2f}x  {w[1]:10s} {w[2]:8s} M={w[3]:<5d} {w[4]}")
if "--by-m" in sys.argv:
    by = {}
    for w in timed:
        by.setdefault(w[3], []).append(w[0])
    for m in sorted(by):
        v = sorted(by[m])
        print(f"M={m:<5d} n={len(v):2d} median {statistics.median(v):.2f} max {v[-1]:.2f}")
