#!/usr/bin/env python3
"""tools/check_heuristic.py -- how good is the heuristic of csrc/api.hip where the arch table has no row?

For every (dtype, shape, M) of the committed sweeps (profiles/r01_sweeps.csv.gz) ask the library for its pick with the
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

import csv

worst, rows = [], 0
cells = {}   # (dtype, fmt, shape, n, k, m) -> {solution: us}
import gzip
for f in sorted((ROOT / "profiles").glob("r*_sweeps.csv.gz")):
    if not f.name.startswith("r01"):
        continue  # the heuristic was fitted on the r01 sweeps; later rounds only add kernels the arch table selects
    for r in csv.DictReader(gzip.open(f, "rt")):
        if r["fmt"] == "dense16" or "native" in r.get("sweep", "") or "vs_dense" in r.get("sweep", ""):
            continue
        if r["dtype"] == "f16" and r["fmt"] == "mx":
            continue  # timings of the round-1 hi / lo split kernels, which no longer exist (round 4: Fp16Mx, the kernels test the scale range)
        key = (r["dtype"], r["fmt"], r["shape"], int(r["n"]), int(r["k"]), int(r["m"]))
        cells.setdefault(key, {})[int(r["solution"], 16)] = float(r["us_median"])
for (dtype, fmt, shape, n, k, m), ok in sorted(cells.items()):
    at = _lib.CXX_DTYPE_BF16 if dtype == "bf16" else _lib.CXX_DTYPE_FP16
    bt = _lib.CXX_DTYPE_FP4_E2M1 if fmt == "nv" else _lib.CXX_DTYPE_MXFP4_E2M1
    hints = _lib.SolutionHints(at, bt, at, 0)
    pick = _lib.lib.petit_gemm_default_solution(C.byref(hints), m, n, k)
    best = min(ok.values())
    rows += 1
    if pick in ok:
        worst.append((ok[pick] / best, f"{dtype}x{fmt}", shape, m, _lib.describe_solution(pick).split("  (")[0]))
    else:
        worst.append((float("nan"), f"{dtype}x{fmt}", shape, m, "not timed: " + _lib.describe_solution(pick).split("  (")[0]))
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
