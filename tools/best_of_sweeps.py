#!/usr/bin/env python3
"""tools/best_of_sweeps.py <tag> [out.tune.txt] -- profiles/<tag>_sweeps.csv.gz -> arch-table rows ($PETIT_AMD_TUNE_FILE format): per
(family, N, K, M) the fastest EXACT kernel over every sweep of the round whose id the current library still enumerates
(kernels pruned since a sweep ran are skipped; native-FP4 ids are never defaults).  Feed the result to tools/make_tuned_inc.py."""
import csv
import ctypes as C
import gzip
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "petit-kernel_amd"))
from petit_kernel import _lib  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
out = Path(sys.argv[2]) if len(sys.argv) > 2 else ROOT / "gpurun_out" / f"{tag}_best.tune.txt"
CXX = {("bf16", "nv"): (_lib.CXX_DTYPE_BF16, _lib.CXX_DTYPE_FP4_E2M1), ("f16", "nv"): (_lib.CXX_DTYPE_FP16, _lib.CXX_DTYPE_FP4_E2M1),
       ("bf16", "mx"): (_lib.CXX_DTYPE_BF16, _lib.CXX_DTYPE_MXFP4_E2M1), ("f16", "mx"): (_lib.CXX_DTYPE_FP16, _lib.CXX_DTYPE_MXFP4_E2M1)}
known = {}


def enumerable(dtype, fmt, m, n, k):
    key = (dtype, fmt, m, n, k)
    if key not in known:
        a, b = CXX[(dtype, fmt)]
        hints = _lib.SolutionHints(a, b, a, 0)
        cnt = C.c_uint(0)
        _lib.lib.petit_gemm_get_solutions(C.byref(hints), m, n, k, None, C.byref(cnt))
        ids = (C.c_uint64 * max(1, cnt.value))()
        _lib.lib.petit_gemm_get_solutions(C.byref(hints), m, n, k, ids, C.byref(cnt))
        known[key] = {int(i) & ~(0xF << 60) for i in ids[:cnt.value]}   # (compare without the K-split nibble)
    return known[key]


best = {}
with gzip.open(ROOT / "profiles" / f"{tag}_sweeps.csv.gz", "rt", newline="") as f:
    for r in csv.DictReader(f):
        if r["checked"] != "ok" or r["desc"].startswith("native"):
            continue
        sid = int(r["solution"], 16)
        m, n, k = int(r["m"]), int(r["n"]), int(r["k"])
        if (sid & ~(0xF << 60)) not in enumerable(r["dtype"], r["fmt"], m, n, k):
            continue
        key = (r["dtype"], r["fmt"], n, k, m)
        if key not in best or float(r["us_median"]) < best[key][0]:
            best[key] = (float(r["us_median"]), sid)
lines = ["# a_type b_type n k m_lo m_hi solution   (tools/best_of_sweeps.py; $PETIT_AMD_TUNE_FILE format)"]
for (dtype, fmt, n, k, m), (_, sid) in sorted(best.items()):
    a, b = CXX[(dtype, fmt)]
    lines.append(f"{a} {b} {n} {k} {m} {m} {sid:x}")
out.write_text("\n".join(lines) + "\n")
print(f"{len(best)} cells -> {out}")
