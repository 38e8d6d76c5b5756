#!/usr/bin/env python3
"""tools/check_heuristic.py [--heldout] [--by-m] [--data FILE] -- how good is the heuristic of csrc/api.hip where the arch table has no row?

For every (dtype, format, shape, M) of the committed tuner logs (profiles/r04_table_candidates.csv.gz: every candidate kernel the in-library
tuner timed on MI355X while tools/build_table.py built the table; written by the library itself, $PETIT_AMD_TUNE_LOG) ask the library for
its pick with the table disabled ($PETIT_AMD_NO_TUNED=1, no GPU needed) and look that kernel up in the log: prints the slowdown of the
heuristic's pick against the best timed candidate.  --heldout: only the shapes that were kept OUT of the table (tools/build_table.py
HELDOUT) -- what an unseen shape gets.  The K split the heuristic may add is looked up as such (a pick the tuner did not time counts as
missing, and is listed).  Rounds 1-3 replayed the r01 sweeps here; those timed kernels that no longer exist for fp16 x MXFP4."""
import ctypes as C
import os
import statistics
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
# --mode heuristic (default): the table is off, the formula heuristic answers.  --mode nearest: the table is on and an unseen shape takes the
# kernel of the nearest tabulated shape (hal.h tuned_nearest) -- meaningful with --heldout only (a table shape would find its own row).
MODE = sys.argv[sys.argv.index("--mode") + 1] if "--mode" in sys.argv else "heuristic"
if MODE == "heuristic":
    os.environ["PETIT_AMD_NO_TUNED"] = "1"
sys.path.insert(0, str(ROOT / "petit-kernel_amd"))
sys.path.insert(0, str(ROOT / "tools"))
from petit_kernel import _lib  # noqa: E402

from build_table import HELDOUT  # noqa: E402
from table_from_candidates import read  # noqa: E402

# --data FILE [--data FILE ...]: tuner logs, oldest first; a problem (dtypes, M, N, K) timed in a later log is taken from THAT log alone (a newer
# library has kernels the older log never timed).  Default: the round-4 full log, then round 5's LAST held-out log (every M bucket incl. prefill, the
# batched-decode kernels, the band raster and the saturating output check in: session 20).
datas = [Path(sys.argv[i + 1]) for i, a in enumerate(sys.argv) if a == "--data"] or \
        [p for p in (ROOT / "profiles" / "r04_table_candidates.csv.gz", ROOT / "profiles" / "r05_heldout_final_candidates.csv.gz") if p.exists()]
best, cands = {}, {}
for data in datas:
    b, c = read(data)
    best.update(b)
    cands.update(c)
only_held = "--heldout" in sys.argv
BUCKETS = [(1, 1), (2, 2), (3, 4), (5, 8), (9, 16), (17, 32), (33, 64), (65, 128), (129, 256), (257, 512), (513, 1024), (1025, 4096), (4097, 1 << 20)]
rows, missing = [], []
for (at, bt, klass, m, n, k), (bsid, bus) in sorted(best.items()):
    if klass != 0 or ((n, k) in HELDOUT) != only_held:
        continue
    hints = _lib.SolutionHints(at, bt, at, 0)
    pick = _lib.lib.petit_gemm_default_solution(C.byref(hints), m, n, k)
    fam = f"{'bf16' if at == _lib.CXX_DTYPE_BF16 else 'f16'}x{'nv' if bt == _lib.CXX_DTYPE_FP4_E2M1 else 'mx'}"
    timed = cands[(at, bt, klass, m, n, k)]
    if pick in timed:
        rows.append((timed[pick] / bus, fam, n, k, m, _lib.describe_solution(pick).split("  (")[0], _lib.describe_solution(bsid).split("  (")[0]))
    else:
        missing.append((fam, n, k, m, _lib.describe_solution(pick).split("  (")[0]))
rows.sort()
print(f"mode {MODE}: {len(rows) + len(missing)} cases ({'held-out shapes' if only_held else 'table shapes'}), {len(rows)} picks found in the tuner's log")
if rows:
    print(f"slowdown vs best: median {statistics.median(r[0] for r in rows):.3f}, p90 {rows[int(0.9 * len(rows))][0]:.3f}, max {rows[-1][0]:.3f}")
    for r in rows[-12:]:
        print(f"  {r[0]:.2f}x  {r[1]:8s} {r[2]}x{r[3]} M={r[4]:<4d} {r[5]}   (best: {r[6]})")
if missing:
    print(f"{len(missing)} picks the tuner did not time (e.g. a kernel its output check rejected), first: {missing[0]}")
if "--by-m" in sys.argv and rows:
    print("| M bucket | cases | median | p90 | max |")
    print("|---|---|---|---|---|")
    for lo, hi in BUCKETS:
        v = sorted(r[0] for r in rows if lo <= r[4] <= hi)
        if v:
            print(f"| {lo}-{hi if hi < 1 << 20 else ''} | {len(v)} | {statistics.median(v):.2f} | {v[int(0.9 * len(v))]:.2f} | {v[-1]:.2f} |")
