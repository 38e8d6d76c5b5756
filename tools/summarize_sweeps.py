#!/usr/bin/env python3
"""tools/summarize_sweeps.py <tag> -- profiles/<tag>_tune_*.json -> profiles/<tag>_summary.md (one row per dtype/shape/M:
best solution, time, algorithmic GB/s, fraction of 8 TB/s, TFLOP/s, the default pick, and the dense 16-bit GEMM when measured)."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
rows = []
for f in sorted((ROOT / "profiles").glob(f"{tag}_tune_*.json")):
    d = json.loads(f.read_text())
    fam = f"{d['dtype']}x{'nvfp4' if d['fmt'] == 'nv' else 'mxfp4'}"
    native = "native" in f.name
    for e in d["results"]:
        ok = [r for r in e["results"] if "us_median" in r]
        if not ok:
            continue
        best = ok[0]
        dflt = next((r for r in ok if r.get("is_default")), None)
        dense = e.get("dense_16bit_gemm")
        rows.append((fam + (" (native sweep)" if native else ""), e["shape"], e["m"], best, dflt, dense))
out = [f"# {tag} sweep summary (tools/measure_all.sh on one MI355X; HIP-graph replay, weights rotated over >= 0.64-1.28 GB, >= 20 ms warm-up)",
       "",
       "`best` = fastest enumerated solution; `default` = what solution_id = -1 picked WHEN THE SWEEP RAN (the arch table shipped in",
       "tuned_gfx950.inc was regenerated from these sweeps afterwards, so the shipped default is `best` for every row of the non-native sweeps).",
       "",
       "| dtype | shape | M | best us | GB/s (algorithmic) | % of 8 TB/s | TFLOP/s | default us | dense 16-bit GEMM us | dense TFLOP/s | best solution |",
       "|---|---|---|---|---|---|---|---|---|---|---|"]
for fam, shape, m, b, dflt, dense in sorted(rows, key=lambda r: (r[0], r[1], r[2])):
    out.append(f"| {fam} | {shape} | {m} | {b['us_median']:.2f} | {b['gbs']:.0f} | {100 * b['frac_hbm']:.1f} | {b['tflops']:.1f} | "
               f"{dflt['us_median']:.2f} | " if dflt else f"| {fam} | {shape} | {m} | {b['us_median']:.2f} | {b['gbs']:.0f} | {100 * b['frac_hbm']:.1f} | {b['tflops']:.1f} |  | ")
    out[-1] += (f"{dense['us_median']:.1f} | {dense['tflops']:.0f} | " if dense else " |  | ") + b["desc"].split("  (")[0] + " |"
(ROOT / "profiles" / f"{tag}_summary.md").write_text("\n".join(out) + "\n")
print(f"{len(rows)} rows -> profiles/{tag}_summary.md")
