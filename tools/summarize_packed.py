#!/usr/bin/env python3
"""tools/summarize_packed.py <tag> -- profiles/<tag>_sweeps.csv.gz (every timed candidate of every sweep of the round, packed by
tools/pack_sweeps.py) -> profiles/<tag>_summary.md: per (family, shape, M) the fastest exact kernel over all sweeps (what the
arch table ships), its time, algorithmic GB/s, fraction of 8 TB/s, TFLOP/s, and the fastest opt-in native kernel where swept."""
import csv
import gzip
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
best, native = {}, {}


def kernel_name(desc: str) -> str:
    """'<shape words>  (wg tile ...) <suffix>' -> '<shape words> <suffix>' (decode / bfp / shared-a suffixes name the kernel kind)"""
    head, _, tail = desc.partition('  (')
    return (head + ' ' + tail.partition(')')[2].strip()).strip()


with gzip.open(ROOT / "profiles" / f"{tag}_sweeps.csv.gz", "rt", newline="") as f:
    for r in csv.DictReader(f):
        if r["checked"] not in ("ok", "", "unchecked"):
            continue
        key = (f"{r['dtype']} x {'nvfp4' if r['fmt'] == 'nv' else 'mxfp4'}", r["shape"], int(r["n"]), int(r["k"]), int(r["m"]))
        tgt = native if r["desc"].startswith("native") else best
        if key not in tgt or float(r["us_median"]) < float(tgt[key]["us_median"]):
            tgt[key] = r
out = [f"# {tag} sweep summary -- best measured kernel per cell over every sweep of the round (one MI355X; tools/tune.py +",
       "# tools/benchlib.py: HIP-graph replay, weights rotated over >= 0.64-1.28 GB, >= 20 ms warm-up, median; every exact candidate's",
       "# output checked before it is timed).  The arch table (tuned_gfx950.inc) ships the `best exact` column.", "",
       "| family | shape | N | K | M | best exact us | GB/s | % of 8 TB/s | TFLOP/s | kernel | best native us | native TFLOP/s |",
       "|---|---|---|---|---|---|---|---|---|---|---|---|"]
for key in sorted(best):
    b, nv = best[key], native.get(key)
    fam, shape, n, k, m = key
    out.append(f"| {fam} | {shape} | {n} | {k} | {m} | {float(b['us_median']):.2f} | {float(b['gbs']):.0f} | {100 * float(b['frac_hbm']):.1f} | "
               f"{float(b['tflops']):.1f} | {kernel_name(b['desc'])} | " + (f"{float(nv['us_median']):.2f} | {float(nv['tflops']):.0f} |" if nv else " | |"))
(ROOT / "profiles" / f"{tag}_summary.md").write_text("\n".join(out) + "\n")
print(f"{len(best)} cells -> profiles/{tag}_summary.md")
