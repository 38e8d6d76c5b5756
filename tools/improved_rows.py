#!/usr/bin/env python3
"""tools/improved_rows.py sweep.json [min_gain] -- arch-table rows ($PETIT_AMD_TUNE_FILE format) for the cells of a tools/tune.py
sweep whose best kernel beats what solution_id = -1 ran in the SAME sweep by at least min_gain (default 3 %): run-to-run noise is
1-2 %, so a re-sweep must not shuffle rows that are level."""
import json
import sys

d = json.load(open(sys.argv[1]))
gain = float(sys.argv[2]) if len(sys.argv) > 2 else 0.03
at = 5 if d["dtype"] == "bf16" else 4
bt = 3 if d["fmt"] == "nv" else 7
for c in d["cells"]:
    best, dflt = c["best"], c["default"]
    if best and dflt and best["us_median"] < (1.0 - gain) * dflt["us_median"]:
        print(f"{at} {bt} {c['n']} {c['k']} {c['m']} {c['m']} {int(best['solution'], 16):x}   # {dflt['us_median']:.2f} -> {best['us_median']:.2f} us")
