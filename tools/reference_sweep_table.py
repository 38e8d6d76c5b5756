#!/usr/bin/env python3
"""tools/reference_sweep_table.py <refsweep.json> [<after.json>] -- markdown table of tools/reference_sweep.sh (stdout).  <after.json>: a `tune.py --only-default`
run of the same entries made after the sweep's winners were merged into the arch table (tools/make_tuned_inc.py)."""
import json
import sys

d = json.load(open(sys.argv[1]))
after = {}
if len(sys.argv) > 2:
    after = {(c["n"], c["k"], c["m"]): c["default"]["us_median"] for c in json.load(open(sys.argv[2]))["cells"]}
print("| M | N | K | default pick us | TFLOP/s | best enumerated us | default / best | "
      + ("default after the table merge us | " if after else "") + "hipBLASLt fp16 dense us | speed-up over dense | kernels timed (refused: a K split the kernel kind has no instance for) |")
print("|---|---|---|---|---|---|---|---|---|---|" + ("---|" if after else ""))
for c in sorted(d["cells"], key=lambda c: (c["m"], c["n"] * c["k"], c["n"])):
    dflt, best, dense = c["default"], c["best"], c["dense_16bit_gemm"]
    a = after.get((c["n"], c["k"], c["m"]))
    print(f"| {c['m']} | {c['n']} | {c['k']} | {dflt['us_median']:.2f} | {dflt['tflops']:.0f} | {best['us_median']:.2f} | {dflt['us_median'] / best['us_median']:.3f} | "
          + (f"{a:.2f} | " if after else "")
          + f"{dense['us_median']:.1f} | {dense['us_median'] / dflt['us_median']:.2f}x | {c['candidates']} ({len(c['dropped'])}) |")
