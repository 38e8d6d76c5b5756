#!/usr/bin/env python3
"""tools/pmc_mfma_to_json.py <tag> -- gpurun_out/pmc_mfma_{tiled,native}_<tag>/p_counter_collection.csv ->
profiles/<tag>_pmc_mfma.json: per-launch medians of the MFMA counters of the M = 512 gate_up launches (default dequant
kernel; largest native kernel) and the derived MFMA-pipe utilisation."""
import csv
import glob
import json
import statistics
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
tag = sys.argv[1]
out = {"command": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE "
                  "-- python3 tools/profile_one.py --m 512 --n 57344 --k 8192 --fmt nv|mx [--native] --iters 20 (tools/collect_profiles.sh)",
       "notes": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs); kernel times under PMC are 10-40 % longer than in the "
                "sweeps and the clock lower (never compare a profiled run with an un-profiled one)"}
for kind in ("tiled", "native", "native8", "native6", "nvnative8", "nvnative6"):
    vals, dur, names = {}, [], set()
    for f in glob.glob(str(ROOT / f"gpurun_out/pmc_mfma_{kind}_{tag}/**/*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm_" not in r["Kernel_Name"] or "reduce" in r["Kernel_Name"] or "quantize" in r["Kernel_Name"]:
                continue
            vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            names.add(r["Kernel_Name"].split("<")[0])
    for f in glob.glob(str(ROOT / f"gpurun_out/pmc_mfma_{kind}_{tag}/**/*kernel_trace.csv"), recursive=True):
        dur += [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(f))
                if "gemm_" in r["Kernel_Name"] and "reduce" not in r["Kernel_Name"] and "quantize" not in r["Kernel_Name"]]
    if not vals:
        continue
    med = {k: statistics.median(v) for k, v in vals.items()}
    cyc = med.get("GRBM_GUI_ACTIVE", 0) / 8
    out[kind] = {"kernel": sorted(names), "counters_median_per_launch": med, "kernel_ns_median_under_pmc": statistics.median(dur) if dur else None,
                 "elapsed_cycles": cyc, "mfma_util": med.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (cyc * 1024) if cyc else None,
                 "valu_per_mfma": med.get("SQ_INSTS_VALU", 0) / med["SQ_INSTS_MFMA"] if med.get("SQ_INSTS_MFMA") else None,
                 "effective_clock_ghz": cyc / statistics.median(dur) if dur else None}
(ROOT / f"profiles/{tag}_pmc_mfma.json").write_text(json.dumps(out, indent=1))
print(json.dumps({k: (v if not isinstance(v, dict) else {kk: v[kk] for kk in ("kernel", "mfma_util", "valu_per_mfma", "effective_clock_ghz", "kernel_ns_median_under_pmc")}) for k, v in out.items() if k in ("tiled", "native", "native8", "native6", "nvnative8", "nvnative6")}, indent=1))
