#!/usr/bin/env python3
"""tools/pmc_by_kernel.py <out.json> <tag>=<dir> [...] -- rocprofv3 --pmc output directories -> one JSON: per tag, per kernel GROUP (gemm / reduce /
quantize / other), the median of every counter over the dispatches, the median kernel duration and the dispatch count.  The groups are what a
K-split call launches: the GEMM proper and the fixed-order reduce pass (csrc/device_common.hpp splitk_reduce_kernel) are separate dispatches
with very different bottlenecks; round 3's tools/pmc_stalls.sh looked at the GEMM only."""
import csv
import glob
import json
import statistics
import sys


def group(name: str) -> str:
    if "splitk_reduce" in name:
        return "reduce"
    if "quantize" in name:
        return "quantize"
    if "gemm_" in name:
        return "gemm"
    return "other"


def main():
    out = {}
    for arg in sys.argv[2:]:
        tag, d = arg.split("=", 1)
        rec = out.setdefault(tag, {})
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            vals = {}
            for row in csv.DictReader(open(f)):
                g = group(row["Kernel_Name"])
                if g == "other":
                    continue
                vals.setdefault((g, row["Counter_Name"]), []).append(float(row["Counter_Value"]))
            for (g, c), v in vals.items():
                rec.setdefault(g, {})[c] = statistics.median(v)
                rec[g].setdefault("kernel", None)
        for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
            dur, names = {}, {}
            for r in csv.DictReader(open(f)):
                g = group(r["Kernel_Name"])
                if g == "other":
                    continue
                dur.setdefault(g, []).append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
                names[g] = r["Kernel_Name"][:160]
            for g, v in dur.items():
                rec.setdefault(g, {})
                rec[g]["kernel_ns_median_under_pmc"] = statistics.median(v)
                rec[g]["dispatches"] = len(v)
                rec[g]["kernel"] = names[g]
        for g, r in rec.items():
            wc = r.get("SQ_WAVE_CYCLES")
            if wc:
                r["frac_of_wave_cycles"] = {k: round(r[k] / wc, 4) for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
                                                                              "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS") if k in r}
            if "TCC_HIT_sum" in r and "TCC_MISS_sum" in r:
                r["l2_hit_rate"] = round(r["TCC_HIT_sum"] / max(1.0, r["TCC_HIT_sum"] + r["TCC_MISS_sum"]), 4)
            if "FETCH_SIZE" in r:   # gfx950: FETCH_SIZE is in KiB and counts 64-B units for 128-B requests -> x2 (MI355X_MICROARCH.md, HBM section)
                r["hbm_read_bytes"] = r["FETCH_SIZE"] * 1024 * 2
            if "WRITE_SIZE" in r:
                r["hbm_write_bytes"] = r["WRITE_SIZE"] * 1024
    json.dump(out, open(sys.argv[1], "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
