#!/usr/bin/env python3
"""tools/pmc_to_json.py <tag> <round-tag> -- gpurun_out/pmc_{fetch,write,sq}_<tag>/p_counter_collection.csv ->
profiles/<round-tag>_pmc_traffic.json (per-launch medians for the bench kernel, gfx950 FETCH_SIZE correction applied as
MI355X_MICROARCH.md's HBM section prescribes) and copies of the rocprofv3 --stats summary of the bench command."""
import csv
import json
import shutil
import statistics
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
tag, rtag = sys.argv[1], sys.argv[2]
ALG = 8192 * 8192 // 2 + 8192 * 8192 // 16 + 2 * 8192 + 2 * 8192 + 4


def medians(path):
    """Per-launch medians over the dispatches of the HEADLINE kernel: the gemm_decode / gemm_stream instance with the largest grid (the
    bench also launches small repack / warm-up kernels)."""
    rows = [r for r in csv.DictReader(open(path)) if "gemm_stream_kernel" in r["Kernel_Name"] or "gemm_decode_kernel" in r["Kernel_Name"]]
    if not rows:
        return {}, 0
    big = max(rows, key=lambda r: int(r.get("Grid_Size", 0) or 0))
    vals = {}
    for row in rows:
        if row["Kernel_Name"] != big["Kernel_Name"] or row.get("Grid_Size") != big.get("Grid_Size"):
            continue
        vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    return {k: statistics.median(v) for k, v in vals.items()}, max((len(v) for v in vals.values()), default=0)


fetch, n = medians(ROOT / f"gpurun_out/pmc_fetch_{tag}/p_counter_collection.csv")
write, _ = medians(ROOT / f"gpurun_out/pmc_write_{tag}/p_counter_collection.csv")
sq, _ = medians(ROOT / f"gpurun_out/pmc_sq_{tag}/p_counter_collection.csv")
rd = 2 * fetch["FETCH_SIZE"] * 1024
wr = write["WRITE_SIZE"] * 1024
out = {
    "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE|SQ_* --output-format csv -- python3 bench.py --no-cpu-baseline --no-cells --no-graph "
               "--steps 200 --warmup 200 (separate passes, tools/collect_profiles.sh)",
    "kernel": "petit_amd::gemm_decode_kernel (bench.py default solution, M=1 N=K=8192 bf16 x nvfp4)",
    "FETCH_SIZE_KB_median": fetch["FETCH_SIZE"],
    "WRITE_SIZE_KB_median": write["WRITE_SIZE"],
    "dispatches": n,
    "correction": "gfx950: FETCH_SIZE reports half the bytes of a wide coalesced streaming read (MI355X_MICROARCH.md, HBM section): "
                  "read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is uncalibrated there and is 16 KB here (the 16 KB output row)",
    "hbm_read_bytes_per_launch": rd,
    "hbm_write_bytes_per_launch": wr,
    "traffic_bytes_per_launch": rd + wr,
    "algorithmic_bytes_per_launch": ALG,
    "traffic_over_algorithmic": (rd + wr) / ALG,
    "sq_counters_median_per_dispatch": sq,
}
(ROOT / f"profiles/{rtag}_pmc_traffic.json").write_text(json.dumps(out, indent=1))
shutil.copy(ROOT / f"gpurun_out/prof_{tag}/bench_kernel_stats.csv", ROOT / f"profiles/{rtag}_bench_kernel_stats.csv")
shutil.copy(ROOT / f"gpurun_out/bench_{tag}.json", ROOT / f"profiles/{rtag}_bench.json")
print(json.dumps(out, indent=1))
