#!/bin/bash
# tools/collect_largem_evidence.sh -- tracked evidence for the M = 512 dequant kernels against the vendor's dense GEMM:
# VALU per MFMA, MFMA-pipe busy fraction and effective clock of (a) the bf16 x NVFP4 default on gate_up, (b) its bf16 x MXFP4 twin,
# (c) hipBLASLt bf16 on a dense weight of the same shape -- one rocprofv3 --pmc pass each (SQ + GRBM counters only).
# Output: gpurun_out/r03_largem/*.json
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/r03_largem
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PMC="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_LDS"
run() { # tag, command...
  TAG=$1; shift
  rm -rf $OUT/pmc_$TAG
  rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmc_$TAG -o p -- "$@" > $OUT/pmc_$TAG.log 2>&1
}
run nv_wide128    python3 $R/tools/profile_one.py --m 512 --n 57344 --k 8192 --fmt nv --solution 142c141113100804 --iters 20
run nv_tiled128x256 python3 $R/tools/profile_one.py --m 512 --n 57344 --k 8192 --fmt nv --solution 1248141113101008 --iters 20
run mx_wide64x256 python3 $R/tools/profile_one.py --m 512 --n 57344 --k 8192 --fmt mx --solution 124c141123101002 --iters 20
run mx_wide128    python3 $R/tools/profile_one.py --m 512 --n 57344 --k 8192 --fmt mx --solution 142c141123100804 --iters 20
run hipblaslt     python3 $R/tools/profile_hipblaslt.py --m 512 --n 57344 --k 8192 --iters 20
python3 - <<PY
import csv, glob, json, statistics
out = {}
for tag in ("nv_wide128", "nv_tiled128x256", "mx_wide64x256", "mx_wide128", "hipblaslt"):
    vals, dur = {}, []
    for f in glob.glob("$OUT/pmc_%s/**/*counter_collection.csv" % tag, recursive=True):
        rows = list(csv.DictReader(open(f)))
        # the GEMM is the kernel with the most MFMA instructions
        by_kernel = {}
        for r in rows:
            if r["Counter_Name"] == "SQ_INSTS_MFMA":
                by_kernel[r["Kernel_Name"]] = max(by_kernel.get(r["Kernel_Name"], 0), float(r["Counter_Value"]))
        name = max(by_kernel, key=by_kernel.get)
        for r in rows:
            if r["Kernel_Name"] == name:
                vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for f in glob.glob("$OUT/pmc_%s/**/*kernel_trace.csv" % tag, recursive=True):
        dur += [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if r["Kernel_Name"] == name]
    v = {k: statistics.median(x) for k, x in vals.items()}
    ns = statistics.median(dur)
    v.update({"kernel": name[:80], "kernel_us": ns / 1e3, "tflops": 2.0 * 512 * 57344 * 8192 / ns / 1e3,
              "valu_per_mfma": v["SQ_INSTS_VALU"] / v["SQ_INSTS_MFMA"], "effective_clock_ghz": v["GRBM_GUI_ACTIVE"] / ns,
              "mfma_busy_frac_of_simd_cycles": v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["GRBM_GUI_ACTIVE"] * 1024)})
    out[tag] = v
    print(tag, json.dumps({k: (round(x, 3) if isinstance(x, float) else x) for k, x in v.items()}))
json.dump(out, open("$OUT/largem_pmc.json", "w"), indent=1)
PY
