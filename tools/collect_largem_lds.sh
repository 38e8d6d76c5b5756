#!/bin/bash
# tools/collect_largem_lds.sh -- round 4, VERDICT r03 item 4: is the LDS pipe what the M = 512 dequant kernels run into, and what would a
# kernel that unpacks W once per workgroup into LDS have to beat?  One rocprofv3 --pmc pass per kernel with the LDS counters next to the
# MFMA ones (gate_up, 57344 x 8192): bf16 x NVFP4 32x32x16 128x128 (the default), its 128x256 form (half the fragment reads per MFMA),
# bf16 x MXFP4 64x256, hipBLASLt's dense bf16 GEMM (256x256x64 macro tile, both operands through LDS).
# Output: gpurun_out/r04_largem/largem_lds.json
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/r04_largem
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PMC="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
run() { # tag, command...
  TAG=$1; shift
  rm -rf $OUT/lds_$TAG
  rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/lds_$TAG -o p -- "$@" > $OUT/lds_$TAG.log 2>&1
}
run nv_wide128     python3 $R/tools/profile_one.py --m 512 --n 57344 --k 8192 --fmt nv --solution 142c141113100804 --iters 20
run nv_wide128x256 python3 $R/tools/profile_one.py --m 512 --n 57344 --k 8192 --fmt nv --solution 124c141113101004 --iters 20
run mx_wide64x256  python3 $R/tools/profile_one.py --m 512 --n 57344 --k 8192 --fmt mx --solution 124c141123101002 --iters 20
run hipblaslt      python3 $R/tools/profile_hipblaslt.py --m 512 --n 57344 --k 8192 --iters 20
python3 - <<PY
import csv, glob, json, statistics
out = {}
for tag in ("nv_wide128", "nv_wide128x256", "mx_wide64x256", "hipblaslt"):
    vals, dur, name = {}, [], None
    for f in glob.glob("$OUT/lds_%s/**/*counter_collection.csv" % tag, recursive=True):
        rows = list(csv.DictReader(open(f)))
        by_kernel = {}
        for r in rows:
            if r["Counter_Name"] == "SQ_INSTS_MFMA":
                by_kernel[r["Kernel_Name"]] = max(by_kernel.get(r["Kernel_Name"], 0), float(r["Counter_Value"]))
        if not by_kernel:
            continue
        name = max(by_kernel, key=by_kernel.get)
        for r in rows:
            if r["Kernel_Name"] == name:
                vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for f in glob.glob("$OUT/lds_%s/**/*kernel_trace.csv" % tag, recursive=True):
        dur += [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if r["Kernel_Name"] == name]
    if not vals or not dur:
        print(tag, "no data")
        continue
    v = {k: statistics.median(x) for k, x in vals.items()}
    ns = statistics.median(dur)
    gui = v["GRBM_GUI_ACTIVE"] / 8                   # the counter is summed over the 8 XCDs (tools/pmc_mfma_to_json.py)
    cu_cycles = gui * 256                            # CU-cycles of the launch (SQ counters are summed over the chip)
    v.update({"kernel": name[:80], "kernel_us": ns / 1e3, "tflops": 2.0 * 512 * 57344 * 8192 / ns / 1e3,
              "lds_instr_per_mfma": v["SQ_INSTS_LDS"] / v["SQ_INSTS_MFMA"],
              "lds_idx_active_frac_of_cu_cycles": v.get("SQ_LDS_IDX_ACTIVE", 0) / cu_cycles,
              "lds_bank_conflict_frac_of_lds_cycles": v.get("SQ_LDS_BANK_CONFLICT", 0) / max(v.get("SQ_LDS_IDX_ACTIVE", 1), 1),
              "mfma_busy_frac_of_simd_cycles": v["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * 1024),
              "effective_clock_ghz": gui / ns})
    out[tag] = v
    print(tag, json.dumps({k: (round(x, 4) if isinstance(x, float) else x) for k, x in v.items()}))
json.dump(out, open("$OUT/largem_lds.json", "w"), indent=1)
PY
