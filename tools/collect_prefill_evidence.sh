#!/bin/bash
# tools/collect_prefill_evidence.sh -- VERDICT r05 item 2 (a)-(c): what the vendor's bf16 GEMM and the exact-class kernels spend at a PREFILL chunk
# (M = 16375 on `o` and gate_up): kernel name (hipBLASLt's names carry the macro tile, the MFMA shape and the LDS settings), duration, VALU / LDS / VMEM
# instructions per MFMA, MFMA-pipe busy fraction, effective clock -- one rocprofv3 --pmc pass each (SQ + GRBM counters only; no trace domains besides
# --kernel-trace) -- and board power / clock / time of the same kernels back to back (tools/power_probe.py).
# Output: gpurun_out/r06_prefill/*.json
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/r06_prefill
M=${M:-16375}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PMC="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_LDS"
run() { # tag, command...
  TAG=$1; shift
  rm -rf $OUT/pmc_$TAG
  rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmc_$TAG -o p -- "$@" > $OUT/pmc_$TAG.log 2>&1
}
for SH in o gate_up; do
  if [ $SH = o ]; then N=8192; K=8192; else N=57344; K=8192; fi
  run ${SH}_nv        python3 $R/tools/profile_one.py --m $M --n $N --k $K --fmt nv --iters 6
  run ${SH}_mx        python3 $R/tools/profile_one.py --m $M --n $N --k $K --fmt mx --iters 6
  run ${SH}_hbl_first python3 $R/tools/profile_hipblaslt.py --m $M --n $N --k $K --iters 6
  run ${SH}_hbl_best  python3 $R/tools/profile_hipblaslt.py --m $M --n $N --k $K --iters 6 --algo best
  # the native class at the same M (why the MFMA pipe is 0.6 busy): MXFP4 weights raw and NVFP4 weights on their image, MXFP8 activations; the vendor's FP8 GEMM
  PETIT_AMD_NO_ROW_SPLIT=1 run ${SH}_native_mx_mxfp8 python3 $R/tools/profile_one.py --m $M --n $N --k $K --fmt mx --sentinel mxfp8 --iters 6
  PETIT_AMD_NO_ROW_SPLIT=1 run ${SH}_native_nv_mxfp8 python3 $R/tools/profile_one.py --m $M --n $N --k $K --fmt nv --sentinel mxfp8 --iters 6
  PETIT_AMD_NO_ROW_SPLIT=1 run ${SH}_native_mx_mxfp4 python3 $R/tools/profile_one.py --m $M --n $N --k $K --fmt mx --sentinel mxfp4 --iters 6
  run ${SH}_hbl_fp8   python3 $R/tools/profile_hipblaslt.py --m $M --n $N --k $K --iters 6 --fp8
done
python3 - <<PY
import csv, glob, json, statistics
out = {}
for sh in ("o", "gate_up"):
  for tag in ("nv", "mx", "hbl_first", "hbl_best", "native_mx_mxfp8", "native_nv_mxfp8", "native_mx_mxfp4", "hbl_fp8"):
    t = f"{sh}_{tag}"
    vals, dur, name = {}, [], None
    for f in glob.glob("$OUT/pmc_%s/**/*counter_collection.csv" % t, recursive=True):
        rows = list(csv.DictReader(open(f)))
        by_kernel = {}
        for r in rows:     # the GEMM is the kernel with the most MFMA instructions in total
            if r["Counter_Name"] == "SQ_INSTS_MFMA":
                by_kernel[r["Kernel_Name"]] = by_kernel.get(r["Kernel_Name"], 0) + float(r["Counter_Value"])
        if not by_kernel:
            continue
        name = max(by_kernel, key=by_kernel.get)
        for r in rows:
            if r["Kernel_Name"] == name:
                vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    if name is None:
        out[t] = {"error": open("$OUT/pmc_%s.log" % t).read()[-400:]}
        continue
    for f in glob.glob("$OUT/pmc_%s/**/*kernel_trace.csv" % t, recursive=True):
        dur += [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if r["Kernel_Name"] == name]
    v = {k: statistics.median(x) for k, x in vals.items()}
    ns = statistics.median(dur) if dur else None
    mfma = v.get("SQ_INSTS_MFMA", 0) or 1
    out[t] = {"kernel": name, "launches": len(dur), "median_ns": ns, "counters": v,
              "valu_per_mfma": v.get("SQ_INSTS_VALU", 0) / mfma, "lds_per_mfma": v.get("SQ_INSTS_LDS", 0) / mfma,
              "mfma_busy_frac_of_sq_busy": v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (v.get("SQ_BUSY_CYCLES", 0) or 1),
              "gui_active_cycles": v.get("GRBM_GUI_ACTIVE"), "effective_mhz": (v.get("GRBM_GUI_ACTIVE", 0) / ns * 1e3) if ns else None,
              "mfma_insts_per_wave": mfma / (v.get("SQ_WAVES", 0) or 1)}
json.dump({"M": $M, "note": "under --pmc the kernels are serialised and slower than in the bench; ratios of instruction counts are what this file is for", "runs": out},
          open("$OUT/prefill_pmc.json", "w"), indent=1)
for k, v in out.items():
    print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if a != "counters"})
PY
cd $R
python3 tools/power_probe.py --m $M --shapes o,gate_up --seconds 2.0 --out $OUT/power_probe_m$M.json 2>&1 | tail -24
# the group-ahead kernel and its MFMA-only skeleton (tools/ablate_wide.sh 47 must have been built): is a bare bf16 MFMA stream power-capped?
for abl in shipped 47 4; do
  if [ $abl = shipped ]; then unset PETIT_AMD_LIB; else export PETIT_AMD_LIB=$R/tools/ablate/wide/libpetit_abl_$abl.so; fi
  [ $abl = shipped ] || [ -f "$PETIT_AMD_LIB" ] || continue
  python3 tools/power_probe.py --m $M --shapes o --nv-solution 124c146113101008 --seconds 2.0 --out $OUT/power_probe_ga256_abl$abl.json 2>&1 | tail -1
done
unset PETIT_AMD_LIB
