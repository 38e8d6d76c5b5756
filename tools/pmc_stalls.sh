#!/bin/bash
# tools/pmc_stalls.sh <tag> <solution-hex> [fmt] [m n k] -- stall breakdown of one kernel under rocprofv3 --pmc (two passes of
# 8 SQ counters; separate from any trace domain, as the pool requires).  <solution-hex> may be sentinel:mxfp8 | sentinel:mxfp6 | sentinel:mxfp4 (the native class's
# default pick; NVFP4 weights on their attached image); $PMC_EXTRA is appended to profile_one.py's arguments (e.g. --native for an explicit native id).
# Output: gpurun_out/pmc_stall_<tag>_{a,b}/
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=$1; SOL=$2; FMT=${3:-nv}; M=${4:-512}; N=${5:-57344}; K=${6:-8192}
case $SOL in sentinel:*) SOLARG="--sentinel ${SOL#sentinel:}";; *) SOLARG="--solution $SOL";; esac
SOLARG="$SOLARG ${PMC_EXTRA:-}"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS \
  --output-format csv -d $R/gpurun_out/pmc_stall_${TAG}_a -o p -- python3 $R/tools/profile_one.py --m $M --n $N --k $K --fmt $FMT $SOLARG --iters 12 > $R/gpurun_out/pmc_stall_${TAG}_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU GRBM_GUI_ACTIVE \
  --output-format csv -d $R/gpurun_out/pmc_stall_${TAG}_b -o p -- python3 $R/tools/profile_one.py --m $M --n $N --k $K --fmt $FMT $SOLARG --iters 12 > $R/gpurun_out/pmc_stall_${TAG}_b.log 2>&1
python3 - <<PY
import csv, statistics, glob, json
out = {}
for part in "ab":
    for f in glob.glob("$R/gpurun_out/pmc_stall_${TAG}_%s/**/*counter_collection.csv" % part, recursive=True):
        vals = {}
        for row in csv.DictReader(open(f)):
            if "gemm_" not in row["Kernel_Name"] or "reduce" in row["Kernel_Name"] or "quantize" in row["Kernel_Name"]:
                continue
            vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
        out.update({k: statistics.median(v) for k, v in vals.items()})
wc = out.get("SQ_WAVE_CYCLES", 0) or 1
out["frac_of_wave_cycles"] = {k: round(out[k] / wc, 4) for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS") if k in out}
print("${TAG}", json.dumps(out))
open("$R/gpurun_out/pmc_stall_${TAG}.json", "w").write(json.dumps(out, indent=1))
PY
# third pass: L2 behaviour of the same launches
cd /tmp
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum --output-format csv -d $R/gpurun_out/pmc_stall_${TAG}_c -o p -- python3 $R/tools/profile_one.py --m $M --n $N --k $K --fmt $FMT $SOLARG --iters 12 > $R/gpurun_out/pmc_stall_${TAG}_c.log 2>&1
python3 - <<PY
import csv, statistics, glob, json
vals = {}
for f in glob.glob("$R/gpurun_out/pmc_stall_${TAG}_c/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "gemm_" not in row["Kernel_Name"] or "reduce" in row["Kernel_Name"] or "quantize" in row["Kernel_Name"]:
            continue
        vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
print("${TAG} L2", json.dumps({k: statistics.median(v) for k, v in vals.items()}))
import glob as g
for f in g.glob("$R/gpurun_out/pmc_stall_${TAG}_a/**/*kernel_trace.csv", recursive=True):
    d = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if "gemm_" in r["Kernel_Name"] and "reduce" not in r["Kernel_Name"]]
    print("${TAG} kernel ns median", statistics.median(d), "n", len(d))
PY
