#!/bin/bash
# tools/pmc_l2.sh <tag> <solution-hex> <fmt> <m> <n> <k> [--native] -- L2 / fabric traffic of one kernel (rocprofv3 --pmc, own pass)
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=$1; SOL=$2; FMT=$3; M=$4; N=$5; K=$6; EXTRA=$7
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum --output-format csv -d $R/gpurun_out/pmc_l2_${TAG} -o p -- python3 $R/tools/profile_one.py --m $M --n $N --k $K --fmt $FMT --solution $SOL $EXTRA --iters 12 > $R/gpurun_out/pmc_l2_${TAG}.log 2>&1
python3 - <<PY
import csv, statistics, glob, json
vals = {}
for f in glob.glob("$R/gpurun_out/pmc_l2_${TAG}/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "gemm_" not in row["Kernel_Name"] or "reduce" in row["Kernel_Name"] or "quantize" in row["Kernel_Name"]:
            continue
        vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
out = {k: statistics.median(v) for k, v in vals.items()}
if "TCC_EA0_RDREQ_sum" in out:
    out["fabric_read_MB_at_128B_per_req"] = out["TCC_EA0_RDREQ_sum"] * 128 / 1e6
print("${TAG}", json.dumps(out))
PY
tail -2 $R/gpurun_out/pmc_l2_${TAG}.log
