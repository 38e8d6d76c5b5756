#!/bin/bash
R=$(pwd); OUT=$R/gpurun_out/r04_shared; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SID=$1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc1 -o p -- python3 $R/tools/profile_one.py --m 512 --n 57344 --k 8192 --fmt nv --solution $SID --iters 10 > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/pmc2 -o p -- python3 $R/tools/profile_one.py --m 512 --n 57344 --k 8192 --fmt nv --solution $SID --iters 10 > $OUT/pmc2.log 2>&1
python3 - <<PY
import csv,glob,statistics
for d in ("pmc1","pmc2"):
    vals={}
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv"%d,recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm_shared" in r["Kernel_Name"]: vals.setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
    dur=[]
    for f in glob.glob("$OUT/%s/**/*kernel_trace.csv"%d,recursive=True):
        dur+=[float(r["End_Timestamp"])-float(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if "gemm_shared" in r["Kernel_Name"]]
    v={k:statistics.median(x) for k,x in vals.items()}
    print(d, "us", statistics.median(dur)/1e3 if dur else None, v)
PY
