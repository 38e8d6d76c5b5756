#!/bin/bash
# tools/collect_midm_evidence.sh [outdir] -- counters for the mid-M regime: `o` / qkv at M = 32 / 64 through solution_id = -1 (or
# $SOL_<tag>=<hex id>), five separate rocprofv3 --pmc passes each (SQ stall breakdown, instruction mix, L2 hit rate, fabric reads, fabric writes: the
# pool wants counters away from every trace domain but the kernel trace), GEMM and reduce dispatches reported separately (tools/pmc_by_kernel.py).
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$R/gpurun_out/r05_midm}
mkdir -p $OUT
OUT=$(cd $OUT && pwd)   # (rocprofv3 runs from /tmp)
cd /tmp && export TMPDIR=/tmp
ARGS=""
for spec in "o_m32 32 8192 8192" "o_m64 64 8192 8192" "qkv_m32 32 10240 8192" "qkv_m64 64 10240 8192" ${EXTRA_SPECS}; do   # EXTRA_SPECS: tag:M:N:K ...
  set -- ${spec//:/ }
  TAG=$1; M=$2; N=$3; K=$4
  SOLVAR=SOL_$TAG; SOL=${!SOLVAR:-auto}
  i=0
  for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
             "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU GRBM_GUI_ACTIVE" \
             "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/${TAG}_p$i -o p -- python3 $R/tools/profile_one.py --m $M --n $N --k $K --fmt ${FMT:-nv} --solution $SOL --iters 12 > $OUT/${TAG}_p$i.log 2>&1
  done
  ARGS="$ARGS $TAG=$OUT/${TAG}_p"
done
PAIRS=""
for a in $ARGS; do
  TAG=${a%%=*}; PREFIX=${a#*=}
  for d in ${PREFIX}[0-9]; do PAIRS="$PAIRS $TAG=$d"; done
done
python3 $R/tools/pmc_by_kernel.py $OUT/midm_pmc.json $PAIRS > $OUT/midm_pmc.log 2>&1
tail -c 600 $OUT/midm_pmc.log
