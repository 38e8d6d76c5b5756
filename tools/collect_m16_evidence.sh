#!/bin/bash
# tools/collect_m16_evidence.sh -- the tracked evidence for the M = 8 / 16 decode cells: PMC stall breakdown of the
# default kernels (three separate rocprofv3 --pmc passes each, tools/pmc_stalls.sh) and the compiled-out-pieces ablation of the M = 16 /
# M = 8 streaming kernel (tools/ablate).  Output: gpurun_out/r03_m16/{pmc_stall_*.json, ablate.json}; copy to profiles/ after a run.
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/r03_m16
mkdir -p $OUT
cd $R
hipcc -O3 -std=c++20 -fPIC -shared --offload-arch=gfx950 tools/ablate/ablate.hip -o tools/ablate/libablate.so || exit 1
python3 tools/ablate/run_ablate.py 16,8 2 > $OUT/ablate.log 2>&1
tail -1 $OUT/ablate.log > $OUT/ablate.json
bash tools/pmc_stalls.sh o_m16 142b411113100201 nv 16 8192 8192 > $OUT/pmc_o_m16.log 2>&1
bash tools/pmc_stalls.sh o_m8 142a411113100201 nv 8 8192 8192 > $OUT/pmc_o_m8.log 2>&1
bash tools/pmc_stalls.sh qkv_m16 124b411113100401 nv 16 10240 8192 > $OUT/pmc_qkv_m16.log 2>&1
bash tools/pmc_stalls.sh o_m1 1814811113100101 nv 1 8192 8192 > $OUT/pmc_o_m1.log 2>&1
cp gpurun_out/pmc_stall_o_m16.json gpurun_out/pmc_stall_o_m8.json gpurun_out/pmc_stall_qkv_m16.json gpurun_out/pmc_stall_o_m1.json $OUT/ 2>/dev/null
grep -h "L2\|kernel ns" $OUT/pmc_*.log > $OUT/pmc_l2_and_kernel_ns.txt
cat $OUT/ablate.json; cat $OUT/pmc_l2_and_kernel_ns.txt
