#!/bin/bash
# tools/reference_sweep.sh [tag] -- the reference benchmark's own sweep on MI355X: its 24 ENTRIES (M in {16, 256, 512} x the eight Llama-3-8B / 70B linear
# shapes) in its default dtype, fp16 x NVFP4 -> fp16 (/root/reference/tools/benchmarks/matmul.py:92-127), timed the way this repo times everything
# (HIP-graph replay, weights rotated over >= 1.28 GB -- the reference's tool reuses one weight buffer, which flatters small shapes), every enumerated
# kernel checked and timed ("-algo tune": matmul/main.cc:269-325), the library's default pick marked, and hipBLASLt's dense fp16 GEMM on the same shapes
# (the comparator of the reference README's only speed claims: 1.2-2.2x at batch < 16, "within 70 %" at large batch).
# Output: gpurun_out/refsweep_<tag>.{json,csv}; table: python tools/reference_sweep_table.py gpurun_out/refsweep_<tag>.json
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
[ "$2" = "more" ] || python $R/tools/tune.py --shapes 4096x4096,4096x14336,6144x4096,8192x8192,8192x28672,10240x8192,28672x4096,57344x8192 --ms 16,256,512 \
   --dtype f16 --fmt nv --splitk 1,2,4 --compare-dense --out $R/gpurun_out/refsweep_${TAG}.json > $R/gpurun_out/refsweep_${TAG}.log 2>&1
[ "$2" = "more" ] || tail -30 $R/gpurun_out/refsweep_${TAG}.log | cut -c1-260
# the other three families on the same entries (arch-table rows; no dense comparison): bash tools/reference_sweep.sh r03 more
if [ "$2" = "more" ]; then
  for fam in "nv bf16" "mx bf16" "mx f16"; do
    set -- $fam
    python $R/tools/tune.py --shapes 4096x4096,4096x14336,6144x4096,8192x8192,8192x28672,10240x8192,28672x4096,57344x8192 --ms 16,256,512 \
       --dtype $2 --fmt $1 --splitk 1,2,4 --out $R/gpurun_out/refsweep_${TAG}_$1_$2.json > $R/gpurun_out/refsweep_${TAG}_$1_$2.log 2>&1
    grep -c " best " $R/gpurun_out/refsweep_${TAG}_$1_$2.log
  done
fi
