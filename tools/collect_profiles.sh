#!/bin/bash
# tools/collect_profiles.sh [tag] -- on the MI355X box: the bench line, the rocprofv3 kernel-trace/stats
# summary of the SAME command, and the HBM-traffic PMC passes (separate runs, as the microarch guide prescribes).
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
python $R/bench.py > $R/gpurun_out/bench_${TAG}.json 2> $R/gpurun_out/bench_${TAG}.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG} -o bench -- python3 $R/bench.py --no-cpu-baseline --no-cells --no-host-overhead > $R/gpurun_out/prof_${TAG}.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch_${TAG} -o p -- python3 $R/bench.py --no-cpu-baseline --no-cells --no-host-overhead --no-graph --steps 200 --warmup 200 > $R/gpurun_out/pmc_fetch_${TAG}.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write_${TAG} -o p -- python3 $R/bench.py --no-cpu-baseline --no-cells --no-host-overhead --no-graph --steps 200 --warmup 200 > $R/gpurun_out/pmc_write_${TAG}.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/pmc_sq_${TAG} -o p -- python3 $R/bench.py --no-cpu-baseline --no-cells --no-host-overhead --no-graph --steps 200 --warmup 200 > $R/gpurun_out/pmc_sq_${TAG}.log 2>&1
cd $R; cat gpurun_out/bench_${TAG}.json; head -3 gpurun_out/prof_${TAG}/bench_kernel_stats.csv | cut -c1-300
# MFMA utilisation at M = 512 (BASELINE config 5): the default (tiled dequant) kernel and the native-FP4 kernel on gate_up
cd /tmp
# (native: the FP4 x FP4 32x32x64 kernel the class table picks on gate_up: 128x256, two workgroups per CU, two k-tiles per stage, two stages ahead)
# (native8: the FP4 x FP8 kernel of round 4 on the same shape: 128x256, lean fragments, two workgroups per CU)
# (native6: the FP4 x FP6 kernel on the same shape: MXFP6 e2m3 activations, 128x256, lean fragments, two workgroups per CU)
# (nvnative8 / nvnative6, round 6: NVFP4 weights on their MFMA-native image through the class sentinels, MXFP8 / MXFP6 activations)
for v in "nv tiled" "mx native" "mx native8" "mx native6" "nv nvnative8" "nv nvnative6"; do
  set -- $v
  EXTRA=""; [ "$2" = "native" ] && EXTRA="--native --solution 124da41623301004"; [ "$2" = "native8" ] && EXTRA="--native --solution 124d541223101004"; [ "$2" = "native6" ] && EXTRA="--native --solution 124d541423501004"
  [ "$2" = "nvnative8" ] && EXTRA="--sentinel mxfp8"; [ "$2" = "nvnative6" ] && EXTRA="--sentinel mxfp6"
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_mfma_$2_${TAG} -o p -- python3 $R/tools/profile_one.py --m 512 --n 57344 --k 8192 --fmt $1 $EXTRA --iters 20 > $R/gpurun_out/pmc_mfma_$2_${TAG}.log 2>&1
done
cd $R
# afterwards, in the repo (gpurun merges gpurun_out/ back): python3 tools/pmc_to_json.py ${TAG} ${TAG}; python3 tools/pmc_mfma_to_json.py ${TAG}
