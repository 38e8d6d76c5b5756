R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=r02
cd /tmp && export TMPDIR=/tmp
for v in "nv tiled" "mx native"; do
  set -- $v
  EXTRA=""; [ "$2" = "native" ] && EXTRA="--native --solution 124d541623301004"
  rm -rf $R/gpurun_out/pmc_mfma_$2_${TAG}
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_mfma_$2_${TAG} -o p -- python3 $R/tools/profile_one.py --m 512 --n 57344 --k 8192 --fmt $1 $EXTRA --iters 20 > $R/gpurun_out/pmc_mfma_$2_${TAG}.log 2>&1
done
