#!/bin/bash
# tools/collect_raster_evidence.sh -- on the MI355X box: what the band raster (csrc/device_common.hpp tile_of_block) changes in the memory system at prefill
# sizes.  FETCH_SIZE / WRITE_SIZE (separate --pmc passes, as the microarch guide prescribes) and the kernel duration for the default pick of gate_up and `o`
# at M = 16375, with the raster band the library chooses and with whole columns ($PETIT_AMD_RASTER_BAND=0, rounds 2-4).  -> gpurun_out/raster_pmc.json
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/raster_pmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS=""
for shape in "gate_up 57344 8192" "o 8192 8192"; do
  set -- $shape
  for band in default 0; do
    if [ $band = default ]; then unset PETIT_AMD_RASTER_BAND; else export PETIT_AMD_RASTER_BAND=$band; fi
    for ctr in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/$1_band${band}_$ctr -o p -- python3 $R/tools/profile_one.py --m 16375 --n $2 --k $3 --fmt nv --iters 6 > $OUT/$1_band${band}_$ctr.log 2>&1
    done
    mkdir -p $OUT/$1_band${band}; cp -r $OUT/$1_band${band}_FETCH_SIZE $OUT/$1_band${band}_WRITE_SIZE $OUT/$1_band${band}/ 2>/dev/null
    ARGS="$ARGS $1_band${band}=$OUT/$1_band${band}"
  done
done
unset PETIT_AMD_RASTER_BAND
cd $R
python3 tools/pmc_by_kernel.py gpurun_out/raster_pmc.json $ARGS
find $OUT -name "*.db" -delete 2>/dev/null
cat gpurun_out/raster_pmc.json | head -60
