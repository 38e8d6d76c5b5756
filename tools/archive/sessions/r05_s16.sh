#!/bin/bash
# session 16: the band raster (tile_of_block) A/B on the prefill cells, then parity of the large-M cells under the default band
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05p; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for band in default 0 2 4 8 16; do
  if [ $band = default ]; then unset PETIT_AMD_RASTER_BAND; else export PETIT_AMD_RASTER_BAND=$band; fi
  timeout 600 python tools/raster_ab.py --ms 1024,4314,16375 --shapes o,gate_up,down --native >> $OUT/raster_ab.jsonl 2>> $OUT/raster_ab.err
done
unset PETIT_AMD_RASTER_BAND
timeout 1500 python -m pytest tests -m gpu -x -q -k "bench_cells_parity or full_size or m512" > $OUT/parity.log 2>&1
tail -3 $OUT/parity.log
