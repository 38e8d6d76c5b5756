#!/bin/bash
# session 18: the row split of ragged prefill M (plan_row_split): parity, then A/B against $PETIT_AMD_NO_ROW_SPLIT=1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05r; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "auto_row_split" > $OUT/parity_split.log 2>&1
tail -3 $OUT/parity_split.log
for tag in split nosplit; do
  if [ $tag = split ]; then unset PETIT_AMD_NO_ROW_SPLIT; else export PETIT_AMD_NO_ROW_SPLIT=1; fi
  PETIT_AB_TAG=$tag timeout 900 python tools/raster_ab.py --ms 600,1100,1500,2084,2200,3000,4314 --shapes o,down,qkv,gate_up >> $OUT/row_split_ab.jsonl 2>> $OUT/row_split_ab.err
done
unset PETIT_AMD_NO_ROW_SPLIT
timeout 1500 python -m pytest tests -m gpu -x -q -k "bench_cells_parity" > $OUT/parity_cells.log 2>&1
tail -3 $OUT/parity_cells.log
