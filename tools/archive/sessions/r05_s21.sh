#!/bin/bash
# round 5, GPU session 21: what the band raster changes in the memory system (PMC), then a SECOND measurement of the prefill rows adopted in sessions 17 / 19
# (a row stays only if it holds in this session too: tools/adopt_rows.py --confirm)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05u; mkdir -p $O
timeout 900 bash tools/collect_raster_evidence.sh > $O/raster_evidence.log 2>&1
cp gpurun_out/raster_pmc.json $O/ 2>/dev/null
tail -c 400 $O/raster_evidence.log
timeout 1500 python tools/build_table.py --ms 1024,2048,8192 --samples 3 --out-dir $O/table > $O/table.log 2>&1
tail -1 $O/table.log
timeout 600 python tools/build_table.py --families mx:f16 --ms 128,256,512 --samples 3 --out-dir $O/table_f16mx > $O/table_f16mx.log 2>&1
tail -1 $O/table_f16mx.log
gzip -f $O/*/candidates_*.csv
find gpurun_out -name "*.db" -delete 2>/dev/null
du -sh $O
