#!/bin/bash
# session 17: prefill rows re-measured under the band raster (exact class: every table shape; native classes: the Llama shapes), the held-out log again
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05q; mkdir -p $O
timeout 1500 python tools/build_table.py --ms 1024,2048,8192 --samples 3 --out-dir $O/table > $O/table.log 2>&1
tail -1 $O/table.log
for klass in native_mxfp8 native_mxfp6 native_mxfp4; do
  timeout 900 python tools/build_table.py --only llama3-70b,llama3-8b,r01-r03 --klass $klass --families mx:bf16,mx:f16 --ms 1024,2048,8192 --samples 3 --out-dir $O/$klass > $O/$klass.log 2>&1
  tail -1 $O/$klass.log
done
timeout 1500 python tools/build_table.py --part heldout --ms 1,2,4,8,16,32,64,128,256,512,1024,2048,8192 --out-dir $O/heldout > $O/heldout.log 2>&1
tail -1 $O/heldout.log
gzip -f $O/*/candidates_*.csv
du -sh $O
