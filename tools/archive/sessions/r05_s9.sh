#!/bin/bash
# round 5, GPU session 9: the native classes' prefill buckets measured on the Llama shapes (their rows above M = 512 were derived from the M = 512 picks)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05i; mkdir -p $O
for klass in native_mxfp8 native_mxfp6 native_mxfp4; do
  timeout 1500 python tools/build_table.py --only llama3-70b,llama3-8b,r01-r03 --klass $klass --families mx:bf16,mx:f16 --ms 1024,2048,8192 --samples 3 --out-dir $O/$klass > $O/$klass.log 2>&1
  tail -1 $O/$klass.log
done
gzip -f $O/*/candidates_*.csv
du -sh $O
