#!/bin/bash
# round 5, GPU session 5: the mid-M buckets of every table shape re-tuned with the batch kernels in; the alignment probe; batch parity (DA = 4 instances)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "batch_kernels" > $O/pytest_sel.log 2>&1
tail -3 $O/pytest_sel.log
(cd tools/probes && hipcc -O2 --offload-arch=gfx950 mfma_scale_align.hip -o mfma_scale_align && timeout 300 ./mfma_scale_align) > $O/mfma_scale_align.txt 2>&1
grep -E "SUMMARY|layout self" $O/mfma_scale_align.txt
timeout 2400 python tools/build_table.py --ms 32,64,128 --families nv:bf16,nv:f16,mx:bf16 --out-dir $O/table > $O/table.log 2>&1
tail -2 $O/table.log
gzip -f $O/table/candidates_table.csv
du -sh $O
