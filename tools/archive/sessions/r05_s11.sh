#!/bin/bash
# round 5, GPU session 11 (experiment): the batched-decode kernels offered to the tuner at M = 512 / 1024 on the Llama-70B shapes
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05k; mkdir -p $O
PETIT_AMD_BATCH_MAX_M=1024 timeout 1200 python tools/build_table.py --only "llama3-70b qkv tp1,llama3-70b o tp1,llama3-70b gate_up tp1,llama3-70b down tp1" --ms 512,1024 --families nv:bf16,mx:bf16 --out-dir $O/table > $O/table.log 2>&1
tail -2 $O/table.log
du -sh $O
(cd tools/probes && hipcc -O2 --offload-arch=gfx950 mfma_scale_align.hip -o mfma_scale_align && timeout 600 ./mfma_scale_align) > $O/mfma_scale_align.txt 2>&1
grep -E "RANDOM|same6" $O/mfma_scale_align.txt | head -80
