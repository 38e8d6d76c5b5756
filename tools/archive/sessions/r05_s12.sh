#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05l; mkdir -p $O
(cd tools/probes && hipcc -O2 --offload-arch=gfx950 mfma_scale_align.hip -o mfma_scale_align && timeout 600 ./mfma_scale_align) > $O/mfma_scale_align.txt 2>&1
grep -E "PAIRSUM" $O/mfma_scale_align.txt
grep -E "pairsum\(d=1\)|pairsum\(d=16\)" $O/mfma_scale_align.txt | grep "32x32 fp8" | head -40
