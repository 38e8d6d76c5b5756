set -x
mkdir -p gpurun_out/pmc
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/pmc/p1 -o p -- python3 $R/tools/profile_one.py > $R/gpurun_out/pmc/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VALU --output-format csv -d $R/gpurun_out/pmc/p2 -o p -- python3 $R/tools/profile_one.py > $R/gpurun_out/pmc/p2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc/p3 -o p -- python3 $R/tools/profile_one.py > $R/gpurun_out/pmc/p3.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $R/gpurun_out/pmc/p4 -o p -- python3 $R/tools/profile_one.py > $R/gpurun_out/pmc/p4.log 2>&1
cd $R; ls -R gpurun_out/pmc | head -30; tail -3 gpurun_out/pmc/p1.log
