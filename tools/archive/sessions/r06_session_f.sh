# round 6, GPU session F: the prefill evidence on the final tree (exact + native class + the vendor's kernels under the counters, power probes, the group-ahead kernel's skeleton),
# stall breakdown of the native class at M = 16375, the fuzzers, the reference's benchmark list for bf16 activations
bash tools/collect_prefill_evidence.sh > gpurun_out/r06_prefill_evidence.log 2>&1; tail -c 1500 gpurun_out/r06_prefill_evidence.log
export PETIT_AMD_NO_ROW_SPLIT=1
PMC_EXTRA="--native" bash tools/pmc_stalls.sh native8_mx_o_m16375 124d541223101004 mx 16375 8192 8192 > gpurun_out/r06_pmc_stall_native8_mx.log 2>&1
bash tools/pmc_stalls.sh native8_nv_o_m16375 sentinel:mxfp8 nv 16375 8192 8192 > gpurun_out/r06_pmc_stall_native8_nv.log 2>&1
bash tools/pmc_stalls.sh native4_mx_o_m16375 sentinel:mxfp4 mx 16375 8192 8192 > gpurun_out/r06_pmc_stall_native4_mx.log 2>&1
unset PETIT_AMD_NO_ROW_SPLIT
grep -h "^native" gpurun_out/r06_pmc_stall_native*.log | cut -c1-900
python tools/fuzz_parity.py 61 240 > gpurun_out/r06_fuzz.txt 2>&1; tail -1 gpurun_out/r06_fuzz.txt
python tools/fuzz_row_split.py 62 150 > gpurun_out/r06_fuzz_row_split.txt 2>&1; tail -1 gpurun_out/r06_fuzz_row_split.txt
python tools/fuzz_row_split.py 63 150 native > gpurun_out/r06_fuzz_row_split_native.txt 2>&1; tail -1 gpurun_out/r06_fuzz_row_split_native.txt
python tools/reference_list_sweep.py --atype bf16 --btype nv --out gpurun_out/r06_reference_list_bf16.jsonl > gpurun_out/r06_reference_list_bf16.log 2>&1; tail -1 gpurun_out/r06_reference_list_bf16.log | cut -c1-300
