# round 6, GPU session V: the fuzzers on the final tree (451 table rows and one kernel form later than session f's)
python tools/fuzz_parity.py 71 200 > gpurun_out/r06_fuzz.txt 2>&1; tail -1 gpurun_out/r06_fuzz.txt
python tools/fuzz_row_split.py 72 150 > gpurun_out/r06_fuzz_row_split.txt 2>&1; tail -1 gpurun_out/r06_fuzz_row_split.txt
python tools/fuzz_row_split.py 73 120 native > gpurun_out/r06_fuzz_row_split_native.txt 2>&1; tail -1 gpurun_out/r06_fuzz_row_split_native.txt
