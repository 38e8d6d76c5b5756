# round 6, GPU session G: a SECOND tuner session over the NVFP4 native-class rows (the rows of session B came from one session: tools/adopt_rows.py --klass wants two),
# and the same kernels timed by the bench's method on the four Llama-70B shapes at M = 2048 (tuner vs bench: do they rank alike?)
for k in native_mxfp8 native_mxfp6 native_mxfp4; do
  python tools/build_table.py --klass $k --families nv:bf16,nv:f16 --ms 128,256,512,1024,2048,8192 --out-dir gpurun_out/r06_nvnative_confirm --samples 3 2>&1 | tail -1
done
python tools/time_cells.py --w nv --mode native_mxfp8 --m 1024 --all-kernels --out gpurun_out/r06_nvnative_allkernels_m1024.jsonl > /dev/null 2>&1
python tools/time_cells.py --w nv --mode native_mxfp8 --m 4314 --all-kernels --out gpurun_out/r06_nvnative_allkernels_m4314.jsonl > /dev/null 2>&1
wc -l gpurun_out/r06_nvnative_allkernels_m*.jsonl
