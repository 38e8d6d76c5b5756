# round 6, GPU session B: the suite on the current tree (log kept), native-class table rows for NVFP4 weights on every table shape
python -m pytest tests -m gpu -q -x > gpurun_out/r06_gputest_b.log 2>&1; tail -4 gpurun_out/r06_gputest_b.log
for k in native_mxfp8 native_mxfp6 native_mxfp4; do
  python tools/build_table.py --klass $k --families nv:bf16,nv:f16 --ms 128,256,512,1024,2048,8192 --out-dir gpurun_out/r06_nvnative_all --samples 3 2>&1 | tail -1
done
