# round 6, GPU session H: the 128 x 512 "quadrant" native instances (NP = 4, one wave per SIMD, accumulators in AGPRs): parity of every native test, then every native
# kernel of the MXFP8 / MXFP4 classes timed by the bench's method at M = 1024 / 4314 / 16375 on the four Llama shapes, both weight formats
python -m pytest tests -m gpu -q -k "native" > gpurun_out/r06_gputest_h.log 2>&1; tail -4 gpurun_out/r06_gputest_h.log
for w in mx nv; do for mode in native_mxfp8 native_mxfp4; do for m in 1024 4314 16375; do
  python tools/time_cells.py --w $w --mode $mode --m $m --all-kernels --out gpurun_out/r06_quad_${w}_${mode}.jsonl > /dev/null 2>&1
done; done; done
wc -l gpurun_out/r06_quad_*.jsonl
