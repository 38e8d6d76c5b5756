# round 6, GPU session A: the suite on the current tree, the same-XCD combine probe, the native class's row split A/B, its fuzz
python -m pytest tests -m gpu -q -x 2>&1 | tail -6
python tools/probes/run_xcd_combine.py gpurun_out/r06_xcd_combine.json 2>&1 | grep -v amdgpu.ids | tail -30
for w in nv mx; do
  python tools/time_cells.py --w $w --mode native --m 1024,2084,4314 --shape o,down --out gpurun_out/r06_native_row_split_on.jsonl 2>&1 | grep -v amdgpu.ids | cut -c1-150
  PETIT_AMD_NO_ROW_SPLIT=1 python tools/time_cells.py --w $w --mode native --m 2084,4314 --shape o,down --out gpurun_out/r06_native_row_split_off.jsonl 2>&1 | grep -v amdgpu.ids | cut -c1-150
done
python tools/fuzz_row_split.py 3 150 native 2>&1 | tail -12
