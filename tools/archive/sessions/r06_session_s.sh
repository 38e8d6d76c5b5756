# round 6, GPU session S: the reference's benchmark list for MXFP4 weights (bf16 activations: the reference's only MX activation type), exact + native class, against hipBLASLt bf16 -algo tune
python tools/reference_list_sweep.py --atype bf16 --btype mx --native --out gpurun_out/r06_reference_list_mx.jsonl > /dev/null 2>&1; wc -l gpurun_out/r06_reference_list_mx.jsonl
