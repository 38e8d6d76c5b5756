// gemm_mx_f16.hip -- streaming-kernel instances: fp16 activations x MXFP4 weights
// (exact bf16 hi/lo split of the activations; see Fp16Split in device_common.hpp).
#define PETIT_TU_AT Fp16Split
#define PETIT_TU_FMT kFmtMx
#define PETIT_TU_TABLE solutions_mx_f16
#define PETIT_TU_SPLIT 1
#define PETIT_TU_NATIVE_AT Fp16
#define PETIT_TU_QUANTIZE quantize32_f16
#include "stream_tu.inc"
