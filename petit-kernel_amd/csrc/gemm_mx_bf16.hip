// gemm_mx_bf16.hip -- streaming-kernel instances: Bf16 activations x Mx FP4 weights.
#define PETIT_TU_AT Bf16
#define PETIT_TU_FMT kFmtMx
#define PETIT_TU_TABLE solutions_mx_bf16
#define PETIT_TU_NATIVE_AT Bf16
#define PETIT_TU_QUANTIZE quantize32_bf16
#include "stream_tu.inc"
