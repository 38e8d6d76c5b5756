// gemm_mx_bf16_p5.hip -- kernel instances, part 5 (native FP4 MFMA kernels; stream_tu.inc): bf16 activations x MXFP4 weights.
#define PETIT_TU_AT Bf16
#define PETIT_TU_FMT kFmtMx
#define PETIT_TU_TABLE solutions_mx_bf16
#define PETIT_TU_NATIVE_AT Bf16
#define PETIT_TU_QUANTIZE quantize32_bf16
#define PETIT_TU_PART 5
#include "stream_tu.inc"
