// gemm_mx_bf16_p4.hip -- kernel instances, part 4 (large-M kernels (tiled, 32x32x16); stream_tu.inc): bf16 activations x MXFP4 weights.
#define PETIT_TU_AT Bf16
#define PETIT_TU_FMT kFmtMx
#define PETIT_TU_TABLE solutions_mx_bf16
#define PETIT_TU_NATIVE_AT Bf16
#define PETIT_TU_QUANTIZE quantize32_bf16
#define PETIT_TU_SHARED
#define PETIT_TU_PART 4
#include "stream_tu.inc"
