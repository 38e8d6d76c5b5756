// gemm_nv_bf16.hip -- streaming-kernel instances: Bf16 activations x Nv FP4 weights.
#define PETIT_TU_AT Bf16
#define PETIT_TU_FMT kFmtNv
#define PETIT_TU_TABLE solutions_nv_bf16
#define PETIT_TU_BFP_AT Bf16Bfp
#define PETIT_TU_DECODE
#include "stream_tu.inc"
