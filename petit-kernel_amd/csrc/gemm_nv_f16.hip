// gemm_nv_f16.hip -- streaming-kernel instances: Fp16 activations x Nv FP4 weights.
#define PETIT_TU_AT Fp16
#define PETIT_TU_FMT kFmtNv
#define PETIT_TU_TABLE solutions_nv_f16
#define PETIT_TU_DECODE
#include "stream_tu.inc"
