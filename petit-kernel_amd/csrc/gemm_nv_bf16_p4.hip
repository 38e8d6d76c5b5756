// gemm_nv_bf16_p4.hip -- kernel instances, part 4 (large-M kernels (tiled, 32x32x16); stream_tu.inc): bf16 activations x NVFP4 weights.
#define PETIT_TU_AT Bf16
#define PETIT_TU_FMT kFmtNv
#define PETIT_TU_TABLE solutions_nv_bf16
#define PETIT_TU_BFP_AT Bf16Bfp
#define PETIT_TU_DECODE
#define PETIT_TU_SHARED
#define PETIT_TU_PART 4
#include "stream_tu.inc"
