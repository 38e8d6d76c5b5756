// gemm_nv_f16_p3.hip -- kernel instances, part 3 (decode + shared-tile kernels; stream_tu.inc): fp16 activations x NVFP4 weights.
#define PETIT_TU_AT Fp16
#define PETIT_TU_FMT kFmtNv
#define PETIT_TU_TABLE solutions_nv_f16
#define PETIT_TU_DECODE
#define PETIT_TU_PART 3
#include "stream_tu.inc"
