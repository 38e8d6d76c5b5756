// gemm_nv_f16_p6.hip -- kernel instances, part 6 (batched decode, 17 <= M <= 128: gemm_batch.hpp; stream_tu.inc): fp16 activations x NVFP4 weights.
#define PETIT_TU_AT Fp16
#define PETIT_TU_FMT kFmtNv
#define PETIT_TU_TABLE solutions_nv_f16
#define PETIT_TU_PART 6
#include "stream_tu.inc"
