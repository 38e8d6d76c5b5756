// gemm_nv_bf16_p6.hip -- kernel instances, part 6 (batched decode, 17 <= M <= 128: gemm_batch.hpp; stream_tu.inc): bf16 activations x NVFP4 weights.
#define PETIT_TU_AT Bf16
#define PETIT_TU_FMT kFmtNv
#define PETIT_TU_TABLE solutions_nv_bf16
#define PETIT_TU_PART 6
#include "stream_tu.inc"
