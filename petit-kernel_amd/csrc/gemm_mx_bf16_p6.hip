// gemm_mx_bf16_p6.hip -- kernel instances, part 6 (batched decode, 17 <= M <= 128: gemm_batch.hpp; stream_tu.inc): bf16 activations x MXFP4 weights.
#define PETIT_TU_AT Bf16
#define PETIT_TU_FMT kFmtMx
#define PETIT_TU_TABLE solutions_mx_bf16
#define PETIT_TU_PART 6
#include "stream_tu.inc"
