// gemm_mx_f16_p6.hip -- kernel instances, part 6 (batched decode, M <= 128, loader-wave form only: gemm_batch.hpp; stream_tu.inc): fp16 activations x MXFP4
// weights.  Fp16Mx (device_common.hpp): fast body while the scale bytes a wave holds lie in 114..140, exact hi / lo fallback for any e8m0 scale otherwise.
#define PETIT_TU_AT Fp16Mx
#define PETIT_TU_FMT kFmtMx
#define PETIT_TU_TABLE solutions_mx_f16
#define PETIT_TU_BATCH_LW_ONLY
#define PETIT_TU_PART 6
#include "stream_tu.inc"
