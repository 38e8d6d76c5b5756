// gemm_mx_f16_p4.hip -- kernel instances, part 4 (large-M kernels (tiled, 32x32x16); stream_tu.inc): fp16 activations x MXFP4 weights (a capability the reference lacks: gemm_fp4_fp16_grid.cc:55-64).  Fp16Mx (device_common.hpp): every kernel
// carries a fast body (weights straight to fp16 with the block scale in the convert, one f16 MFMA per fragment: exact while the scale bytes lie in
// 114..140, which the wave checks on the records it holds) and an exact fallback for any e8m0 scale; no caller-side promise, no side channel.
#define PETIT_TU_AT Fp16Mx
#define PETIT_TU_FMT kFmtMx
#define PETIT_TU_TABLE solutions_mx_f16
#define PETIT_TU_NATIVE_AT Fp16
#define PETIT_TU_QUANTIZE quantize32_f16
#define PETIT_TU_NO_GA // (the group-ahead 32x32x16 form has no fast / fallback pair)
#define PETIT_TU_PART 4
#include "stream_tu.inc"
