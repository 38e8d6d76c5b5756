// gemm_mx_f16r.hip -- fp16 activations x MXFP4 weights whose block scales are promised to lie in the fp16-safe range (e8m0 bytes 114..140:
// PETIT_DTYPE_MXFP4_E2M1_F16RANGE, include/petit_amd.h): weights convert straight to fp16 with the scale in the convert (4 VALU per word, no multiply),
// one f16 MFMA per fragment, every kernel family of the plain 16-bit path -- where gemm_mx_f16.hip (any e8m0 scale) pays two bf16 MFMAs per fragment for
// the exact hi / lo split of the activations and has the streaming and 16x16 tiled kernels only.
#define PETIT_TU_AT Fp16
#define PETIT_TU_FMT kFmtMx
#define PETIT_TU_TABLE solutions_mx_f16r
#include "stream_tu.inc"
