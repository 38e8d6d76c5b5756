// gemm_nv_bf16_p5.hip -- kernel instances, part 5 (native block-scaled MFMA kernels on the MFMA-native image of NVFP4 weights, nvnative.hip; stream_tu.inc): bf16 activations.
#define PETIT_TU_AT Bf16
#define PETIT_TU_FMT kFmtNv
#define PETIT_TU_TABLE solutions_nv_bf16
#define PETIT_TU_NATIVE_AT Bf16
#define PETIT_TU_NV6
#define PETIT_TU_PART 5
#include "stream_tu.inc"
