// causalflow/petit/gemm.h -- the reference's C++ API, re-published for the
// MI355X build as inline wrappers over the C ABI (include/petit_amd.h).
//
// The reference keeps this API in a private header
// (lib/gemm/rocm/quantization/gemm.h:8-146 + types.h:4-13); its
// include/causalflow/petit/ directory only carries config.h.in and the TAL
// headers.  Names, namespaces, argument order and
// return conventions below are the reference's, so a C++ caller (llama.cpp
// style) only changes its include path and links libpetit_amd.so.
#pragma once

#include <stdint.h>

#include "petit_amd.h"

#if __has_include(<hip/hip_runtime_api.h>)
#include <hip/hip_runtime_api.h>
#else
typedef struct ihipStream_t *hipStream_t;
#endif

namespace causalflow::petit::rocm::quantization {

// types.h:4-13
enum DataType {
    kDataTypeInt4 = PETIT_DTYPE_INT4,
    kDataTypeFp8e4m3 = PETIT_DTYPE_FP8_E4M3,
    kDataTypeFp8e8m0 = PETIT_DTYPE_FP8_E8M0,
    kDataTypeFp4e2m1 = PETIT_DTYPE_FP4_E2M1,
    kDataTypeFp16 = PETIT_DTYPE_FP16,
    kDataTypeBf16 = PETIT_DTYPE_BF16,
    kDataTypeFp8e5m2Fnuz = PETIT_DTYPE_FP8_E5M2_FNUZ,
    kDataTypeMxFp4e2m1 = PETIT_DTYPE_MXFP4_E2M1,
};

// gemm.h:8-31
enum MatmulFeatures { kMatmulFeatures_Global = 0, kMatmulFeatures_Grid = 1, kMatmulFeatures_HighPrecision = 2 };
enum MatmulElementB { kMatmulTypeBInt4 = 0, kMatmulTypeBNvFp4 = 1, kMatmulTypeBMxFp4 = 2 };
enum MatmulMfmaType { kMatmulMfmaTypeFp16 = 0, kMatmulMfmaTypeBf16 = 1, kMatmulMfmaTypeFp8 = 2 };
enum MatmulWarpPartition { kMatmulWarpPartition_NK = 0, kMatmulWarpPartition_Cooperative = 1 };

// gemm.h:33-105.  Same 64-bit field positions; the twelve bits the reference
// pads are named here because the gfx950 kernels use them
// (petit-kernel_amd/csrc/solution.h).
struct SolutionId {
    uint64_t tile_m : 8;
    uint64_t tile_n : 8;
    uint64_t tile_k : 8;
    uint64_t features : 4;
    uint64_t element_b : 4;
    uint64_t mfma_type : 4;
    uint64_t warp_partition_m : 4;
    uint64_t warp_partition_n : 4;
    uint64_t warp_partition_k : 4;
    uint64_t warp_partition : 4;
    uint64_t n_tiles_per_wave : 4; // reference: padding
    uint64_t ring_depth : 4;       // reference: padding
    uint64_t split_k : 4;          // reference: padding

    unsigned long Repr() const {
        uint64_t r;
        __builtin_memcpy(&r, this, 8);
        return r;
    }
    static SolutionId FromRepr(unsigned long repr) {
        SolutionId s;
        uint64_t r = repr;
        __builtin_memcpy(&s, &r, 8);
        return s;
    }
    // gemm.h:68-86 -- an id from the reference's field values (tile_k in elements / 16: the reference stores tile_k / 4 of a count in units of 4).  The
    // three gfx950 fields start at their neutral values (one n-tile per wave is implied by tile_n / warp_partition_n, ring depth 0 = "as the kernel
    // table says", no K split); an id built this way names a kernel of THIS build only if petit_gemm_get_solutions lists it.
    static constexpr SolutionId MultiStage(MatmulFeatures features, MatmulElementB element_b, MatmulMfmaType mfma_type, unsigned tile_m, unsigned tile_n,
                                           unsigned tile_k, MatmulWarpPartition warp_partition, unsigned warp_partition_m, unsigned warp_partition_n,
                                           unsigned warp_partition_k) {
        SolutionId s{};
        s.tile_m = tile_m, s.tile_n = tile_n, s.tile_k = tile_k / 4, s.features = features, s.element_b = element_b, s.mfma_type = mfma_type;
        s.warp_partition_m = warp_partition_m, s.warp_partition_n = warp_partition_n, s.warp_partition_k = warp_partition_k;
        s.warp_partition = warp_partition, s.n_tiles_per_wave = 0, s.ring_depth = 0, s.split_k = 0;
        return s;
    }
    // gemm.h:88-104 -- the reference's placeholder id (16 x 64 x 8, fp16 x NVFP4, warps 1 x 2 x 2); NOT the library default: that is
    // (unsigned long)-1 = PETIT_SOLUTION_AUTO, as in the reference's callers (lib/pybind/fp4.cc:189-191)
    static constexpr SolutionId Default() {
        return MultiStage(kMatmulFeatures_Grid, kMatmulTypeBNvFp4, kMatmulMfmaTypeFp16, 1, 4, 8, kMatmulWarpPartition_NK, 1, 2, 2);
    }
};
static_assert(sizeof(SolutionId) == 8, "SolutionId must stay a 64-bit value");

static constexpr int kErrorProblemShape = PETIT_ERROR_PROBLEM_SHAPE; // gemm.h:107
static constexpr int kErrorKernelShape = PETIT_ERROR_KERNEL_SHAPE;   // gemm.h:108

// gemm.h:112-117
struct PetitSolutionHints {
    DataType a_type;
    DataType b_type;
    DataType c_type;
    bool require_high_precision;
};

namespace fp4 {

namespace detail {
inline petit_solution_hints to_c(const PetitSolutionHints &h) {
    return petit_solution_hints{(int32_t)h.a_type, (int32_t)h.b_type, (int32_t)h.c_type,
                                h.require_high_precision ? 1 : 0};
}
} // namespace detail

// gemm.h:120-124
inline int GemmFp4Fp16Grid(unsigned *c, const unsigned *a, const unsigned *b, const unsigned *scales,
                           const float *global_scale, const unsigned m, const unsigned n, const unsigned k,
                           const PetitSolutionHints &hints, unsigned long solution_id, hipStream_t stream) {
    const petit_solution_hints h = detail::to_c(hints);
    return petit_gemm_fp4_fp16_grid(c, a, b, scales, global_scale, m, n, k, &h, (uint64_t)solution_id, stream);
}

// gemm.h:126-130
inline int GemmMxFp4Fp16Grid(unsigned *c, const unsigned *a, const unsigned *b, const unsigned *scales,
                             const float *global_scale, const unsigned m, const unsigned n, const unsigned k,
                             const PetitSolutionHints &hints, unsigned long solution_id, hipStream_t stream) {
    const petit_solution_hints h = detail::to_c(hints);
    return petit_gemm_mxfp4_fp16_grid(c, a, b, scales, global_scale, m, n, k, &h, (uint64_t)solution_id, stream);
}

// gemm.h:132-133
inline int GemmGetSolutions(const PetitSolutionHints &hints, unsigned m, unsigned n, unsigned k,
                            SolutionId *sols, unsigned *n_sols) {
    const petit_solution_hints h = detail::to_c(hints);
    return petit_gemm_get_solutions(&h, m, n, k, reinterpret_cast<uint64_t *>(sols), n_sols);
}

// gemm.h:135-145 (void in the reference: errors are reported by the C ABI only)
inline void RepackNvFp4ToPetitFp4Weights(unsigned *output, const unsigned *input, unsigned in_chan,
                                         unsigned out_chan, hipStream_t stream) {
    (void)petit_repack_nvfp4_weights(output, input, in_chan, out_chan, stream);
}
inline void RepackNvFp4ToPetitFp4Scales(unsigned *out_scales, const unsigned *scales, unsigned in_chan,
                                        unsigned out_chan, hipStream_t stream) {
    (void)petit_repack_nvfp4_scales(out_scales, scales, in_chan, out_chan, stream);
}
inline void RepackMxFp4ToPetitFp4Scales(unsigned *out_scales, const unsigned *scales, unsigned in_chan,
                                        unsigned out_chan, hipStream_t stream) {
    (void)petit_repack_mxfp4_scales(out_scales, scales, in_chan, out_chan, stream);
}

} // namespace fp4
} // namespace causalflow::petit::rocm::quantization
