// solution.h -- 64-bit solution ids and the kernel table of libpetit_amd.so.
//
// Ids keep the bit layout of the reference's SolutionId bitfield
// (lib/gemm/rocm/quantization/gemm.h:33-66) so that tools which print or
// persist ids as hex keep working; the 12 bits the reference leaves as padding
// (:45) carry the knobs that only exist in the gfx950 kernels.
//
//   bits  0- 7  tile_m            m-tiles (of 16) per workgroup        = MT
//   bits  8-15  tile_n            n-tiles (of 16) per workgroup        = WN*NT
//   bits 16-23  tile_k            k per span in units of 64            = 2*KS (low 5 bits);
//                                 bits 21-23: log2 of the direct-path activation
//                                 prefetch distance PA
//   bits 24-27  features          Grid (1) | HighPrecision (2)  -- always 3 here:
//                                 every gfx950 kernel dequantises exactly
//   bits 28-31  element_b         1 NVFP4, 2 MXFP4          (MatmulElementB)
//   bits 32-35  mfma_type         0 fp16, 1 bf16            (MatmulMfmaType)
//   bits 36-39  warp_partition_m  1; 2 = the 5 <= M <= 16 kernel whose WN waves of a K part share one
//                                 activation tile in LDS (gemm_mid.hpp; warp_partition = 10 / 11 as the
//                                 staged streaming kernels: 8 / 16 rows), and with warp_partition = 0 the same
//                                 organisation for 17 <= M <= 128 (gemm_batch.hpp: tile_m = MT m-tiles per workgroup)
//   bits 40-43  warp_partition_n  WN
//   bits 44-47  warp_partition_k  WK
//   bits 48-51  warp_partition    the reference's NK(0)/Cooperative(1) enum, never 1 in
//                                 its shipped table; here: activation path,
//                                 0 = direct L2 fragments, 1/2/3 = 1/2/4 rows staged
//                                 through wave-private LDS (AM in gemm_stream.hpp),
//                                 10/11 = 8/16 rows staged the same way;
//                                 5/6/7 = the same staged paths on the fp16 pipeline
//                                 with block-floating-point activations (bf16 x NVFP4),
//                                 9 = the native-FP4 kernel (gemm_native.hpp; mfma_type
//                                 nibble = 2, the reference's unused kMatmulMfmaTypeFp8,
//                                 bit 35 set when the activations are fp16),
//                                 8 = the tiled large-M kernel (gemm_tiled.hpp), whose
//                                 fields read: tile_m = MT, warp_partition_n = WAVES,
//                                 bits 52-55 = NTW, warp_partition_k = 1;
//                                 12 = the 32x32x16-MFMA large-M kernel (gemm_wide.hpp):
//                                 tile_m = MB (m32-blocks), bits 52-55 = 2 NP;
//                                 13 = the 32x32x64 native-FP4 kernel (gemm_native32.hpp);
//                                 4 / 14 / 15 = the NVFP4 decode kernel that scales after the
//                                 MFMA (gemm_decode.hpp), 1 / 2 / 4 activation rows (15 with warp_partition_m = 2: 8 rows)
//   bits 52-55  [was padding]     NT  n-tiles per wave
//   bits 56-59  [was padding]     D   W ring depth (tiles in flight per n-tile)
//   bits 60-63  [was padding]     split-K across workgroups (gridDim.z), >= 1
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "petit_internal.h"

namespace petit_amd {

enum : unsigned { kFeatGrid = 1u, kFeatHighPrecision = 2u };
enum : unsigned { kElemBNvFp4 = 1u, kElemBMxFp4 = 2u }; // (3 was round 3's "MXFP4, scales promised in fp16's range": read as 2 now, find_explicit in solutions.hip)
enum : unsigned { kMfmaFp16 = 0u, kMfmaBf16 = 1u, kMfmaFp8 = 2u, kMfmaFp8ActFp16 = 2u | 8u, kMfmaFp4 = 6u, kMfmaFp4ActFp16 = 6u | 8u, kMfmaFp6 = 4u, kMfmaFp6ActFp16 = 4u | 8u };

struct StreamShape {
    int ks, mt, nt, wn, wk, d, am; // am == kTiledAm marks the tiled kernel
    int pa = 1;                    // direct-path activation prefetch distance (1, 2, 4 or 8 tiles)
    int wm = 1;                    // warp_partition_m nibble: 2 = the shared-activation-tile kernel (gemm_mid.hpp)
};
constexpr unsigned pa_code(int pa) { return pa == 8 ? 3u : pa == 4 ? 2u : pa == 2 ? 1u : 0u; }
constexpr int kTiledAm = -1;
// the native-FP4 kernels (gemm_native.hpp): MXFP4 weights straight into the block-scaled MFMA,
// activations quantised to MXFP8; opt-in, never a default
constexpr int kNativeAm = -2;
// the 32x32x16-MFMA large-M kernel (gemm_wide.hpp): code 12; fields read tile_m = MB (m32-blocks), bits 52-55 = n-tiles
// per wave (2 NP), warp_partition_n = WAVES
constexpr int kWideAm = -3;
// the native-FP4 kernels on the 32x32x64 block-scaled MFMA (gemm_native32.hpp): code 13; fields as kWideAm; pa = 1: activations
// quantised to MXFP8 (mfma_type 2), pa = 2: to MXFP4 (mfma_type 6: FP4 x FP4); warp_partition_m = WM waves along M (tile_m =
// MB * WM m32-blocks per workgroup); opt-in, never a default
constexpr int kNative32Am = -4;
constexpr bool is_native_am(int am) { return am == kNativeAm || am == kNative32Am; }
// am 101 / 102 / 104: staged activations (1 / 2 / 4 rows) converted to the fp16 pipeline with
// per-span block floating point (Bf16Bfp in gemm_stream.hpp); codes 5 / 6 / 7
constexpr int kBfpAm = 100;
// am 201 / 202 / 204 / 208: the NVFP4 decode kernel with the group scale applied after the MFMA (gemm_decode.hpp), holding
// 1 / 2 / 4 / 8 activation rows; codes 4 / 14 / 15 / 15 (the 16 codes of the nibble are taken: the 8-row form is code 15 with
// warp_partition_m = 2)
constexpr int kDecodeAm = 200;
constexpr int am_rows(int am) { return am >= kDecodeAm ? am - kDecodeAm : am >= kBfpAm ? am - kBfpAm : am; }
constexpr unsigned am_code(int am) {
    return am == kNativeAm ? 9u
           : am == kNative32Am ? 13u
           : am == kWideAm  ? 12u
           : am == kTiledAm ? 8u
           : am == 0        ? 0u
           : am == 8        ? 10u
           : am == 16       ? 11u
           : am >= kDecodeAm ? (am == kDecodeAm + 1 ? 4u : am == kDecodeAm + 2 ? 14u : 15u)
                            : (am_rows(am) == 1 ? 1u : am_rows(am) == 2 ? 2u : 3u) + (am >= kBfpAm ? 4u : 0u);
}

constexpr uint64_t make_solution_id(const StreamShape &s, unsigned elem_b, unsigned mfma, unsigned splitk) {
    return (uint64_t)(s.mt & 0xff) | ((uint64_t)((s.wn * s.nt) & 0xff) << 8) |
           ((uint64_t)(((2 * s.ks) & 0x1f) | (pa_code(s.pa) << 5)) << 16) | ((uint64_t)(kFeatGrid | kFeatHighPrecision) << 24) |
           ((uint64_t)(elem_b & 0xf) << 28) | ((uint64_t)(mfma & 0xf) << 32) | ((uint64_t)(s.wm & 0xf) << 36) |
           ((uint64_t)(s.wn & 0xf) << 40) | ((uint64_t)(s.wk & 0xf) << 44) | ((uint64_t)am_code(s.am) << 48) |
           ((uint64_t)(s.nt & 0xf) << 52) |
           ((uint64_t)(s.d & 0xf) << 56) | ((uint64_t)(splitk & 0xf) << 60);
}
constexpr unsigned solution_splitk(uint64_t id) { return (unsigned)(id >> 60) & 0xf; }
constexpr uint64_t solution_without_splitk(uint64_t id) { return (id & ~((uint64_t)0xf << 60)) | ((uint64_t)1 << 60); }

using LaunchFn = int (*)(const GemmArgs &, unsigned splitk, hipStream_t);
// several GEMMs sharing the activation rows in one launch (GroupTable, petit_internal.h): the launcher fills wg_end
using LaunchGroupedFn = int (*)(GroupTable, const void *a, unsigned m, unsigned k, hipStream_t);

struct SolutionEntry {
    StreamShape shape;
    int a_type; // kDataTypeBf16 / kDataTypeFp16
    int fmt;    // kFmtNv / kFmtMx (gemm_stream.hpp)
    LaunchFn launch;
    LaunchGroupedFn launch_grouped = nullptr; // the decode and the staged streaming kernels have one (M <= 16)
};

// one table per (activation type, weight format) family, concatenated once (solutions.hip) from the parts its translation units export
// (stream_tu.inc: gemm_<family>_p<part>.hip)
const SolutionEntry *solutions_nv_bf16(int *count);
const SolutionEntry *solutions_nv_f16(int *count);
const SolutionEntry *solutions_mx_bf16(int *count);
const SolutionEntry *solutions_mx_f16(int *count); // Fp16Mx kernels (device_common.hpp)
#define PETIT_DECLARE_PARTS(fam)                      \
    const SolutionEntry *solutions_##fam##_p1(int *); \
    const SolutionEntry *solutions_##fam##_p2(int *); \
    const SolutionEntry *solutions_##fam##_p3(int *); \
    const SolutionEntry *solutions_##fam##_p4(int *);
PETIT_DECLARE_PARTS(nv_bf16)
PETIT_DECLARE_PARTS(nv_f16)
PETIT_DECLARE_PARTS(mx_bf16)
PETIT_DECLARE_PARTS(mx_f16)
#undef PETIT_DECLARE_PARTS
const SolutionEntry *solutions_nv_bf16_p6(int *); // batched decode (gemm_batch.hpp); fp16 x MXFP4: the loader-wave form only (fast body + exact fallback)
const SolutionEntry *solutions_mx_f16_p6(int *);
const SolutionEntry *solutions_nv_f16_p6(int *);
const SolutionEntry *solutions_mx_bf16_p6(int *);
const SolutionEntry *solutions_mx_bf16_p5(int *); // native FP4 MFMA kernels: MXFP4 weights only
const SolutionEntry *solutions_mx_f16_p5(int *);
const SolutionEntry *solutions_nv_bf16_p5(int *); // the same kernels on the MFMA-native image of NVFP4 weights (nvnative.hip)
const SolutionEntry *solutions_nv_f16_p5(int *);
// the activation quantiser of the 32x32x64 native kernels, stand-alone (gemm_mx_{bf16,f16}.hip): format 8 = MXFP8, 4 = MXFP4
int quantize32_bf16(const void *a, void *qa, unsigned m, unsigned k, int format, hipStream_t stream);
int quantize32_f16(const void *a, void *qa, unsigned m, unsigned k, int format, hipStream_t stream);

} // namespace petit_amd
