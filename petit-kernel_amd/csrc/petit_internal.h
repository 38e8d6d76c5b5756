// petit_internal.h -- declarations shared by the translation units of
// libpetit_amd.so.  Not installed; the public surface is include/petit_amd.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace petit_amd {

// Error codes.  0/1/2 are the reference's (quantization/gemm.h:107-108);
// the rest are additions of this library (launch failures are polled here,
// the reference never does -- SURVEY.md section 5).
enum : int {
    kOk = 0,
    kErrProblemShape = 1, // kErrorProblemShape
    kErrKernelShape = 2,  // kErrorKernelShape
    kErrLaunch = 3,       // hipGetLastError() != hipSuccess after a launch
    kErrBadArgument = 4,  // null pointer / unknown dtype
    // internal, never returned to a caller: a launcher asked for SiLU-mul in the reduce pass found that the K range does not split (fewer
    // spans than parts -> one part, no reduce pass); gemm_impl then runs the kernel unsplit with SiLU-mul in its own epilogue
    kErrSplitCollapsed = 100,
};

// Arithmetic / element types, numbered as the reference's C++ DataType
// (quantization/types.h:4-13).
enum DataType : int {
    kDataTypeInt4 = 0,
    kDataTypeFp8e4m3 = 1,
    kDataTypeFp8e8m0 = 2,
    kDataTypeFp4e2m1 = 3,
    kDataTypeFp16 = 4,
    kDataTypeBf16 = 5,
    kDataTypeFp8e5m2Fnuz = 6,
    kDataTypeMxFp4e2m1 = 7,
    // round 3's extension "MXFP4 whose every e8m0 block scale lies in 114..140" (PETIT_DTYPE_MXFP4_E2M1_F16RANGE, include/petit_amd.h): still
    // accepted, and means plain MXFP4 -- the fp16 kernels test the range themselves (Fp16Mx, device_common.hpp), so there is nothing to promise
    kDataTypeMxFp4e2m1F16Range = 8,
};

constexpr bool is_mx_type(int b_type) { return b_type == kDataTypeMxFp4e2m1 || b_type == kDataTypeMxFp4e2m1F16Range; }
constexpr int canonical_b_type(int b_type) { return b_type == kDataTypeMxFp4e2m1F16Range ? (int)kDataTypeMxFp4e2m1 : b_type; }

// the largest M any entry point accepts (PETIT_ERROR_PROBLEM_SHAPE beyond): the open-ended last bucket of the arch tables ends here, and every
// 32-bit quantity the kernels form from M (grid rows, row * k * 2 inside one workgroup's activation block) stays in range below it
constexpr unsigned kMaxM = 1u << 20;

struct GemmArgs {
    void *c;            // [m][n] 16-bit, row-major
    const void *a;      // [m][k] 16-bit, row-major
    const void *w;      // packed weights (layout.h)
    const void *s;      // packed scales  (layout.h)
    const float *gs;    // device pointer, one float
    const void *bias;   // optional fused epilogue: [n] in c's dtype, added before the single rounding; may be null
    unsigned act;       // 0 none; 1 SiLU-mul: c is [m][n/2], c[m][j] = silu(y[m][j]) * y[m][j + n/2]  (y = acc*gs + bias)
    unsigned reduce_act; // 1: SiLU-mul with a cross-workgroup K split -- the kernels see act = 0 (plain slabs), the reduce pass applies it
    float *workspace;   // fp32 split-K slabs (may be null when splitk == 1)
    unsigned m, n, k;
    unsigned spans_per_wave; // set by the launcher: ceil(spans / (split_k * WK))
    unsigned flags;          // large-M kernels: kFlagPrio | kFlagXcdRaster | band << kFlagBandShift (launch_flags(), stream_tu.inc)
    // native-FP4 pipeline (gemm_native32.hpp): activations the CALLER already holds quantised (no quantiser launch), and a
    // SiLU-mul epilogue that emits the NEXT GEMM's quantised activations instead of a 16-bit matrix
    const void *qa;          // pre-quantised activations in the k-tile-major scratch layout of format qa_format, or null
    unsigned qa_format;      // 8 (MXFP8) / 4 (MXFP4) when qa is set
    unsigned out_format;     // 0: c is a 16-bit matrix; 8 / 4: c receives [m][n/2] activations quantised to MXFP8 / MXFP4 (act = 1 only)
};
enum : unsigned { kFlagPrio = 1u, kFlagXcdRaster = 2u, kFlagBandShift = 8 }; // bits 8..15: m-tiles per raster band (0 = whole columns)

// A group of GEMMs that share the activation rows, for the grouped kernels (petit_gemm_fp4_fp16_grouped): passed by value as a
// kernel argument; wg_end[i] = workgroups of members 0..i (prefix sums along grid x).
constexpr int kMaxGroup = 8;
struct GroupTable {
    unsigned count;
    unsigned wg_end[kMaxGroup];
    unsigned n[kMaxGroup];
    const void *w[kMaxGroup];
    const void *s[kMaxGroup];
    void *c[kMaxGroup];
    const float *gs[kMaxGroup];
    const void *bias[kMaxGroup];
};

// dispatch.hip: the dispatcher behind every GEMM entry point (solution_id: explicit id or one of the AUTO sentinels)
} // namespace petit_amd
struct petit_solution_hints;
struct petit_epilogue;
namespace petit_amd {
// native pipeline options of a call (petit_gemm_mxfp4_native): a_format 8 / 4 = `a` already holds quantised activations,
// out_format 8 / 4 = the SiLU-mul epilogue emits quantised activations instead of a 16-bit matrix
struct NativeIo {
    unsigned a_format, out_format;
    const void *image = nullptr; // NVFP4 weights: their MFMA-native image (nvnative.hip), when the caller hands it over per call
};
int gemm_impl(int b_type, unsigned *c, const unsigned *a, const unsigned *b, const unsigned *scales, const float *global_scale, unsigned m,
              unsigned n, unsigned k, const petit_solution_hints *hints, uint64_t solution_id, const petit_epilogue *epilogue, void *call_ws,
              uint64_t call_ws_bytes, void *stream, const NativeIo *io = nullptr);
int tune_candidates(int a_type, int b_type, int klass, unsigned m, unsigned n, unsigned k, uint64_t max_ws, uint64_t *ids, uint64_t *needs,
                    int cap);

// tune.hip
struct TuneRequest {
    void *c;                 // [m][n] output buffer the candidates may overwrite
    const void *a;           // [m][k] activations
    const void *const *b;    // n_copies packed weight pointers
    const void *const *s;    // n_copies packed scale pointers
    unsigned n_copies;
    const float *gs;
    unsigned m, n, k;
    int a_type, b_type, klass;   // klass 0 exact, 8 / 4 native with MXFP8 / MXFP4 activations
    void *ws;
    uint64_t ws_bytes;
    bool own_workspace;      // allocate scratch for the largest candidate instead of using ws
    void *stream;
    unsigned launches, samples;
    float tolerance;
    bool persist;
    unsigned m_lo, m_hi;
    size_t rotate_bytes;     // n_copies == 1: clone the weights until the rotation covers this many bytes
};
int tune_problem(const TuneRequest &rq, uint64_t *best_solution, float *best_us);
bool autotune_enabled();
void autotune_on_first_sight(int b_type, unsigned *c, const unsigned *a, const unsigned *b, const unsigned *scales, const float *gs, unsigned m,
                             unsigned n, unsigned k, int a_type, void *ws, uint64_t ws_bytes, void *stream);

// repack.hip
int repack_weights(void *out, const void *in, unsigned k, unsigned n, hipStream_t stream);
int repack_nvscales(void *out, const void *in, unsigned k, unsigned n, hipStream_t stream);
int repack_mxscales(void *out, const void *in, unsigned k, unsigned n, hipStream_t stream);
int repack_weights_host(void *out, const void *in, unsigned k, unsigned n);
int repack_nvscales_host(void *out, const void *in, unsigned k, unsigned n);
int repack_mxscales_host(void *out, const void *in, unsigned k, unsigned n);
// reference-packed ("Petit" format of the reference wheel) -> this build's packed layout, host memory
int convert_reference_weights_host(void *out, const void *in, unsigned k, unsigned n);
int convert_reference_nvscales_host(void *out, const void *in, unsigned k, unsigned n);
int convert_reference_mxscales_host(void *out, const void *in, unsigned k, unsigned n);
// nvnative.hip: the MFMA-native image of NVFP4 weights ("petit-cdna4-nv6/1", layout.h) from the packed tensors; host twin; dense expansion (host)
int nv6_image(void *image, const void *pw, const void *ps, unsigned n, unsigned k, hipStream_t stream);
int nv6_image_host(void *image, const void *pw, const void *ps, unsigned n, unsigned k);
int nv6_image_dequant_host(float *out, const void *image, unsigned n, unsigned k);
// dequant.hip: dense expansion of packed weights (debug aid); out_kind 0 f32, 1 bf16, 2 fp16
int dequant_packed(void *out, const void *w, const void *s, float gs, unsigned n, unsigned k, int b_type, int out_kind, hipStream_t stream);

} // namespace petit_amd
