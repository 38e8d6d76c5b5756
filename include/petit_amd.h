/*
 * petit_amd.h -- C ABI of libpetit_amd.so, the MI355X (gfx950) build of the
 * petit FP4 mixed-precision GEMM.
 *
 * This is the drop-in boundary: plain pointers and sizes, no torch types, no
 * C++ in the signatures.  Every entry point names the reference interface it
 * replaces (paths relative to the reference tree, causalflow-ai/petit-kernel
 * v0.0.3).  The C++ namespace API of the reference
 * (lib/gemm/rocm/quantization/gemm.h:119-146) is provided as inline wrappers
 * over these symbols in include/causalflow/petit/gemm.h; the Python surface
 * (petit_kernel/__init__.py:8-79) binds them through ctypes.
 *
 * Ownership: the caller owns every buffer; the library keeps no pointers past
 * the call except the optional workspace registered with
 * petit_set_workspace().  All work is enqueued on `stream` (a hipStream_t);
 * nothing synchronises the host, so every call is HIP-graph capturable.
 * Thread safety: all entry points may be called concurrently from several
 * host threads and streams; kernels that need scratch memory take it per call
 * (petit_gemm_*_ws) or from the per-device registered workspace, which serves
 * ONE stream at a time (see "Scratch memory" below); petit_set_workspace()
 * itself must not race with GEMM calls on the same device.
 */
#ifndef PETIT_AMD_H_
#define PETIT_AMD_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Return codes.  0/1/2 are the reference's (quantization/gemm.h:107-108):
 * kErrorProblemShape = 1, kErrorKernelShape = 2.  3 and 4 are additions: the
 * reference never polls launch errors (gemm_fp4_fp16_grid.cuh:556-559). */
#define PETIT_OK 0
#define PETIT_ERROR_PROBLEM_SHAPE 1
#define PETIT_ERROR_KERNEL_SHAPE 2
#define PETIT_ERROR_LAUNCH 3
#define PETIT_ERROR_BAD_ARGUMENT 4

/* Element types, numbered as the reference's C++ enum
 * (quantization/types.h:4-13).  NOTE: the reference's *Python* DataType enum
 * (petit_kernel/__init__.py:8-15) is numbered differently; the Python layer
 * of this build translates. */
typedef enum petit_data_type {
    PETIT_DTYPE_INT4 = 0,
    PETIT_DTYPE_FP8_E4M3 = 1,
    PETIT_DTYPE_FP8_E8M0 = 2,
    PETIT_DTYPE_FP4_E2M1 = 3, /* NVFP4: e4m3 scales, group 16 */
    PETIT_DTYPE_FP16 = 4,
    PETIT_DTYPE_BF16 = 5,
    PETIT_DTYPE_FP8_E5M2_FNUZ = 6,
    PETIT_DTYPE_MXFP4_E2M1 = 7, /* MXFP4: e8m0 scales, group 32 */
    /* Deprecated alias of PETIT_DTYPE_MXFP4_E2M1 (round 3: "MXFP4 whose every e8m0 block scale byte lies in 114..140", a caller's promise that
     * selected a faster fp16 kernel family).  Nobody has to promise anything any more: with fp16 activations every MXFP4 kernel tests, per wave
     * and span, the scale bytes it holds anyway, converts the weights straight to fp16 (one MFMA per fragment) while they lie in
     * PETIT_MXFP4_F16RANGE_SCALE_MIN .. _MAX (2^-13 .. 2^13: e2m1 x scale a normal fp16 number -- every real checkpoint), and finishes its K range
     * in an exact bf16 hi / lo fallback body from the first byte that does not.  Both bodies are exact, so the value 8 carries no information;
     * it is accepted wherever PETIT_DTYPE_MXFP4_E2M1 is and treated as it. */
    PETIT_DTYPE_MXFP4_E2M1_F16RANGE = 8
} petit_data_type;
#define PETIT_MXFP4_F16RANGE_SCALE_MIN 114
#define PETIT_MXFP4_F16RANGE_SCALE_MAX 140

/* PetitSolutionHints, quantization/gemm.h:112-117.  Ignored when an explicit
 * solution id is passed.  require_high_precision is accepted for
 * compatibility: every gfx950 kernel here dequantises exactly, so it never
 * changes the result. */
typedef struct petit_solution_hints {
    int32_t a_type; /* PETIT_DTYPE_FP16 or PETIT_DTYPE_BF16 */
    int32_t b_type; /* PETIT_DTYPE_FP4_E2M1 or PETIT_DTYPE_MXFP4_E2M1 */
    int32_t c_type; /* must equal a_type */
    int32_t require_high_precision;
} petit_solution_hints;

/* "Let the library choose": the reference's (unsigned long)-1 sentinel,
 * fp4/gemm_fp4_fp16_grid.cc:46-48. */
#define PETIT_SOLUTION_AUTO UINT64_MAX
/* "Let the library choose INSIDE the native-FP4 class" (MXFP4 entry points; NVFP4 entry points once the weights have an MFMA-native image attached --
 * "NVFP4 weights on the native class" below; see "Native-FP4 kernels"): the caller
 * opts into quantised activations by naming the sentinel -- MXFP8 activations (FP4 x FP8 block-scaled MFMA), MXFP6 (e2m3
 * elements: the three mantissa bits of e4m3 at the instruction's FP4 rate) or MXFP4 activations (FP4 x FP4).  Needs per-call scratch (petit_gemm_workspace_bytes with the same sentinel); without it the call
 * returns PETIT_ERROR_KERNEL_SHAPE rather than silently running another accuracy class.  The Python layers spell them
 * solution_id = -2 / -3 / -4. */
#define PETIT_SOLUTION_AUTO_NATIVE_MXFP8 (UINT64_MAX - 1)
#define PETIT_SOLUTION_AUTO_NATIVE_MXFP4 (UINT64_MAX - 2)
#define PETIT_SOLUTION_AUTO_NATIVE_MXFP6 (UINT64_MAX - 3)

/*
 * c[m][n] = a[m][k] . dequant(b)[n][k]^T * (*global_scale), f32 accumulate,
 * one round-to-nearest-even to the 16-bit output type.
 *   replaces fp4::GemmFp4Fp16Grid      quantization/gemm.h:120-124
 *            (impl fp4/gemm_fp4_fp16_grid.cc:36-77)
 *   c, a          device, row-major 16-bit (hints->a_type); a is [m][k]
 *   b             device, packed weights from petit_repack_nvfp4_weights
 *   scales        device, packed scales from petit_repack_nvfp4_scales
 *   global_scale  DEVICE pointer to one float (lib/pybind/fp4.cc:198)
 *   n % 16 == 0, k % 256 == 0, m arbitrary; m|n|k == 0 returns PETIT_OK
 *   solution_id   PETIT_SOLUTION_AUTO or an id from petit_gemm_get_solutions
 */
int petit_gemm_fp4_fp16_grid(unsigned *c, const unsigned *a, const unsigned *b,
                             const unsigned *scales, const float *global_scale,
                             unsigned m, unsigned n, unsigned k,
                             const petit_solution_hints *hints,
                             uint64_t solution_id, void *stream);

/* Same for MXFP4 weights (e8m0 scales, group 32).
 *   replaces fp4::GemmMxFp4Fp16Grid    quantization/gemm.h:126-130
 *            (impl fp4/gemm_fp4_fp16_grid.cc:79-95)
 * The reference accepts bf16 activations only (gemm_fp4_fp16_grid.cc:55-64);
 * this build also accepts fp16. */
int petit_gemm_mxfp4_fp16_grid(unsigned *c, const unsigned *a, const unsigned *b,
                               const unsigned *scales, const float *global_scale,
                               unsigned m, unsigned n, unsigned k,
                               const petit_solution_hints *hints,
                               uint64_t solution_id, void *stream);

/*
 * The same two GEMMs with a fused epilogue (no counterpart in the reference,
 * whose callers add the bias in a separate torch op after the kernel has already rounded to 16 bit):
 *   c[m][n] = round16( acc[m][n] * (*global_scale) + bias[n] )
 * bias: device pointer to n elements of c's type (hints->c_type), 8-byte aligned, or NULL.
 * epilogue == NULL or {NULL, 0, 0} is exactly the plain call.
 * activation = PETIT_ACTIVATION_SILU_MUL (the gate_up projection of a gated MLP, vLLM's SiluAndMul fused
 * in): with y = acc * (*global_scale) + bias, c is [m][n/2] and
 *   c[m][j] = round16( silu(y[m][j]) * y[m][j + n/2] ),   silu(x) = x / (1 + exp(-x)).
 * Needs n % 32 == 0 and a kernel with an even number of n-tiles per wave: PETIT_SOLUTION_AUTO picks one;
 * an explicit id without that property (or with a cross-workgroup K split) returns PETIT_ERROR_KERNEL_SHAPE.
 * Any other activation value returns PETIT_ERROR_BAD_ARGUMENT.
 */
#define PETIT_ACTIVATION_NONE 0
#define PETIT_ACTIVATION_SILU_MUL 1
typedef struct petit_epilogue {
    const void *bias;
    int32_t activation;
    int32_t reserved;
} petit_epilogue;

int petit_gemm_fp4_fp16_grid_ex(unsigned *c, const unsigned *a, const unsigned *b,
                                const unsigned *scales, const float *global_scale,
                                unsigned m, unsigned n, unsigned k,
                                const petit_solution_hints *hints, uint64_t solution_id,
                                const petit_epilogue *epilogue, void *stream);
int petit_gemm_mxfp4_fp16_grid_ex(unsigned *c, const unsigned *a, const unsigned *b,
                                  const unsigned *scales, const float *global_scale,
                                  unsigned m, unsigned n, unsigned k,
                                  const petit_solution_hints *hints, uint64_t solution_id,
                                  const petit_epilogue *epilogue, void *stream);

/* Enumerate the kernels that can run (hints, m, n, k).  Count-then-fill: call
 * with sols == NULL to get *n_sols, then again with a buffer of that size.
 *   replaces fp4::GemmGetSolutions     quantization/gemm.h:132-133
 *            (impl fp4/algo_chooser.cc:14-62)
 * Returns 0, or -1 when hints->b_type is not an FP4 type (algo_chooser.cc:20-23).
 * Ids use the reference's 64-bit SolutionId bit layout (gemm.h:33-66). */
int petit_gemm_get_solutions(const petit_solution_hints *hints, unsigned m,
                             unsigned n, unsigned k, uint64_t *sols,
                             unsigned *n_sols);

/* The id the library would pick for PETIT_SOLUTION_AUTO (arch table first,
 * heuristic second); 0 when nothing fits.
 *   replaces fp4::ChooseDefaultFp4Fp16Solution  fp4/algo_chooser.cc:64-132 */
uint64_t petit_gemm_default_solution(const petit_solution_hints *hints,
                                     unsigned m, unsigned n, unsigned k);
/* The concrete id a call (hints, m, n, k, solution_id, epilogue) that hands over `workspace_bytes` of scratch would RUN:
 * solution_id may be PETIT_SOLUTION_AUTO, one of the PETIT_SOLUTION_AUTO_NATIVE_* sentinels, or an explicit id (returned
 * normalised, or 0 when that call would be refused).  petit_gemm_default_solution() answers for "as much scratch as the pick
 * wants" (what the Python layers provide): possibly an id with a K split (bits 60-63 > 1), which a caller WITHOUT scratch
 * cannot run -- such a caller (petit_gemm_fp4_fp16_grid / _ex with no registered workspace) gets the kernel this function
 * names for workspace_bytes = 0.  epilogue may be NULL. */
uint64_t petit_gemm_resolve_solution(const petit_solution_hints *hints, unsigned m, unsigned n, unsigned k, uint64_t solution_id,
                                     const petit_epilogue *epilogue, uint64_t workspace_bytes);
/* Rows are independent, so a default-pick call (PETIT_SOLUTION_AUTO, exact class) at a prefill M whose tile grid ends a little past a
 * whole number of rounds of the chip runs as TWO launches on the caller's stream: the first `rows` rows with the kernel picked for them
 * (a grid of whole rounds), the remaining m - rows rows as a default-pick problem of their own (petit-kernel_amd/csrc/pick.hip
 * plan_row_split; petit_gemm_workspace_bytes covers both).  Returns `rows`, or 0 when the call runs as one launch (always for explicit
 * ids, the native class, m <= 512, $PETIT_AMD_NO_ROW_SPLIT=1).  petit_gemm_default_solution / _resolve_solution name the kernel of
 * the problem as a whole; resolve them at (rows) and (m - rows) for the two launches.  No reference counterpart: the reference's
 * 234-kernel chooser (fp4/algo_chooser.cc:64-132) takes the grid as it comes. */
unsigned petit_gemm_auto_row_split(const petit_solution_hints *hints, unsigned m, unsigned n, unsigned k, const petit_epilogue *epilogue);
/* The same for any solution_id (round 6).  PETIT_SOLUTION_AUTO: as above.  A native-class sentinel: the rows the call runs IN THE CLASS when its grid of 128-row tiles
 * ends a little past a whole number of rounds -- the remaining few dozen rows (<= 128) then go through the EXACT default pick (a batched-decode kernel: the class has no
 * small-M kernel), i.e. they are computed exactly, never less accurately than the class promises; both parts share the call's scratch.  Only for calls that hand over 16-bit
 * activations and take a 16-bit result (no petit_native_args formats) and, for NVFP4 weights, through the entry point that has the packed tensors (the attached image).
 * 0 = one launch (always for explicit ids). */
unsigned petit_gemm_row_split(const petit_solution_hints *hints, unsigned m, unsigned n, unsigned k, uint64_t solution_id, const petit_epilogue *epilogue);
/* A test aid: the C tile (n-tile column *bn, m-tile row *bm) that workgroup `block` (= blockIdx.y * gridDim.x + blockIdx.x) of an nx x ny grid of the
 * large-M kernels computes under the XCD-aware raster with bands of `band` m-tiles (0 = whole columns) -- the very function the kernels call
 * (csrc/device_common.hpp tile_of_linear), so that its bijection is checked without a GPU.  No reference counterpart (the reference's grid is
 * blockIdx as is, gemm_fp4_fp16_grid.cuh:554-555). */
void petit_raster_tile(unsigned nx, unsigned ny, unsigned band, unsigned block, unsigned *bn, unsigned *bm);

/*
 * Offline repack of checkpoint tensors into the packed layout the GEMM reads
 * (petit-kernel_amd/csrc/layout.h).  Same byte counts in and out.
 *   in_chan = K, out_chan = N (argument order of the reference).
 *
 * petit_repack_nvfp4_weights   replaces fp4::RepackNvFp4ToPetitFp4Weights
 *     quantization/gemm.h:135-137 (impl fp4/quantization_utils.cu:729-746)
 *     input  u32 [N][K/8], nibble i of word k8 = element 8*k8+i
 *     needs  N % 16 == 0, K % 128 == 0   (lib/pybind/fp4.cc:40-43)
 *     Also used for MXFP4 weights (petit_kernel/__init__.py:27-28).
 * petit_repack_nvfp4_scales    replaces fp4::RepackNvFp4ToPetitFp4Scales
 *     quantization/gemm.h:139-141 (impl quantization_utils.cu:748-760)
 *     input  e4m3 [N][K/16];  needs N % 16 == 0, K % 256 == 0 (fp4.cc:82-92)
 * petit_repack_mxfp4_scales    replaces fp4::RepackMxFp4ToPetitFp4Scales
 *     quantization/gemm.h:143-145 (impl quantization_utils.cu:762-773)
 *     input  e8m0 [N][K/32];  needs N % 16 == 0, K % 256 == 0 (fp4.cc:126-135)
 *
 * Unlike the reference (which returns void and silently skips remainders,
 * quantization_utils.cu:734), these return PETIT_ERROR_PROBLEM_SHAPE for a
 * shape they cannot honour.
 */
int petit_repack_nvfp4_weights(unsigned *output, const unsigned *input,
                               unsigned in_chan, unsigned out_chan, void *stream);
int petit_repack_nvfp4_scales(unsigned *out_scales, const unsigned *scales,
                              unsigned in_chan, unsigned out_chan, void *stream);
int petit_repack_mxfp4_scales(unsigned *out_scales, const unsigned *scales,
                              unsigned in_chan, unsigned out_chan, void *stream);

/*
 * Dense dequantisation of PACKED weights, a debug / test aid:
 *   out[n][k] = fp4(b[n][k]) * scale[n][k / g] * global_scale     row-major [n][k]
 *   replaces DequantPetitFp4 / DequantPetitMxFp4  fp4/quantization_utils.cu:542-727 (the reference's test-only GPU
 *            dequant kernels, quantization_utils_fp4_test.cc:103-133)
 *   b, scales   packed tensors from petit_repack_*; b_type PETIT_DTYPE_FP4_E2M1 (e4m3 scales, g = 16) or
 *               PETIT_DTYPE_MXFP4_E2M1 (e8m0, g = 32); n % 16 == 0, k % 256 == 0
 *   out_type    PETIT_DTYPE_FP32 (exact for every code x scale), PETIT_DTYPE_BF16 or PETIT_DTYPE_FP16 (one RNE rounding)
 * Uses the same hardware converts and scale decode as the GEMM kernels; the GEMM never calls it.
 */
#define PETIT_DTYPE_FP32 100 /* (not in the reference's enum: only this entry point takes it) */
int petit_dequant_packed_weights(void *out, const unsigned *b, const unsigned *scales, float global_scale,
                                 unsigned n, unsigned k, int b_type, int out_type, void *stream);

/*
 * Offline twins of the three repack entry points for HOST memory: convert a checkpoint's native
 * NVFP4 / MXFP4 tensors into the packed layout on the CPU, so load-time GPU repack becomes
 * optional (the reference has no counterpart -- its repack exists only as
 * GPU kernels, quantization_utils.cu:208-304).  Same shapes, same bytes out as the device
 * versions (tests compare them bit for bit); out of place only.  These are NOT a CPU fallback of
 * the GEMM: the packed tensors are consumed by the GPU kernels alone.
 */
int petit_repack_nvfp4_weights_host(unsigned *output, const unsigned *input, unsigned in_chan, unsigned out_chan);
int petit_repack_nvfp4_scales_host(unsigned *out_scales, const unsigned *scales, unsigned in_chan, unsigned out_chan);
int petit_repack_mxfp4_scales_host(unsigned *out_scales, const unsigned *scales, unsigned in_chan, unsigned out_chan);

/*
 * Ingest of tensors that were already packed by the REFERENCE build (host memory, offline): a checkpoint repacked with
 * the reference wheel's repack_nvfp4 / process_*_scales is converted to this build's layout without going back to the
 * native tensors.  Input formats: RepackQWeightLayout64x32 + PetitFormat (quantization_utils.cu:20-87,183-253),
 * RepackScaleLayout64x32 with the e4m3 -> "e5m3" byte transform (:89-162,255-304), RepackMxScaleLayout64x32 (:165-181).
 * Same byte counts in and out, out of place only.  Weights: out_chan % 32 == 0, in_chan % 128 == 0; NV scales
 * out_chan % 64 == 0 (the reference's kernel tiles 64 x 64, :755); MX scales out_chan % 32 == 0; in_chan % 256 == 0.
 * The reference stores -0 as +0 (PetitFormat), which the conversion cannot and need not undo.
 */
int petit_convert_reference_weights_host(unsigned *output, const unsigned *input, unsigned in_chan, unsigned out_chan);
int petit_convert_reference_nvfp4_scales_host(unsigned *out_scales, const unsigned *scales, unsigned in_chan, unsigned out_chan);
int petit_convert_reference_mxfp4_scales_host(unsigned *out_scales, const unsigned *scales, unsigned in_chan, unsigned out_chan);

/*
 * Scratch memory ("workspace").  The reference API has no workspace argument (SURVEY.md section 8b "Ownership"); two
 * kinds of kernels here need device scratch: those that split K across workgroups (fp32 partial slabs, summed in a
 * fixed order by a second pass: deterministic, no float atomics) and the native-FP4 kernels (quantised activations).
 * petit_gemm_workspace_bytes() says how much a call needs; 0 for most kernels.
 *
 * Per call (preferred; the only form that is safe with several streams or concurrently running graphs):
 *   petit_gemm_*_ws(..., epilogue, workspace, workspace_bytes, stream) -- caller-owned device memory that must stay
 *   untouched until the work enqueued on `stream` by this call has finished (stream-ordered allocators give exactly
 *   that).  With PETIT_SOLUTION_AUTO a missing / too small workspace selects a kernel that needs none; with an
 *   explicit id that needs scratch it is PETIT_ERROR_KERNEL_SHAPE (none) / PETIT_ERROR_BAD_ARGUMENT (too small).
 *   epilogue may be NULL.  The Python layer allocates this per call from torch's caching allocator.
 * Registered (legacy convenience for single-stream programs): petit_set_workspace(ptr, bytes) per device; used by the
 *   entry points without a workspace argument.  One buffer cannot serve two streams at once: it binds to the first
 *   stream that uses it and calls from any other stream are refused with PETIT_ERROR_BAD_ARGUMENT (explicit ids) or
 *   fall back to a kernel without scratch (AUTO) until petit_set_workspace is called again.  Pass (NULL, 0) to
 *   unregister.  The memory stays owned by the caller.
 * Alignment: a workspace pointer (per call or registered) must be 256-byte aligned -- the kernels store f32x4 slabs
 *   and read the quantised activations with 16-byte loads at 256-byte-aligned offsets from it; a misaligned pointer is
 *   PETIT_ERROR_BAD_ARGUMENT, never a misaligned access.  (hipMalloc and torch allocations are 256 / 512-byte aligned.)
 */
int petit_gemm_fp4_fp16_grid_ws(unsigned *c, const unsigned *a, const unsigned *b,
                                const unsigned *scales, const float *global_scale,
                                unsigned m, unsigned n, unsigned k,
                                const petit_solution_hints *hints, uint64_t solution_id,
                                const petit_epilogue *epilogue, void *workspace, uint64_t workspace_bytes, void *stream);
int petit_gemm_mxfp4_fp16_grid_ws(unsigned *c, const unsigned *a, const unsigned *b,
                                  const unsigned *scales, const float *global_scale,
                                  unsigned m, unsigned n, unsigned k,
                                  const petit_solution_hints *hints, uint64_t solution_id,
                                  const petit_epilogue *epilogue, void *workspace, uint64_t workspace_bytes, void *stream);
/* Bytes of scratch the call (hints, m, n, k, solution_id) uses when it is given enough; solution_id may be
 * PETIT_SOLUTION_AUTO (the arch table may name a K-split kernel for the shape).  0: none needed. */
uint64_t petit_gemm_workspace_bytes(const petit_solution_hints *hints, unsigned m, unsigned n, unsigned k,
                                    uint64_t solution_id);
/* The same with the epilogue of the call taken into account: PETIT_SOLUTION_AUTO resolves differently under
 * PETIT_ACTIVATION_SILU_MUL (unsplit, only kernels that hold a gate / up tile pair per wave qualify; with a cross-workgroup K
 * split any kernel does -- the slabs hold the plain [m][n] product and the reduce pass applies SiLU-mul). */
uint64_t petit_gemm_workspace_bytes_ex(const petit_solution_hints *hints, unsigned m, unsigned n, unsigned k,
                                       uint64_t solution_id, const petit_epilogue *epilogue);
int petit_set_workspace(void *device_ptr, uint64_t bytes);
/* fp32-slab bytes the split-K nibble of an id implies for (m, n) (kept for round-1 callers; prefer
 * petit_gemm_workspace_bytes, which also covers the native kernels). */
uint64_t petit_workspace_bytes(uint64_t solution_id, unsigned m, unsigned n);

/*
 * Native-FP4 kernels (no counterpart in the reference): MXFP4 weights go straight into the CDNA4
 * block-scaled MFMA and the 16-bit activations are quantised on the fly to MXFP8 (e4m3 + one e8m0
 * scale per 32 k), MXFP6 (e2m3: the same three mantissa bits over three binades, at the instruction's
 * FP4 rate; mfma_type 4) or MXFP4 (e2m1; mfma_type 6: experimental accuracy).  That quantisation costs
 * ~2^-4 (e4m3, e2m3) / 2^-2 (e2m1) relative per activation, so these kernels are a
 * different accuracy class: they are NEVER chosen by PETIT_SOLUTION_AUTO and are only enumerated
 * by petit_gemm_get_solutions after petit_enable_native_fp4(1) (or $PETIT_AMD_NATIVE_FP4=1).  Their
 * ids carry mfma_type = 2 (the reference's unused kMatmulMfmaTypeFp8, gemm.h:20-24).  They need a
 * registered workspace of petit_native_workspace_bytes(m, k) bytes for the quantised activations.
 *
 * Exactness of the class, given the quantised activations (derived from the instruction, not fitted: tools/probes/mfma_scale_align.hip,
 * profiles/r05_mfma_scale_align.txt; the same for v_mfma_scale_f32_32x32x64 and 16x16x128).
 *  (1) every activation format: the partial sums of one 32-element block and the incoming accumulator are aligned to the largest of them and each
 *      is TRUNCATED to a multiple of 2^(E - 24), E = floor(log2(largest)); the sum of the aligned values is exact.  A term in another block of the
 *      same instruction survives beside +-big of any size; with FP6 / FP4 activations a block's own sum is exact.
 *  (2) FP8 (e4m3) activations only: inside a block the products are first summed in groups of 8 consecutive k, aligned to the group's largest
 *      product and truncated 14 bits below it (unit 2^(e_a + e_w - 13)).
 * With P_b / P_g the largest |a w| of block b / group g, T the sum over blocks of |block sum| (no accumulation order or K split has a larger partial
 * sum) and gs the global scale, every output satisfies
 *     |c - exact| <= gs * [ 2^-24 * (33 * sum_b max(P_b, T) + 16 * T)  +  (FP8 activations) 7 * 2^-13 * sum_g P_g ]  +  one 16-bit rounding:
 * a worst case (every truncation a full unit, all in one direction) of ~1e-4 of sum |a w| for FP6 / FP4 activations and ~5e-4 for FP8 at K = 8192;
 * typical errors are one 16-bit rounding of the result.  The tests (tests/test_gpu_parity.py native_exact_bound) and tools/fuzz_parity.py hold every
 * native kernel to it; rounds 3-4 used an empirical 1e-5 ... 4e-5 of sum |a w|, which a longer fuzz run always exceeded somewhere.
 */
int petit_enable_native_fp4(int enable);
uint64_t petit_native_workspace_bytes(unsigned m, unsigned k);
/* A process-wide opt-in for call sites that cannot name a sentinel (an unchanged serving stack calls the MXFP4 entry points with
 * PETIT_SOLUTION_AUTO): activation_format 8 / 6 / 4 makes PETIT_SOLUTION_AUTO on MXFP4 weights run the default pick of THAT native
 * class (MXFP8 / MXFP6 / MXFP4 activations) for m >= $PETIT_AMD_NATIVE_MIN_M (default 64), whenever the call has the scratch the class
 * needs (petit_gemm_workspace_bytes(.., PETIT_SOLUTION_AUTO) then reports it; both Python layers pass it per call) -- without scratch the
 * exact default runs, as before.  0 switches it off (the default).  Initial value: $PETIT_AMD_MXFP4_ACTIVATIONS = mxfp8 | mxfp6 | mxfp4.
 * NVFP4 weights, explicit ids and the sentinels are not affected. */
int petit_set_mxfp4_default_class(int activation_format);
int petit_get_mxfp4_default_class(void);

/*
 * The native class as a PIPELINE (no counterpart in the reference).  petit_gemm_mxfp4_fp16_grid_ws with a native id runs two
 * launches per GEMM (activation quantiser, then the block-scaled-MFMA kernel).  Two hand-over points remove the quantiser:
 *   a_format   8 / 6 / 4: `a` is not the 16-bit matrix but activations ALREADY quantised to MXFP8 / MXFP6 / MXFP4 for (m, k), produced by
 *              petit_quantize_activations() (once, for any number of GEMMs that share the input: q / k / v, gate / up) or by
 *              a producer GEMM's epilogue (next item).  The bytes are opaque ("petit-qact/1": k-tile-major, the 32x32x64
 *              kernels' operand order); petit_quantized_activation_bytes() sizes them.  0: `a` is the 16-bit [m][k] matrix.
 *   out_format 8 / 6 / 4, with epilogue->activation = PETIT_ACTIVATION_SILU_MUL: `c` receives silu(y_gate) * y_up QUANTISED for the
 *              next GEMM (m, k' = n / 2) -- petit_quantized_activation_bytes(m, n / 2, out_format) bytes -- instead of the
 *              16-bit [m][n/2] matrix: gate_up -> SiLU-mul -> down of a gated MLP in two launches.  Needs n % 512 == 0 and a
 *              kernel with 128 x 256 workgroup tiles (the sentinels pick one).  Quantised from the f32 result with the
 *              quantiser's own rule (E8M0 scale from the block maximum of 32 columns).
 * solution_id: PETIT_SOLUTION_AUTO_NATIVE_MXFP8 / _MXFP6 / _MXFP4 (must match a_format when given) or an explicit native kernel id;
 * with a_format or out_format set only the 32x32x64 kernels qualify (PETIT_ERROR_KERNEL_SHAPE otherwise).  hints->a_type
 * names the 16-bit type of the matrix input / output and of the bias.  workspace: what petit_gemm_native_workspace_bytes()
 * says for the same arguments (with a_format set: only the slabs of a K split; often 0).
 */
typedef struct petit_native_args {
    uint32_t struct_bytes; /* sizeof(petit_native_args) */
    int32_t a_format;      /* 0, 8 (MXFP8), 6 (MXFP6 e2m3) or 4 (MXFP4) */
    int32_t out_format;    /* 0, 8, 6 or 4 */
    int32_t reserved;      /* 0 */
} petit_native_args;
int petit_gemm_mxfp4_native(void *c, const void *a, const unsigned *b, const unsigned *scales, const float *global_scale, unsigned m,
                            unsigned n, unsigned k, const petit_solution_hints *hints, uint64_t solution_id, const petit_epilogue *epilogue,
                            const petit_native_args *native, void *workspace, uint64_t workspace_bytes, void *stream);
uint64_t petit_gemm_native_workspace_bytes(const petit_solution_hints *hints, unsigned m, unsigned n, unsigned k, uint64_t solution_id,
                                           const petit_epilogue *epilogue, const petit_native_args *native);
/* 16-bit activations [m][k] (a_type PETIT_DTYPE_BF16 / _FP16) -> "petit-qact/1" bytes of `format` (8 / 4) in qa; k % 256 == 0,
 * both pointers 16-byte aligned. */
uint64_t petit_quantized_activation_bytes(unsigned m, unsigned k, int format);
int petit_quantize_activations(void *qa, const void *a, unsigned m, unsigned k, int a_type, int format, void *stream);

/*
 * NVFP4 weights on the native class (no counterpart in the reference; BASELINE north_star: "a native fp4/fp8 MFMA variant").
 *
 * NVFP4's e4m3 group-16 scales do not fit the block-scaled MFMA (one E8M0 scale per 32 k), and multiplying the e4m3 mantissa into the elements
 * inside the GEMM costs what the exact kernels' unpack costs (petit-kernel_amd/csrc/nvnative.hip has the arithmetic).  So the weights are
 * re-encoded ONCE, at load time, into an MFMA-native image ("petit-cdna4-nv6/1", csrc/layout.h): per 32-k block of a weight row one E8M0 scale
 * 2^E, E = floor(log2(max |fp4 x e4m3|)) - 2, and FP6 e2m3 elements RNE(fp4 x e4m3 / 2^E) -- 6.25 bits per weight next to the 4.5 of the packed
 * tensors, which the exact kernels (every decode call) keep reading.  The instruction then runs at the rate of the ACTIVATION format: MXFP6 / MXFP4
 * activations at the FP4 rate, MXFP8 at the FP8 rate; global_scale stays in the epilogue.
 *
 * Accuracy class: the native class's (quantised activations, see above) PLUS the re-rounding of the weights: fp4 x e4m3 has up to 6 significant
 * bits, e2m3 keeps 4, and the group with the smaller scale of a block loses one more bit per binade of distance.  Per element
 *     |w_image - w_nvfp4| <= 2^-4 |w_nvfp4|  (elements >= 2^E, i.e. within 3 binades of the block maximum),   <= 2^(E-4)  below,
 * hence per output |c_image - c_nvfp4| <= gs * sum_k |a_k| * max(2^-4 |w_k|, 2^(E_k - 4)).  On weights quantised by the checkpoint recipe
 * (tools/quantize_weights.py) 22-38 % of the elements move, by 2.3 % rms of the weight: the weight's total quantisation error goes from 9.51 % to
 * 9.79 % of its rms (profiles/r06_nv6_reencode.md; MLP / stacked budgets: profiles/r06_*accuracy_budget*.json).  Given the image and the
 * quantised activations the kernels are exact in the sense of "Exactness of the class" above (same instruction, same bound).
 *
 *   petit_nvfp4_native_image_bytes(in_chan = K, out_chan = N)    bytes of the image (0 for a shape the class does not take: N % 16, K % 256)
 *   petit_nvfp4_native_image(image, b, scales, K, N, stream)     device: from the PACKED tensors of petit_repack_nvfp4_weights / _scales;
 *                                                                image 256-byte aligned; N * K * 3 / 4 < 2^32
 *   petit_nvfp4_native_image_host                                host twin, bit-identical (offline conversion; packed host tensors from
 *                                                                petit_repack_nvfp4_*_host)
 *   petit_nvfp4_native_image_dequant_host(out, image, K, N)      test / debug aid: out[n][k] f32 = element x 2^(scale - 127), no global scale
 *
 * Running it -- two ways, same kernels:
 *   petit_gemm_nvfp4_native(c, a, image, ...)   names the image per call; solution_id = PETIT_SOLUTION_AUTO_NATIVE_MXFP8 / _MXFP6 / _MXFP4 (the
 *       activation format) or an explicit native id of the NVFP4 family; native / workspace exactly as petit_gemm_mxfp4_native (pre-quantised
 *       activations, the quantising SiLU-mul epilogue, petit_gemm_native_workspace_bytes with hints->b_type = PETIT_DTYPE_FP4_E2M1).
 *   petit_nvfp4_native_attach(b, image)         for call sites that keep calling the reference's entry point: afterwards
 *       petit_gemm_fp4_fp16_grid_ws(c, a, b, scales, ..., PETIT_SOLUTION_AUTO_NATIVE_*, ...) -- mul_nvfp4_a16(..., solution_id = -2 / -3 / -4) --
 *       runs on the image attached to `b`.  The image stays the caller's memory and must outlive the attachment; image = NULL detaches.  A
 *       sentinel (or explicit native id) on weights without an image returns PETIT_ERROR_KERNEL_SHAPE -- never another accuracy class.
 * PETIT_SOLUTION_AUTO on NVFP4 weights is never affected: it stays the exact class.
 */
uint64_t petit_nvfp4_native_image_bytes(unsigned in_chan, unsigned out_chan);
int petit_nvfp4_native_image(void *image, const unsigned *b, const unsigned *scales, unsigned in_chan, unsigned out_chan, void *stream);
int petit_nvfp4_native_image_host(void *image, const unsigned *b, const unsigned *scales, unsigned in_chan, unsigned out_chan);
int petit_nvfp4_native_image_dequant_host(float *out, const void *image, unsigned in_chan, unsigned out_chan);
int petit_nvfp4_native_attach(const void *b, const void *image);
const void *petit_nvfp4_native_attached(const void *b);
int petit_gemm_nvfp4_native(void *c, const void *a, const void *image, const float *global_scale, unsigned m, unsigned n, unsigned k,
                            const petit_solution_hints *hints, uint64_t solution_id, const petit_epilogue *epilogue,
                            const petit_native_args *native, void *workspace, uint64_t workspace_bytes, void *stream);

/*
 * Grouped launch (no counterpart in the reference): up to PETIT_GROUP_MAX weight matrices that share the activation rows --
 * q / k / v (or their tensor-parallel shards) kept as separate tensors, gate and up, the experts a token routes to -- in ONE
 * kernel launch: c_i[m][n_i] = a[m][k] . dequant(b_i)[n_i][k]^T * (*global_scale_i) (+ bias_i).  At decode batch sizes a
 * TP-8 shard's GEMM takes 3-4 us of which ~1.6 us is the dependent-dispatch gap between launches on MI355X; a group pays it
 * once.  Same numerics, bit for bit, as count separate calls with the same kernel id.
 *   m <= 16 (larger m is not launch-bound: PETIT_ERROR_KERNEL_SHAPE, call per member); every n_i % 16 == 0, k % 256 == 0;
 *   hints->b_type selects NVFP4 or MXFP4 for ALL members; solution_id PETIT_SOLUTION_AUTO (picked for the concatenated
 *   problem) or the id of a decode / staged streaming kernel; no scratch, no SiLU-mul.
 */
#define PETIT_GROUP_MAX 8
typedef struct petit_group_member {
    void *c;                   /* [m][n] output, hints->c_type */
    const void *b;             /* packed weights of this member */
    const void *scales;        /* packed scales */
    const float *global_scale; /* device pointer */
    const void *bias;          /* [n] or NULL */
    uint32_t n;
    uint32_t reserved;         /* 0 */
} petit_group_member;
int petit_gemm_fp4_fp16_grouped(const petit_group_member *members, unsigned count, const unsigned *a, unsigned m, unsigned k,
                                const petit_solution_hints *hints, uint64_t solution_id, void *stream);

/*
 * Tune-and-persist (replaces the reference's `bench_matmul -algo tune`, tools/benchmarks/matmul/main.cc:269-325, which
 * enumerates and times every solution on the user's device but leaves the winning id for the user to carry around).
 *
 * petit_gemm_tune() runs every kernel of the class that fits (m, n, k) and `workspace_bytes`, with the K splits its kind
 * supports: first CHECKS the candidate's output against the class's reference kernel (|c - ref| <= tolerance * max(rms(ref), |ref|),
 * every element), then times it with HIP events on `stream` (launches rotate over the weight copies so that the 256 MB
 * Infinity Cache cannot serve them), and returns the fastest id and its microseconds per launch.  With persist != 0 the
 * winner becomes what PETIT_SOLUTION_AUTO (klass 0) or PETIT_SOLUTION_AUTO_NATIVE_* (klass 8 / 4) picks for (dtypes, n, k)
 * and the M bucket of m (or [m_lo, m_hi] when given) from now on, in this process; petit_tune_save() writes all such rows in
 * the $PETIT_AMD_TUNE_FILE format, which a later process loads on its first call.  `c` is scratch output (overwritten).
 * The call synchronises `stream`, allocates device memory (from the pool of petit_tune_reserve when there is one, else with hipMalloc)
 * and must not run inside a graph capture (PETIT_ERROR_BAD_ARGUMENT).  Returns PETIT_ERROR_KERNEL_SHAPE when no candidate passed.
 * $PETIT_AMD_AUTOTUNE=1 does the same implicitly: the first PETIT_SOLUTION_AUTO call of a (dtypes, n, k, M bucket) that no table
 * knows is tuned in place before it runs -- once per key and process, with the scratch of THAT call (so the winner is a kernel this
 * caller can run) and device memory from the pool of petit_tune_reserve (else hipMalloc).  Every tuning run exchanges the calling
 * thread's stream-capture mode to hipStreamCaptureModeRelaxed for its duration: measured on ROCm 7.2 (tools/probes/capture_legal.hip),
 * that is what lets it synchronise its own stream and allocate while a capture is in progress on ANOTHER stream of the process (global
 * capture mode: torch.cuda.graph) without invalidating that capture; a capture on the tuning stream itself is refused as above.  When
 * $PETIT_AMD_TUNE_FILE is set the new row is merged into that file (rows other processes saved meanwhile are kept; the file is
 * replaced by rename(), under an advisory lock on "<file>.lock").
 */
typedef struct petit_tune_params {
    uint32_t struct_bytes;       /* sizeof(petit_tune_params) */
    int32_t klass;               /* 0: the exact kernels; 8 / 4: the native class, MXFP8 / MXFP4 activations (MXFP4 weights only) */
    uint32_t n_copies;           /* >= 1 packed (weights, scales) pairs to rotate over */
    uint32_t launches;           /* launches per timed sample; 0 = sized so that a sample lasts ~0.3 ms */
    const void *const *b;        /* n_copies device pointers: packed weights */
    const void *const *scales;   /* n_copies device pointers: packed scales */
    uint64_t rotate_bytes;       /* n_copies == 1 only: clone the pair on the device until the rotation covers this many bytes
                                    (0 = time on the single copy: cache-resident numbers for small shapes) */
    uint32_t samples;            /* timed samples per candidate, median reported; 0 = 5 */
    float tolerance;             /* 0 = 2e-2 */
    int32_t persist;             /* insert the winner into the run-time arch table */
    uint32_t m_lo, m_hi;         /* M range of the persisted row; 0, 0 = the bucket of m (1, 2, 3-4, 5-8, ..., 129-256, 257+) */
    uint32_t reserved;
} petit_tune_params;
int petit_gemm_tune(unsigned *c, const unsigned *a, const float *global_scale, unsigned m, unsigned n, unsigned k,
                    const petit_solution_hints *hints, const petit_tune_params *params, void *workspace, uint64_t workspace_bytes,
                    void *stream, uint64_t *best_solution, float *best_us);
/* Hand the tuner a block of device memory (256-byte aligned; the caller keeps ownership and must keep it alive until it reserves
 * another block or nullptr) for the current device: the reference output, result words and the clones of the caller's weights that a
 * tuning run rotates over come out of it.  Sized for the largest problem to be tuned: M*N*2 bytes + as many (weights + scales) copies
 * as fit, ideally > 256 MB of them (fewer copies = timings that see the Infinity Cache; the ranking still holds); what does not fit is
 * allocated with hipMalloc for the run.  Optional: it spares a serving process the allocation of a few hundred MB inside a GEMM call, and
 * it is what makes tuning possible when the process's own allocator already holds the whole device. */
int petit_tune_reserve(void *device_ptr, uint64_t bytes);
/* Add one row by hand (e.g. from a sweep done elsewhere): solution must be a kernel id of this build. */
int petit_tune_insert(const petit_solution_hints *hints, unsigned n, unsigned k, unsigned m_lo, unsigned m_hi, uint64_t solution);
/* Write the run-time rows and the rows loaded from $PETIT_AMD_TUNE_FILE to `path` (same format). */
int petit_tune_save(const char *path);
/* Bumped whenever a row is added: callers that memoise petit_gemm_workspace_bytes / default picks key their cache on it. */
uint64_t petit_tune_generation(void);

/* Human-readable text for a return code. */
const char *petit_error_string(int code);
/* Layout tag of the packed tensors ("petit-cdna4/1") and library version. */
const char *petit_layout_tag(void);
const char *petit_version(void);
/* One-line description of a solution id ("stream bf16xnvfp4 mt1 nt1 wn1 wk8 d8 ..."),
 * written into buf (at most len bytes, NUL-terminated). Returns 0 or
 * PETIT_ERROR_KERNEL_SHAPE for an unknown id. */
int petit_describe_solution(uint64_t solution_id, char *buf, unsigned len);

#ifdef __cplusplus
}
#endif
#endif /* PETIT_AMD_H_ */
