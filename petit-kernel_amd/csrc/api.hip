// api.hip -- the C ABI of include/petit_amd.h: argument checks and forwarding; the work is in dispatch.hip (gemm_impl), pick.hip (default picks),
// solutions.hip (ids), repack.hip / nvnative.hip / tune.hip.
//
// Replaces (reference paths under lib/gemm/rocm/quantization/):
//   fp4/gemm_fp4_fp16_grid.cc:11-95   Dispatcher, GemmFp4Fp16GridImpl, GemmMxFp4Fp16Grid
//   fp4/algo_chooser.cc:14-132        GemmGetSolutions, ChooseDefaultFp4Fp16Solution
#include <hip/hip_runtime.h>

#include <atomic>
#include <mutex>
#include <unordered_map>
#include <vector>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/petit_amd.h"
#include "dispatch.h"

using namespace petit_amd;

extern "C" {


int petit_gemm_fp4_fp16_grid(unsigned *c, const unsigned *a, const unsigned *b, const unsigned *scales,
                             const float *global_scale, unsigned m, unsigned n, unsigned k,
                             const petit_solution_hints *hints, uint64_t solution_id, void *stream) {
    return gemm_impl(kDataTypeFp4e2m1, c, a, b, scales, global_scale, m, n, k, hints, solution_id, nullptr, nullptr, 0, stream);
}

int petit_gemm_fp4_fp16_grid_ex(unsigned *c, const unsigned *a, const unsigned *b, const unsigned *scales,
                                const float *global_scale, unsigned m, unsigned n, unsigned k,
                                const petit_solution_hints *hints, uint64_t solution_id,
                                const petit_epilogue *epilogue, void *stream) {
    return gemm_impl(kDataTypeFp4e2m1, c, a, b, scales, global_scale, m, n, k, hints, solution_id, epilogue, nullptr, 0, stream);
}

int petit_gemm_fp4_fp16_grid_ws(unsigned *c, const unsigned *a, const unsigned *b, const unsigned *scales,
                                const float *global_scale, unsigned m, unsigned n, unsigned k,
                                const petit_solution_hints *hints, uint64_t solution_id,
                                const petit_epilogue *epilogue, void *workspace, uint64_t workspace_bytes, void *stream) {
    return gemm_impl(kDataTypeFp4e2m1, c, a, b, scales, global_scale, m, n, k, hints, solution_id, epilogue, workspace,
                     workspace_bytes, stream);
}

int petit_gemm_mxfp4_fp16_grid(unsigned *c, const unsigned *a, const unsigned *b, const unsigned *scales,
                               const float *global_scale, unsigned m, unsigned n, unsigned k,
                               const petit_solution_hints *hints, uint64_t solution_id, void *stream) {
    // the reference forces element_b = MxFp4 into the id (gemm_fp4_fp16_grid.cc:79-95): find_explicit does the same
    return gemm_impl(kDataTypeMxFp4e2m1, c, a, b, scales, global_scale, m, n, k, hints, solution_id, nullptr, nullptr, 0, stream);
}

int petit_gemm_mxfp4_fp16_grid_ex(unsigned *c, const unsigned *a, const unsigned *b, const unsigned *scales,
                                  const float *global_scale, unsigned m, unsigned n, unsigned k,
                                  const petit_solution_hints *hints, uint64_t solution_id,
                                  const petit_epilogue *epilogue, void *stream) {
    return gemm_impl(kDataTypeMxFp4e2m1, c, a, b, scales, global_scale, m, n, k, hints, solution_id, epilogue, nullptr, 0, stream);
}

int petit_gemm_mxfp4_fp16_grid_ws(unsigned *c, const unsigned *a, const unsigned *b, const unsigned *scales,
                                  const float *global_scale, unsigned m, unsigned n, unsigned k,
                                  const petit_solution_hints *hints, uint64_t solution_id,
                                  const petit_epilogue *epilogue, void *workspace, uint64_t workspace_bytes, void *stream) {
    return gemm_impl(kDataTypeMxFp4e2m1, c, a, b, scales, global_scale, m, n, k, hints, solution_id, epilogue, workspace,
                     workspace_bytes, stream);
}

// what a call with this epilogue would resolve PETIT_SOLUTION_AUTO to (the SiLU-mul epilogue restricts the candidates)
static bool epilogue_act(const petit_epilogue *epilogue, bool *ok) {
    *ok = !epilogue || ((epilogue->activation == PETIT_ACTIVATION_NONE || epilogue->activation == PETIT_ACTIVATION_SILU_MUL) &&
                        epilogue->reserved == 0);
    return epilogue && epilogue->activation == PETIT_ACTIVATION_SILU_MUL;
}

uint64_t petit_gemm_workspace_bytes_ex(const petit_solution_hints *hints, unsigned m, unsigned n, unsigned k,
                                       uint64_t solution_id, const petit_epilogue *epilogue) {
    Family fam;
    bool ok;
    const bool act = epilogue_act(epilogue, &ok);
    if (!ok || !hints)
        return 0;
    const petit_solution_hints eff = effective_hints(hints);
    hints = &eff;
    if (hints->c_type != hints->a_type || !family_for(hints->a_type, hints->b_type, &fam) || !shape_ok(n, k) || !problem_in_range(m, n, k))
        return 0;
    if (is_auto_id(solution_id)) {
        int klass = auto_class(solution_id);
        if (const int dflt = auto_default_class(solution_id, hints->b_type, m)) // (the process-wide default class: size the scratch it needs)
            klass = dflt;
        const int dev = current_device();
        const AutoChoice ch = choose_auto(fam, dev, hints->a_type, hints->b_type, act, m, n, k, klass);
        if (!ch.entry)
            return 0;
        if (klass == kClassExact && !autotune_enabled()) {
            if (const unsigned m1 = plan_row_split(*ch.entry, ch.splitk, m, n, k, arch_info(dev).num_cus)) { // bulk + tail share the scratch
                const AutoChoice c1 = choose_auto(fam, dev, hints->a_type, hints->b_type, act, m1, n, k, klass);
                const AutoChoice c2 = choose_auto(fam, dev, hints->a_type, hints->b_type, act, m - m1, n, k, klass);
                return std::max(c1.entry ? workspace_need(*c1.entry, c1.splitk, m1, n, k) : 0, c2.entry ? workspace_need(*c2.entry, c2.splitk, m - m1, n, k) : 0);
            }
        }
        return workspace_need(*ch.entry, ch.splitk, m, n, k);
    }
    const SolutionEntry *e = find_explicit(fam, solution_id);
    const unsigned splitk = solution_splitk(solution_id);
    return e && splitk ? workspace_need(*e, splitk, m, n, k) : 0;
}

static bool native_args_ok(const petit_native_args *na) {
    return !na || (na->struct_bytes == sizeof(petit_native_args) && na->reserved == 0 &&
                   (na->a_format == 0 || na->a_format == 8 || na->a_format == 6 || na->a_format == 4) &&
                   (na->out_format == 0 || na->out_format == 8 || na->out_format == 6 || na->out_format == 4));
}

int petit_gemm_mxfp4_native(void *c, const void *a, const unsigned *b, const unsigned *scales, const float *global_scale, unsigned m,
                            unsigned n, unsigned k, const petit_solution_hints *hints, uint64_t solution_id, const petit_epilogue *epilogue,
                            const petit_native_args *native, void *workspace, uint64_t workspace_bytes, void *stream) {
    if (!native_args_ok(native))
        return kErrBadArgument;
    const NativeIo io{native ? (unsigned)native->a_format : 0u, native ? (unsigned)native->out_format : 0u};
    if (solution_id == PETIT_SOLUTION_AUTO)
        return kErrKernelShape; // this entry point is the native class's: name a sentinel or a native kernel id
    return gemm_impl(kDataTypeMxFp4e2m1, (unsigned *)c, (const unsigned *)a, b, scales, global_scale, m, n, k, hints, solution_id, epilogue,
                     workspace, workspace_bytes, stream, &io);
}

uint64_t petit_gemm_native_workspace_bytes(const petit_solution_hints *hints, unsigned m, unsigned n, unsigned k, uint64_t solution_id,
                                           const petit_epilogue *epilogue, const petit_native_args *native) {
    Family fam;
    bool ok;
    const bool act = epilogue_act(epilogue, &ok);
    if (!ok || !native_args_ok(native) || !hints || hints->c_type != hints->a_type || (hints->b_type != kDataTypeMxFp4e2m1 && hints->b_type != kDataTypeFp4e2m1) ||
        !family_for(hints->a_type, hints->b_type, &fam) || !shape_ok(n, k) || m == 0 || solution_id == PETIT_SOLUTION_AUTO)
        return 0;
    const unsigned a_format = native ? (unsigned)native->a_format : 0u, out_format = native ? (unsigned)native->out_format : 0u;
    const unsigned restrict_ = (a_format ? kNeedK32 : 0u) | (out_format ? kNeedQuantOut : 0u);
    const SolutionEntry *e = nullptr;
    unsigned splitk = 1;
    if (is_auto_id(solution_id)) {
        const AutoChoice ch = choose_auto(fam, current_device(), hints->a_type, hints->b_type, act, m, n, k, auto_class(solution_id), restrict_);
        e = ch.entry, splitk = ch.splitk;
    } else {
        e = find_explicit(fam, solution_id);
        splitk = solution_splitk(solution_id);
    }
    return e && splitk ? workspace_need(*e, splitk, m, n, k, a_format != 0) : 0;
}

uint64_t petit_nvfp4_native_image_bytes(unsigned in_chan, unsigned out_chan) {
    return (out_chan % kTileN || in_chan % 256) ? 0 : nv6_image_bytes(out_chan, in_chan);
}
int petit_nvfp4_native_image(void *image, const unsigned *b, const unsigned *scales, unsigned in_chan, unsigned out_chan, void *stream) {
    if ((!image || !b || !scales || ((uintptr_t)image & 255)) && in_chan && out_chan)
        return kErrBadArgument;
    return nv6_image(image, b, scales, out_chan, in_chan, (hipStream_t)stream);
}
int petit_nvfp4_native_image_host(void *image, const unsigned *b, const unsigned *scales, unsigned in_chan, unsigned out_chan) {
    if ((!image || !b || !scales) && in_chan && out_chan)
        return kErrBadArgument;
    return nv6_image_host(image, b, scales, out_chan, in_chan);
}
int petit_nvfp4_native_image_dequant_host(float *out, const void *image, unsigned in_chan, unsigned out_chan) {
    if ((!out || !image) && in_chan && out_chan)
        return kErrBadArgument;
    return nv6_image_dequant_host(out, image, out_chan, in_chan);
}
int petit_nvfp4_native_attach(const void *b, const void *image) {
    if (!b || ((uintptr_t)image & 255))
        return kErrBadArgument;
    return attach_image(b, image);
}
const void *petit_nvfp4_native_attached(const void *b) { return b ? attached_image(b) : nullptr; }

int petit_gemm_nvfp4_native(void *c, const void *a, const void *image, const float *global_scale, unsigned m, unsigned n, unsigned k,
                            const petit_solution_hints *hints, uint64_t solution_id, const petit_epilogue *epilogue,
                            const petit_native_args *native, void *workspace, uint64_t workspace_bytes, void *stream) {
    if (!native_args_ok(native) || ((uintptr_t)image & 255))
        return kErrBadArgument;
    const NativeIo io{native ? (unsigned)native->a_format : 0u, native ? (unsigned)native->out_format : 0u, image};
    if (solution_id == PETIT_SOLUTION_AUTO)
        return kErrKernelShape; // this entry point is the native class's: name a sentinel or a native kernel id
    // (b / scales: the image stands in for both -- gemm_impl reads neither once it has the image)
    return gemm_impl(kDataTypeFp4e2m1, (unsigned *)c, (const unsigned *)a, (const unsigned *)image, (const unsigned *)image, global_scale, m, n, k, hints,
                     solution_id, epilogue, workspace, workspace_bytes, stream, &io);
}

int petit_gemm_fp4_fp16_grouped(const petit_group_member *members, unsigned count, const unsigned *a, unsigned m, unsigned k,
                                const petit_solution_hints *hints, uint64_t solution_id, void *stream) {
    if (count == 0 || m == 0 || k == 0)
        return kOk;
    if (!members || !a || !hints || count > (unsigned)kMaxGroup)
        return kErrBadArgument;
    if (hints->c_type != hints->a_type)
        return kErrKernelShape;
    const petit_solution_hints eff = effective_hints(hints);
    hints = &eff;
    Family fam;
    if (!family_for(hints->a_type, hints->b_type, &fam))
        return kErrKernelShape;
    GroupTable g{};
    g.count = count;
    uint64_t n_total = 0;
    for (unsigned i = 0; i < count; ++i) {
        const petit_group_member &mb = members[i];
        if (!mb.c || !mb.b || !mb.scales || !mb.global_scale || mb.reserved != 0)
            return kErrBadArgument;
        if (mb.n == 0 || !shape_ok(mb.n, k))
            return kErrProblemShape;
        g.n[i] = mb.n, g.w[i] = mb.b, g.s[i] = mb.scales, g.c[i] = mb.c, g.gs[i] = mb.global_scale, g.bias[i] = mb.bias;
        n_total += mb.n;
    }
    if ((uint64_t)k * 16 * 4 * 2 >= (1ull << 31) || n_total >= (1ull << 31))
        return kErrProblemShape;
    if (m > 16)
        return kErrKernelShape; // grouped launches serve the decode regime (launch-gap-bound shapes); larger M: call per member
    const SolutionEntry *entry = nullptr;
    if (solution_id == PETIT_SOLUTION_AUTO) {
        // the pick for the CONCATENATED problem (the whole grid is what fills the chip), among the kernels that have a grouped form
        const AutoChoice ch = choose_auto(fam, current_device(), hints->a_type, hints->b_type, false, m, (unsigned)n_total, k);
        entry = ch.entry && ch.entry->launch_grouped && ch.splitk == 1 ? ch.entry : heuristic(fam, m, (unsigned)n_total, k, false, nullptr, true);
    } else {
        entry = find_explicit(fam, solution_id);
        if (entry && (!entry_fits(*entry, m, k) || solution_splitk(solution_id) != 1))
            return kErrProblemShape;
    }
    if (!entry || !entry->launch_grouped)
        return kErrKernelShape;
    return entry->launch_grouped(g, a, m, k, (hipStream_t)stream);
}

uint64_t petit_quantized_activation_bytes(unsigned m, unsigned k, int format) {
    return format == 8 ? native32_ws_bytes<8>(m, k) : format == 6 ? native32_ws_bytes<6>(m, k) : format == 4 ? native32_ws_bytes<4>(m, k) : 0;
}

int petit_quantize_activations(void *qa, const void *a, unsigned m, unsigned k, int a_type, int format, void *stream) {
    if (m == 0 || k == 0)
        return kOk;
    if (!qa || !a || ((uintptr_t)qa & 15) || ((uintptr_t)a & 15))
        return kErrBadArgument;
    if (k % 256 != 0)
        return kErrProblemShape;
    if (a_type == kDataTypeBf16)
        return quantize32_bf16(a, qa, m, k, format, (hipStream_t)stream);
    if (a_type == kDataTypeFp16)
        return quantize32_f16(a, qa, m, k, format, (hipStream_t)stream);
    return kErrKernelShape;
}

uint64_t petit_gemm_workspace_bytes(const petit_solution_hints *hints, unsigned m, unsigned n, unsigned k,
                                    uint64_t solution_id) {
    return petit_gemm_workspace_bytes_ex(hints, m, n, k, solution_id, nullptr);
}

int petit_gemm_get_solutions(const petit_solution_hints *hints, unsigned m, unsigned n, unsigned k,
                             uint64_t *sols, unsigned *n_sols) {
    if (!hints || !n_sols)
        return -1;
    if (hints->b_type != kDataTypeFp4e2m1 && !is_mx_type(hints->b_type))
        return -1; // algo_chooser.cc:20-23
    const petit_solution_hints eff = effective_hints(hints);
    hints = &eff;
    Family fam;
    unsigned count = 0;
    const unsigned cap = sols ? *n_sols : 0;
    if (hints->c_type == hints->a_type && family_for(hints->a_type, hints->b_type, &fam) && shape_ok(n, k) && problem_in_range(m, n, k)) {
        for (int i = 0; i < fam.count; ++i) {
            if (!entry_fits(fam.entries[i], m, k))
                continue;
            if (is_native_am(fam.entries[i].shape.am) && !native_enabled())
                continue;
            if (sols && count < cap)
                sols[count] = entry_id(fam, fam.entries[i]);
            ++count;
        }
    }
    *n_sols = count;
    return 0;
}

uint64_t petit_gemm_resolve_solution(const petit_solution_hints *hints, unsigned m, unsigned n, unsigned k, uint64_t solution_id,
                                     const petit_epilogue *epilogue, uint64_t workspace_bytes) {
    Family fam;
    bool ok;
    const bool act = epilogue_act(epilogue, &ok);
    if (!ok || !hints)
        return 0;
    const petit_solution_hints eff = effective_hints(hints);
    hints = &eff;
    if (hints->c_type != hints->a_type || !family_for(hints->a_type, hints->b_type, &fam) || !shape_ok(n, k) || !problem_in_range(m, n, k))
        return 0;
    if (act && (n % 32 != 0 || (uint64_t)n * k / 2 >= (1ull << 32)))
        return 0; // (gemm_impl: SiLU-mul needs gate / up halves of whole n-tiles inside one descriptor)
    if (!is_auto_id(solution_id)) {
        const SolutionEntry *e = find_explicit(fam, solution_id);
        const unsigned sk = solution_splitk(solution_id);
        if (!e || !sk || !entry_fits(*e, m, k) || (act && !act_runs(*e, sk)) || workspace_need(*e, sk, m, n, k) > workspace_bytes)
            return 0;
        return make_solution_id(e->shape, fam.elem_b, entry_mfma(fam, *e), sk);
    }
    int klass = auto_class(solution_id);
    if (klass != kClassExact && ((uint64_t)m * k >= (1ull << 32) || m > 65535u))
        return 0; // (the native class's descriptor range: gemm_impl refuses the same)
    if (const int dflt = auto_default_class(solution_id, hints->b_type, m)) { // the process-wide default class, when `workspace_bytes` covers its pick (gemm_impl)
        const AutoChoice chn = choose_auto(fam, current_device(), hints->a_type, hints->b_type, act, m, n, k, dflt);
        if (chn.entry && workspace_need(*chn.entry, chn.splitk, m, n, k) <= workspace_bytes)
            klass = dflt;
    }
    AutoChoice ch = choose_auto(fam, current_device(), hints->a_type, hints->b_type, act, m, n, k, klass);
    if (ch.entry && workspace_need(*ch.entry, ch.splitk, m, n, k) > workspace_bytes) {
        // exactly what gemm_impl does when the caller's scratch does not cover the pick
        if (klass == kClassExact) {
            ch.entry = heuristic(fam, m, n, k, act), ch.splitk = 1;
        } else {
            if (act && !act_ok(*ch.entry)) { // (unsplit, SiLU-mul is the kernel's own epilogue's: gemm_impl re-picks the same way)
                unsigned sk1 = 1;
                ch.entry = heuristic_native(fam, klass, m, n, k, true, /*have_slabs=*/false, &sk1);
            }
            if (ch.entry && workspace_need(*ch.entry, 1, m, n, k) <= workspace_bytes)
                ch.splitk = 1;
            else
                ch.entry = nullptr;
        }
    }
    return ch.entry ? make_solution_id(ch.entry->shape, fam.elem_b, entry_mfma(fam, *ch.entry), ch.splitk) : 0;
}

uint64_t petit_gemm_default_solution(const petit_solution_hints *hints, unsigned m, unsigned n, unsigned k) {
    return petit_gemm_resolve_solution(hints, m, n, k, PETIT_SOLUTION_AUTO, nullptr, UINT64_MAX);
}

void petit_raster_tile(unsigned nx, unsigned ny, unsigned band, unsigned block, unsigned *bn, unsigned *bm) {
    unsigned n_ = 0, m_ = 0;
    if (nx && ny && block < nx * ny)
        tile_of_linear(block, nx, ny, kFlagXcdRaster | ((band & 0xffu) << kFlagBandShift), n_, m_);
    if (bn)
        *bn = n_;
    if (bm)
        *bm = m_;
}

unsigned petit_gemm_row_split(const petit_solution_hints *hints, unsigned m, unsigned n, unsigned k, uint64_t solution_id, const petit_epilogue *epilogue) {
    if (solution_id == PETIT_SOLUTION_AUTO)
        return petit_gemm_auto_row_split(hints, m, n, k, epilogue);
    Family fam;
    bool ok;
    const bool act = epilogue_act(epilogue, &ok);
    const int klass = auto_class(solution_id);
    if (!ok || !hints || klass == kClassExact)
        return 0; // (explicit ids run as named)
    const petit_solution_hints eff = effective_hints(hints);
    if (eff.c_type != eff.a_type || !family_for(eff.a_type, eff.b_type, &fam) || !shape_ok(n, k) || !problem_in_range(m, n, k) || autotune_enabled() ||
        (uint64_t)m * k >= (1ull << 32) || m > 65535u)
        return 0;
    const int dev = current_device();
    const AutoChoice ch = choose_auto(fam, dev, eff.a_type, eff.b_type, act, m, n, k, klass);
    return ch.entry ? plan_row_split_native(*ch.entry, klass, ch.splitk, m, n, k, arch_info(dev).num_cus) : 0;
}

unsigned petit_gemm_auto_row_split(const petit_solution_hints *hints, unsigned m, unsigned n, unsigned k, const petit_epilogue *epilogue) {
    Family fam;
    bool ok;
    const bool act = epilogue_act(epilogue, &ok);
    if (!ok || !hints)
        return 0;
    const petit_solution_hints eff = effective_hints(hints);
    if (eff.c_type != eff.a_type || !family_for(eff.a_type, eff.b_type, &fam) || !shape_ok(n, k) || m == 0 || m > kMaxM || autotune_enabled() ||
        auto_default_class(PETIT_SOLUTION_AUTO, eff.b_type, m))
        return 0;
    const int dev = current_device();
    const AutoChoice ch = choose_auto(fam, dev, eff.a_type, eff.b_type, act, m, n, k, kClassExact);
    return ch.entry ? plan_row_split(*ch.entry, ch.splitk, m, n, k, arch_info(dev).num_cus) : 0;
}

int petit_repack_nvfp4_weights(unsigned *output, const unsigned *input, unsigned in_chan, unsigned out_chan,
                               void *stream) {
    if ((!output || !input) && in_chan && out_chan)
        return kErrBadArgument;
    return repack_weights(output, input, in_chan, out_chan, (hipStream_t)stream);
}
int petit_repack_nvfp4_scales(unsigned *out_scales, const unsigned *scales, unsigned in_chan, unsigned out_chan,
                              void *stream) {
    if ((!out_scales || !scales) && in_chan && out_chan)
        return kErrBadArgument;
    return repack_nvscales(out_scales, scales, in_chan, out_chan, (hipStream_t)stream);
}
int petit_repack_mxfp4_scales(unsigned *out_scales, const unsigned *scales, unsigned in_chan, unsigned out_chan,
                              void *stream) {
    if ((!out_scales || !scales) && in_chan && out_chan)
        return kErrBadArgument;
    return repack_mxscales(out_scales, scales, in_chan, out_chan, (hipStream_t)stream);
}

int petit_repack_nvfp4_weights_host(unsigned *output, const unsigned *input, unsigned in_chan, unsigned out_chan) {
    if ((!output || !input || output == input) && in_chan && out_chan)
        return kErrBadArgument;
    return repack_weights_host(output, input, in_chan, out_chan);
}
int petit_repack_nvfp4_scales_host(unsigned *out_scales, const unsigned *scales, unsigned in_chan, unsigned out_chan) {
    if ((!out_scales || !scales || out_scales == scales) && in_chan && out_chan)
        return kErrBadArgument;
    return repack_nvscales_host(out_scales, scales, in_chan, out_chan);
}
int petit_repack_mxfp4_scales_host(unsigned *out_scales, const unsigned *scales, unsigned in_chan, unsigned out_chan) {
    if ((!out_scales || !scales || out_scales == scales) && in_chan && out_chan)
        return kErrBadArgument;
    return repack_mxscales_host(out_scales, scales, in_chan, out_chan);
}

int petit_convert_reference_weights_host(unsigned *output, const unsigned *input, unsigned in_chan, unsigned out_chan) {
    if ((!output || !input || output == input) && in_chan && out_chan)
        return kErrBadArgument;
    return convert_reference_weights_host(output, input, in_chan, out_chan);
}
int petit_convert_reference_nvfp4_scales_host(unsigned *out_scales, const unsigned *scales, unsigned in_chan, unsigned out_chan) {
    if ((!out_scales || !scales || out_scales == scales) && in_chan && out_chan)
        return kErrBadArgument;
    return convert_reference_nvscales_host(out_scales, scales, in_chan, out_chan);
}
int petit_convert_reference_mxfp4_scales_host(unsigned *out_scales, const unsigned *scales, unsigned in_chan, unsigned out_chan) {
    if ((!out_scales || !scales || out_scales == scales) && in_chan && out_chan)
        return kErrBadArgument;
    return convert_reference_mxscales_host(out_scales, scales, in_chan, out_chan);
}

int petit_dequant_packed_weights(void *out, const unsigned *b, const unsigned *scales, float global_scale, unsigned n, unsigned k,
                                 int b_type, int out_type, void *stream) {
    if ((!out || !b || !scales) && n && k)
        return kErrBadArgument;
    const int kind = out_type == kDataTypeBf16 ? 1 : out_type == kDataTypeFp16 ? 2 : out_type == PETIT_DTYPE_FP32 ? 0 : -1;
    return dequant_packed(out, b, scales, global_scale, n, k, b_type, kind, (hipStream_t)stream);
}

int petit_set_workspace(void *device_ptr, uint64_t bytes) {
    if ((uintptr_t)device_ptr & (kWorkspaceAlign - 1))
        return kErrBadArgument;
    Workspace &ws = g_workspace[current_device()];
    ws.bytes.store(0);
    ws.ptr.store(device_ptr);
    ws.stream.store(Workspace::kUnbound); // binds again to the first stream that uses it
    ws.bytes.store(device_ptr ? bytes : 0);
    return kOk;
}

uint64_t petit_workspace_bytes(uint64_t solution_id, unsigned m, unsigned n) {
    return splitk_bytes(solution_splitk(solution_id), m, n); // (split-K slabs only: see petit_gemm_workspace_bytes)
}

int petit_enable_native_fp4(int enable) {
    g_native_enabled.store(enable ? 1 : 0);
    return kOk;
}

int petit_set_mxfp4_default_class(int activation_format) {
    if (activation_format != 0 && activation_format != 8 && activation_format != 6 && activation_format != 4)
        return kErrBadArgument;
    g_mxfp4_default_class.store(activation_format, std::memory_order_relaxed);
    return kOk;
}
int petit_get_mxfp4_default_class(void) { return mxfp4_default_class(); }

uint64_t petit_native_workspace_bytes(unsigned m, unsigned k) { return native_ws_bytes(m, k); }

const char *petit_layout_tag(void) { return "petit-cdna4/1"; }
const char *petit_version(void) { return "petit-kernel_amd 0.1.0 (gfx950)"; }

} // extern "C"
