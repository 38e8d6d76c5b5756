// api.hip -- host side of libpetit_amd.so: the C ABI of include/petit_amd.h,
// solution lookup and the default-solution choice.
//
// Replaces (reference paths under lib/gemm/rocm/quantization/):
//   fp4/gemm_fp4_fp16_grid.cc:11-95   Dispatcher, GemmFp4Fp16GridImpl, GemmMxFp4Fp16Grid
//   fp4/algo_chooser.cc:14-132        GemmGetSolutions, ChooseDefaultFp4Fp16Solution
//   fp4/solution_map.cc, fp4/gen_solution_list.cc   (build-time kernel list)
// The reference scans a 234-entry map on every call with solution_id = -1
// (algo_chooser.cc:116-126); here the choice is a handful of integer compares
// against the arch table (hal.h) and the table lookup is a linear scan over
// < 20 entries of one (dtype, format) family.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/petit_amd.h"
#include "gemm_native.cuh"
#include "gemm_stream.cuh"
#include "hal.h"
#include "layout.h"
#include "petit_internal.h"
#include "solution.h"

namespace petit_amd {
namespace {

struct Family {
    const SolutionEntry *entries;
    int count;
    unsigned elem_b, mfma;
};

bool family_for(int a_type, int b_type, Family *out) {
    const bool mx = (b_type == kDataTypeMxFp4e2m1);
    if (b_type != kDataTypeFp4e2m1 && !mx)
        return false;
    if (a_type == kDataTypeBf16 && !mx) {
        out->entries = solutions_nv_bf16(&out->count);
        out->elem_b = kElemBNvFp4, out->mfma = kMfmaBf16;
        return true;
    }
    if (a_type == kDataTypeFp16 && !mx) {
        out->entries = solutions_nv_f16(&out->count);
        out->elem_b = kElemBNvFp4, out->mfma = kMfmaFp16;
        return true;
    }
    if (a_type == kDataTypeBf16 && mx) {
        out->entries = solutions_mx_bf16(&out->count);
        out->elem_b = kElemBMxFp4, out->mfma = kMfmaBf16;
        return true;
    }
    if (a_type == kDataTypeFp16 && mx) { // not in the reference (gemm_fp4_fp16_grid.cc:55-64 rejects it)
        out->entries = solutions_mx_f16(&out->count);
        out->elem_b = kElemBMxFp4, out->mfma = kMfmaFp16;
        return true;
    }
    return false;
}

bool shape_ok(unsigned n, unsigned k) { return n % kTileN == 0 && k % 256 == 0; }

// Can this entry run (m, n, k)?  KS has to match the layout K implies, and the
// staged-activation kernels hold at most AM rows.
bool entry_fits(const SolutionEntry &e, unsigned m, unsigned k) {
    return e.shape.ks == span_tiles_for_k(k) && (e.shape.am <= 0 || m <= (unsigned)am_rows(e.shape.am));
}

// Per-device registered split-K workspace.
constexpr int kMaxDevices = 64;
struct Workspace {
    std::atomic<void *> ptr{nullptr};
    std::atomic<uint64_t> bytes{0};
};
Workspace g_workspace[kMaxDevices];

int current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices)
        return 0;
    return dev;
}

uint64_t splitk_bytes(unsigned splitk, unsigned m, unsigned n) {
    return splitk > 1 ? (uint64_t)splitk * m * n * sizeof(float) : 0;
}

// Default choice when the arch table has no entry: pick the shape whose
// workgroup count best fills the chip without starving each wave of work.
// (The reference's heuristic ignores the CU count altogether and leaves half of
// a 256-CU part idle on 4096^2 -- SURVEY.md Appendix C.)
// SiLU-mul epilogue: a wave must hold the gate and the up tile of an output tile -> even n-tiles per wave
bool act_ok(const SolutionEntry &e) { return e.shape.nt % 2 == 0; }

// M > 16: a cost model calibrated on the r01 sweeps (profiles/r01_tune_midm_*.json, r01_tune_bigm_*.json; microseconds
// on MI355X, bf16 x NVFP4; the other families scale uniformly, which does not change the argmin much):
//  * streaming kernel, MT m-tiles per workgroup, NT n-tiles per wave: every 16*MT-row block repeats the unpack and
//    pulls its activation fragments once per n-tile: (0.5 + 2 MT/NT) e-7 us per weight fits MT = 1 / 2 / 4 at
//    NT = 4 (1.0 / 1.5 / 2.5) and MT = 4 at NT = 2 (4.5 modelled, 5.2 measured);
//  * tiled kernel: K/128 steps of t1(tile) each, times the number of rounds the grid needs on the chip (workgroups
//    are dispatched dynamically, so rounds is fractional; two-per-CU residency buys ~14 %).
// tools/check_heuristic.py replays it against every swept case.
double stream_cost_us(const SolutionEntry &e, unsigned m, unsigned n, unsigned k, int num_cus) {
    const StreamShape &s = e.shape;
    const bool split = e.a_type == kDataTypeFp16 && e.fmt == kFmtMx; // hi/lo activations: two MFMAs and two fragments per word
    const double per_weight = (0.5 + 2.0 * s.mt / s.nt) * (split ? 1.5e-7 : 1e-7); // unpack once per block + fragment loads per (m-tile, n-tile) pair
    const unsigned blocks = (m + 16 * s.mt - 1) / (16 * s.mt);
    const double wgs = (double)blocks * ((n / kTileN + s.nt * s.wn - 1) / (s.nt * s.wn));
    // VALU-bound: a CU that holds two workgroups takes twice as long, one that holds none idles
    const double rounds = (double)(((unsigned)wgs + num_cus - 1) / num_cus);
    return 2.0 + blocks * (double)n * (double)k * per_weight * rounds * num_cus / wgs;
}
double tiled_cost_us(const SolutionEntry &e, unsigned m, unsigned n, unsigned k, int num_cus) {
    const StreamShape &s = e.shape;
    const int acc = s.mt * s.nt; // accumulator tiles per wave: 8 = 64x128 / 128x64, 16 = 64x256 / 128x128
    const bool split = e.a_type == kDataTypeFp16 && e.fmt == kFmtMx; // two MFMAs per fragment, two LDS images
    double t1, resident;
    if (split) {
        t1 = s.mt == 4 && s.nt == 2 ? 1.14 : s.mt == 4 && s.nt == 4 ? 1.67 : s.mt == 8 && s.nt == 2 ? 3.0 : 0.19 * acc + 0.3;
        resident = acc <= 8 && s.mt <= 4 ? 1.14 : 1.0;
    } else {
        t1 = s.mt == 4 && s.nt == 2 ? 0.71 : s.mt == 4 && s.nt == 4 ? 1.31 : s.mt == 8 && s.nt == 2 ? 1.135
           : s.mt == 8 && s.nt == 1 ? 0.94 : s.mt == 1 && s.nt == 4 ? 0.80 : s.mt == 2 && s.nt == 4 ? 0.97 : 0.09 * acc + 0.2;
        if (e.fmt == kFmtMx)
            t1 *= 0.75; // no group-scale multiplies in the unpack
        resident = (acc <= 8 || (s.mt == 8 && s.nt == 2)) ? 1.14 : 1.0;
    }
    const unsigned per_wg = s.nt * s.wn;
    const double wgs = (double)((m + 16 * s.mt - 1) / (16 * s.mt)) * (double)((n / kTileN + per_wg - 1) / per_wg);
    double rounds = wgs / (num_cus * resident);
    if (rounds < 1.0)
        rounds = 1.0;
    return 2.0 + (k / kTileK) * t1 * rounds;
}

const SolutionEntry *heuristic(const Family &fam, unsigned m, unsigned n, unsigned k, bool need_pairs = false) {
    // Rules distilled from the MI355X sweeps (profiles/):
    //  * M <= 16: stage the activations through LDS (AM = smallest that holds M);
    //  * M <= 4: what saturates HBM is bytes in flight: as many resident waves as the grid allows, every wave with
    //    its whole ring outstanding -> the shape whose wave count is closest to (preferably above) 4 per SIMD;
    //  * 5 <= M <= 16: the activation block every workgroup pulls through L2 starts to matter: two n-tiles per wave
    //    (four when N is large), K split over 4 waves, one wave per SIMD is enough;
    //  * M > 16: the cost model above picks between the streaming shapes (MT = 1 / 2 / 4) and the tiled kernel.
    const ArchInfo &arch = arch_info(current_device());
    const unsigned ntiles = n / kTileN;
    const unsigned nspans = k / (kTileK * span_tiles_for_k(k));
    if (m > 16) {
        const SolutionEntry *best = nullptr;
        double best_us = 1e30;
        for (int i = 0; i < fam.count; ++i) {
            const SolutionEntry &e = fam.entries[i];
            const StreamShape &s = e.shape;
            if (!entry_fits(e, m, k) || s.am == kNativeAm || (need_pairs && !act_ok(e)))
                continue; // (never the native-FP4 kernels: different accuracy class)
            double us;
            if (s.am == kTiledAm) {
                us = tiled_cost_us(e, m, n, k, arch.num_cus);
                us -= 0.001 * s.d; // deeper ring on a tie
            } else {
                if (s.am != 0 || s.wn != 1)
                    continue;
                us = stream_cost_us(e, m, n, k, arch.num_cus);
                // the swept winners: WK = 4, fragments requested 2 tiles ahead
                us *= 1.0 + 0.05 * (s.wk != 4) + 0.02 * (s.pa != 2);
                if (nspans < (unsigned)s.wk)
                    us *= (double)s.wk / nspans; // idle K waves
            }
            if (us < best_us)
                best_us = us, best = &e;
        }
        if (best)
            return best;
    }
    if (m > 8 && ntiles >= 12u * arch.num_cus) {
        // very wide N (gate_up): the 16 x 256 tiled shape shares one activation tile among 256 columns; the streaming
        // kernel would pull the activations through L2 once per 32-64 columns (measured 52.9 vs 57.0 us at M = 16)
        for (int i = 0; i < fam.count; ++i) {
            const SolutionEntry &e = fam.entries[i];
            if (e.shape.am == kTiledAm && e.shape.mt == 1 && e.shape.nt == 4 && entry_fits(e, m, k) && (!need_pairs || act_ok(e)))
                return &e;
        }
    }
    const int want_mt = 1;
    const int want_am = m <= 1 ? 1 : m <= 2 ? 2 : m <= 4 ? 4 : m <= 8 ? 8 : 16;
    const bool mid = m > 4; // 5..16
    const int want_nt = m <= 2 ? 1 : (mid && ntiles >= 4u * arch.num_cus) ? 4 : 4u * ntiles >= 5u * arch.num_cus ? 2 : 1;
    const double target_waves = (double)arch.num_cus * (mid ? 4 : m > 2 ? 8 : 16);
    const SolutionEntry *best = nullptr;
    double best_score = -1e30;
    for (int i = 0; i < fam.count; ++i) {
        const SolutionEntry &e = fam.entries[i];
        if (!entry_fits(e, m, k) || (need_pairs && !act_ok(e)))
            continue;
        const StreamShape &s = e.shape;
        if (s.mt != want_mt || s.am == kTiledAm || s.am == kNativeAm)
            continue;
        const unsigned wgs = (ntiles + s.wn * s.nt - 1) / (s.wn * s.nt);
        const unsigned busy_wk = nspans < (unsigned)s.wk ? nspans : (unsigned)s.wk;
        const double busy = (double)wgs * s.wn * busy_wk;
        double score = 0.0;
        score -= 4.0 * (am_rows(s.am) != want_am);
        score += 0.5 * (s.am >= kBfpAm); // bf16 x NVFP4, M <= 4: the fp16 pipeline unpacks cheaper
        score -= 1.0 * (s.nt != want_nt);
        // wave count: under-filling costs more than over-filling
        score -= busy < target_waves ? 3.0 * (1.0 - busy / target_waves) : 0.25 * (busy / target_waves - 1.0);
        // spans must divide evenly over the K waves, or some waves idle in the tail
        const unsigned per = (nspans + s.wk - 1) / s.wk;
        score -= 2.0 * (1.0 - (double)nspans / ((double)per * s.wk));
        score += 0.01 * s.d;
        if (score > best_score)
            best_score = score, best = &e;
    }
    if (!best) { // relax the m-tile preference
        for (int i = 0; i < fam.count; ++i)
            if (entry_fits(fam.entries[i], m, k) && fam.entries[i].shape.am != kTiledAm &&
                fam.entries[i].shape.am != kNativeAm && (!need_pairs || act_ok(fam.entries[i])) &&
                (!best || fam.entries[i].shape.mt > best->shape.mt))
                best = &fam.entries[i];
    }
    return best;
}

unsigned entry_mfma(const Family &fam, const SolutionEntry &e) {
    if (e.shape.am == kNativeAm)
        return e.a_type == kDataTypeFp16 ? kMfmaFp8ActFp16 : kMfmaFp8;
    return fam.mfma;
}
uint64_t entry_id(const Family &fam, const SolutionEntry &e) {
    return make_solution_id(e.shape, fam.elem_b, entry_mfma(fam, e), 1);
}
const SolutionEntry *find_entry(const Family &fam, uint64_t id) {
    const uint64_t key = solution_without_splitk(id);
    for (int i = 0; i < fam.count; ++i)
        if (entry_id(fam, fam.entries[i]) == key)
            return &fam.entries[i];
    return nullptr;
}

// native-FP4 kernels are opt-in (own accuracy class): petit_enable_native_fp4 / $PETIT_AMD_NATIVE_FP4
std::atomic<int> g_native_enabled{-1};
bool native_enabled() {
    int v = g_native_enabled.load();
    if (v < 0) {
        const char *e = getenv("PETIT_AMD_NATIVE_FP4");
        v = (e && *e && *e != '0') ? 1 : 0;
        g_native_enabled.store(v);
    }
    return v != 0;
}

int gemm_impl(int b_type, unsigned *c, const unsigned *a, const unsigned *b, const unsigned *scales,
              const float *global_scale, unsigned m, unsigned n, unsigned k,
              const petit_solution_hints *hints, uint64_t solution_id, const petit_epilogue *epilogue, void *stream) {
    if (epilogue && ((epilogue->activation != PETIT_ACTIVATION_NONE && epilogue->activation != PETIT_ACTIVATION_SILU_MUL) ||
                     epilogue->reserved != 0))
        return kErrBadArgument; // reject what a newer caller might ask for
    const bool act = epilogue && epilogue->activation == PETIT_ACTIVATION_SILU_MUL;
    if (m == 0 || n == 0 || k == 0)
        return kOk; // gemm_fp4_fp16_grid.cc:42-44
    if (!hints || !c || !a || !b || !scales || !global_scale)
        return kErrBadArgument;
    if (hints->c_type != hints->a_type)
        return kErrKernelShape;
    Family fam;
    if (!family_for(hints->a_type, b_type, &fam))
        return kErrKernelShape;
    if (!shape_ok(n, k))
        return kErrProblemShape;
    // SiLU-mul: gate / up halves made of whole n-tiles, and one descriptor spans half the matrix
    if (act && (n % 32 != 0 || (uint64_t)n * k / 2 >= (1ull << 32)))
        return kErrProblemShape;
    // 32-bit buffer offsets inside one n-tile row / activation block
    if ((uint64_t)k * 16 * 4 * 2 >= (1ull << 31) || (uint64_t)k * 64 * 4 >= (1ull << 31))
        return kErrProblemShape;

    const SolutionEntry *entry = nullptr;
    unsigned splitk = 1;
    if (solution_id == PETIT_SOLUTION_AUTO) {
        uint64_t tuned = tuned_solution(current_device(), hints->a_type, b_type, m, n, k);
        if (tuned) {
            entry = find_entry(fam, tuned);
            splitk = solution_splitk(tuned);
        }
        if (entry && (!entry_fits(*entry, m, k) || (act && (!act_ok(*entry) || splitk != 1))))
            entry = nullptr;
        if (!entry) {
            entry = heuristic(fam, m, n, k, act);
            splitk = 1;
        }
        if (!entry)
            return kErrKernelShape;
    } else {
        entry = find_entry(fam, solution_id);
        if (!entry)
            return kErrKernelShape;
        if (!entry_fits(*entry, m, k))
            return kErrProblemShape;
        splitk = solution_splitk(solution_id);
        if (splitk == 0)
            return kErrKernelShape;
        if (act && (!act_ok(*entry) || splitk != 1))
            return kErrKernelShape; // needs an even number of n-tiles per wave and no cross-workgroup K split
    }

    GemmArgs args{};
    args.c = c, args.a = a, args.w = b, args.s = scales, args.gs = global_scale;
    args.m = m, args.n = n, args.k = k;
    args.bias = epilogue ? epilogue->bias : nullptr;
    args.act = act ? 1u : 0u;
    if (entry->shape.am == kNativeAm) {
        Workspace &ws = g_workspace[current_device()];
        if (ws.ptr.load() == nullptr || ws.bytes.load() < native_ws_bytes(m, k))
            return kErrKernelShape; // needs petit_set_workspace(>= petit_native_workspace_bytes(m, k))
        args.workspace = (float *)ws.ptr.load();
    }
    if (splitk > 1) {
        Workspace &ws = g_workspace[current_device()];
        const uint64_t need = splitk_bytes(splitk, m, n);
        if (ws.ptr.load() == nullptr || ws.bytes.load() < need) {
            if (solution_id != PETIT_SOLUTION_AUTO)
                return kErrKernelShape; // explicit id that needs a workspace nobody registered
            splitk = 1;                 // tuned pick without workspace: fall back in-family
        } else {
            args.workspace = (float *)ws.ptr.load();
        }
    }
    return entry->launch(args, splitk, (hipStream_t)stream);
}

} // namespace
} // namespace petit_amd

using namespace petit_amd;

extern "C" {

int petit_gemm_fp4_fp16_grid(unsigned *c, const unsigned *a, const unsigned *b, const unsigned *scales,
                             const float *global_scale, unsigned m, unsigned n, unsigned k,
                             const petit_solution_hints *hints, uint64_t solution_id, void *stream) {
    return gemm_impl(kDataTypeFp4e2m1, c, a, b, scales, global_scale, m, n, k, hints, solution_id, nullptr, stream);
}

int petit_gemm_fp4_fp16_grid_ex(unsigned *c, const unsigned *a, const unsigned *b, const unsigned *scales,
                                const float *global_scale, unsigned m, unsigned n, unsigned k,
                                const petit_solution_hints *hints, uint64_t solution_id,
                                const petit_epilogue *epilogue, void *stream) {
    return gemm_impl(kDataTypeFp4e2m1, c, a, b, scales, global_scale, m, n, k, hints, solution_id, epilogue, stream);
}

int petit_gemm_mxfp4_fp16_grid(unsigned *c, const unsigned *a, const unsigned *b, const unsigned *scales,
                               const float *global_scale, unsigned m, unsigned n, unsigned k,
                               const petit_solution_hints *hints, uint64_t solution_id, void *stream) {
    // the reference forces element_b = MxFp4 into the id (gemm_fp4_fp16_grid.cc:79-95)
    return gemm_impl(kDataTypeMxFp4e2m1, c, a, b, scales, global_scale, m, n, k, hints, solution_id, nullptr, stream);
}

int petit_gemm_mxfp4_fp16_grid_ex(unsigned *c, const unsigned *a, const unsigned *b, const unsigned *scales,
                                  const float *global_scale, unsigned m, unsigned n, unsigned k,
                                  const petit_solution_hints *hints, uint64_t solution_id,
                                  const petit_epilogue *epilogue, void *stream) {
    return gemm_impl(kDataTypeMxFp4e2m1, c, a, b, scales, global_scale, m, n, k, hints, solution_id, epilogue, stream);
}

int petit_gemm_get_solutions(const petit_solution_hints *hints, unsigned m, unsigned n, unsigned k,
                             uint64_t *sols, unsigned *n_sols) {
    if (!hints || !n_sols)
        return -1;
    if (hints->b_type != kDataTypeFp4e2m1 && hints->b_type != kDataTypeMxFp4e2m1)
        return -1; // algo_chooser.cc:20-23
    Family fam;
    unsigned count = 0;
    const unsigned cap = sols ? *n_sols : 0;
    if (hints->c_type == hints->a_type && family_for(hints->a_type, hints->b_type, &fam) && shape_ok(n, k)) {
        for (int i = 0; i < fam.count; ++i) {
            if (!entry_fits(fam.entries[i], m, k))
                continue;
            if (fam.entries[i].shape.am == kNativeAm && !native_enabled())
                continue;
            if (sols && count < cap)
                sols[count] = entry_id(fam, fam.entries[i]);
            ++count;
        }
    }
    *n_sols = count;
    return 0;
}

uint64_t petit_gemm_default_solution(const petit_solution_hints *hints, unsigned m, unsigned n, unsigned k) {
    Family fam;
    if (!hints || hints->c_type != hints->a_type || !family_for(hints->a_type, hints->b_type, &fam) ||
        !shape_ok(n, k) || m == 0)
        return 0;
    uint64_t tuned = tuned_solution(current_device(), hints->a_type, hints->b_type, m, n, k);
    if (tuned) {
        const SolutionEntry *e = find_entry(fam, tuned);
        if (e && entry_fits(*e, m, k))
            return tuned;
    }
    const SolutionEntry *e = heuristic(fam, m, n, k);
    return e ? entry_id(fam, *e) : 0;
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

int petit_set_workspace(void *device_ptr, uint64_t bytes) {
    Workspace &ws = g_workspace[current_device()];
    ws.bytes.store(0);
    ws.ptr.store(device_ptr);
    ws.bytes.store(device_ptr ? bytes : 0);
    return kOk;
}

uint64_t petit_workspace_bytes(uint64_t solution_id, unsigned m, unsigned n) {
    return splitk_bytes(solution_splitk(solution_id), m, n);
}

int petit_enable_native_fp4(int enable) {
    g_native_enabled.store(enable ? 1 : 0);
    return kOk;
}

uint64_t petit_native_workspace_bytes(unsigned m, unsigned k) { return native_ws_bytes(m, k); }

const char *petit_error_string(int code) {
    switch (code) {
    case kOk: return "ok";
    case kErrProblemShape: return "incompatible problem shape";
    case kErrKernelShape: return "no kernel implementation for this solution id / dtype combination";
    case kErrLaunch: return "kernel launch failed";
    case kErrBadArgument: return "bad argument";
    default: return "unknown error";
    }
}

const char *petit_layout_tag(void) { return "petit-cdna4/1"; }
const char *petit_version(void) { return "petit-kernel_amd 0.1.0 (gfx950)"; }

int petit_describe_solution(uint64_t id, char *buf, unsigned len) {
    if (!buf || len == 0)
        return kErrBadArgument;
    const unsigned elem_b = (unsigned)(id >> 28) & 0xf, mfma = (unsigned)(id >> 32) & 0xf;
    const int a_type = (mfma == kMfmaBf16 || mfma == kMfmaFp8) ? kDataTypeBf16 : kDataTypeFp16;
    const int b_type = elem_b == kElemBMxFp4 ? kDataTypeMxFp4e2m1 : kDataTypeFp4e2m1;
    Family fam;
    const SolutionEntry *e = family_for(a_type, b_type, &fam) ? find_entry(fam, id) : nullptr;
    if (!e) {
        snprintf(buf, len, "unknown solution 0x%llx", (unsigned long long)id);
        return kErrKernelShape;
    }
    const StreamShape &s = e->shape;
    if (s.am == kNativeAm) {
        snprintf(buf, len, "native-fp4 %sxmxfp4 (activations -> mxfp8) ks%d mt%d ntw%d waves%d d%d  (wg tile %dx%d, %d threads)",
                 a_type == kDataTypeBf16 ? "bf16" : "fp16", s.ks, s.mt, s.nt, s.wn, s.d, 16 * s.mt, 16 * s.wn * s.nt,
                 64 * s.wn);
        return kOk;
    }
    if (s.am == kTiledAm) {
        snprintf(buf, len, "tiled %sx%s ks%d mt%d ntw%d waves%d d%d  (wg tile %dx%d, %d threads)",
                 a_type == kDataTypeBf16 ? "bf16" : "fp16", b_type == kDataTypeMxFp4e2m1 ? "mxfp4" : "nvfp4", s.ks,
                 s.mt, s.nt, s.wn, s.d, 16 * s.mt, 16 * s.wn * s.nt, 64 * s.wn);
        return kOk;
    }
    snprintf(buf, len, "stream %sx%s ks%d mt%d nt%d wn%d wk%d d%d am%d splitk%u  (wg tile %dx%d, %d threads)",
             a_type == kDataTypeBf16 ? "bf16" : "fp16", b_type == kDataTypeMxFp4e2m1 ? "mxfp4" : "nvfp4", s.ks,
             s.mt, s.nt, s.wn, s.wk, s.d, am_rows(s.am), solution_splitk(id), 16 * s.mt, 16 * s.wn * s.nt,
             64 * s.wn * s.wk);
    if (s.am >= kBfpAm)
        strncat(buf, " bfp16", len - strlen(buf) - 1);
    if (s.pa > 1) {
        char t[16];
        snprintf(t, sizeof(t), " pa%d", s.pa);
        strncat(buf, t, len - strlen(buf) - 1);
    }
    return kOk;
}

} // extern "C"
