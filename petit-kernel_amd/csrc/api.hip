// api.hip -- host side of libpetit_amd.so: the C ABI of include/petit_amd.h,
// solution lookup and the default-solution choice.
//
// Replaces (reference paths under lib/gemm/rocm/quantization/):
//   fp4/gemm_fp4_fp16_grid.cc:11-95   Dispatcher, GemmFp4Fp16GridImpl, GemmMxFp4Fp16Grid
//   fp4/algo_chooser.cc:14-132        GemmGetSolutions, ChooseDefaultFp4Fp16Solution
//   fp4/solution_map.cc, fp4/gen_solution_list.cc   (build-time kernel list)
// The reference scans a 234-entry map on every call with solution_id = -1
// (algo_chooser.cc:116-126); here the choice is made once per (thread, problem) -- arch table (hal.h), else the
// cost model over the family's kernels -- and then served from a thread-local cache (choose_auto).
#include <hip/hip_runtime.h>

#include <atomic>
#include <mutex>
#include <unordered_map>
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/petit_amd.h"
#include "gemm_native32.hpp"
#include "gemm_stream.hpp"
#include "hal.h"
#include "layout.h"
#include "petit_internal.h"
#include "solution.h"

namespace petit_amd {

// The family tables: the parts exported by the family's translation units, concatenated once (streaming kernels first, as the
// heuristic and the tuner's reference-kernel choice expect: the plain direct-path kernel is the first entry).
using PartFn = const SolutionEntry *(*)(int *);
static const SolutionEntry *concat_parts(std::vector<SolutionEntry> &store, std::initializer_list<PartFn> parts, int *count) {
    if (store.empty())
        for (PartFn fn : parts) {
            int n = 0;
            const SolutionEntry *e = fn(&n);
            store.insert(store.end(), e, e + n);
        }
    *count = (int)store.size();
    return store.data();
}
#define PETIT_FAMILY_TABLE(fam, ...)                                                                          \
    const SolutionEntry *solutions_##fam(int *count) {                                                        \
        static std::vector<SolutionEntry> store;                                                              \
        static const SolutionEntry *const table = concat_parts(store, {__VA_ARGS__}, count);                  \
        *count = (int)store.size();                                                                           \
        return table;                                                                                         \
    }
PETIT_FAMILY_TABLE(nv_bf16, solutions_nv_bf16_p1, solutions_nv_bf16_p2, solutions_nv_bf16_p3, solutions_nv_bf16_p4, solutions_nv_bf16_p5, solutions_nv_bf16_p6)
PETIT_FAMILY_TABLE(nv_f16, solutions_nv_f16_p1, solutions_nv_f16_p2, solutions_nv_f16_p3, solutions_nv_f16_p4, solutions_nv_f16_p5, solutions_nv_f16_p6)
PETIT_FAMILY_TABLE(mx_bf16, solutions_mx_bf16_p1, solutions_mx_bf16_p2, solutions_mx_bf16_p3, solutions_mx_bf16_p4, solutions_mx_bf16_p5, solutions_mx_bf16_p6)
PETIT_FAMILY_TABLE(mx_f16, solutions_mx_f16_p1, solutions_mx_f16_p2, solutions_mx_f16_p3, solutions_mx_f16_p4, solutions_mx_f16_p5, solutions_mx_f16_p6)
#undef PETIT_FAMILY_TABLE

namespace {

struct Family {
    const SolutionEntry *entries;
    int count;
    unsigned elem_b, mfma;
};

bool family_for(int a_type, int b_type, Family *out) {
    const bool mx = is_mx_type(b_type);
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
    if (a_type == kDataTypeFp16 && mx) { // not in the reference (gemm_fp4_fp16_grid.cc:55-64 rejects it); Fp16Mx kernels: fast body + exact fallback
        out->entries = solutions_mx_f16(&out->count);
        out->elem_b = kElemBMxFp4, out->mfma = kMfmaFp16;
        return true;
    }
    return false;
}

bool shape_ok(unsigned n, unsigned k) { return n % kTileN == 0 && k % 256 == 0; }
// ... and the ranges gemm_impl refuses with PETIT_ERROR_PROBLEM_SHAPE (32-bit buffer offsets inside one n-tile row / activation block; M beyond the tables'
// last bucket): the enumeration and the default-pick queries answer "nothing" for exactly the problems the launcher would refuse (the reference's
// enumeration filters by what its kernels accept, algo_chooser.cc:14-62) -- an empty problem (m, n or k = 0: the launcher's no-op) has no kernel either
bool problem_in_range(unsigned m, unsigned n, unsigned k) {
    return m != 0 && n != 0 && k != 0 && m <= kMaxM && (uint64_t)k * 16 * 4 * 2 < (1ull << 31) && (uint64_t)k * 64 * 4 < (1ull << 31);
}

// Can this entry run (m, n, k)?  KS has to match the layout K implies, and the
// staged-activation kernels hold at most AM rows.
bool entry_fits(const SolutionEntry &e, unsigned m, unsigned k) {
    return e.shape.ks == span_tiles_for_k(k) && (e.shape.am <= 0 || m <= (unsigned)am_rows(e.shape.am));
}

// Scratch memory.  A call that needs scratch (fp32 slabs of a cross-workgroup K split, the quantised activations of the
// native-FP4 path) takes it, in this order, from
//   1. the per-call workspace handed to petit_gemm_*_ws (caller-owned, stream-ordered by construction: what the Python
//      layer does with torch's caching allocator, and what concurrent streams / graphs must use);
//   2. the workspace registered per device with petit_set_workspace.  One buffer cannot serve two streams at once, so it
//      BINDS to the first stream that uses it; a call from any other stream is refused (PETIT_ERROR_BAD_ARGUMENT) until
//      petit_set_workspace is called again -- never a silent race.
constexpr int kMaxDevices = 64;
constexpr uintptr_t kWorkspaceAlign = 256;
struct Workspace {
    std::atomic<void *> ptr{nullptr};
    std::atomic<uint64_t> bytes{0};
    std::atomic<uintptr_t> stream{kUnbound};
    static constexpr uintptr_t kUnbound = ~(uintptr_t)0;
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
// bytes of scratch a (kernel, split) needs for (m, n, k): [native: quantised activations, 256-B aligned][slabs]
uint64_t workspace_need(const SolutionEntry &e, unsigned splitk, unsigned m, unsigned n, unsigned k, bool have_qa = false) {
    const uint64_t slabs = splitk_bytes(splitk, m, n);
    if (is_native_am(e.shape.am) && !have_qa) // (sized for MXFP8 activations; the MXFP4 form needs less)
        return slabs ? native_ws_aligned(m, k) + slabs : native_ws_bytes(m, k);
    return slabs;
}
// the registered workspace of `dev` for a call on `stream`: pointer, or nullptr (too small / none / *busy = other stream)
void *registered_workspace(int dev, void *stream, uint64_t need, bool *busy) {
    Workspace &ws = g_workspace[dev];
    void *ptr = ws.ptr.load();
    *busy = false;
    if (!ptr || ws.bytes.load() < need)
        return nullptr;
    uintptr_t expect = Workspace::kUnbound;
    if (!ws.stream.compare_exchange_strong(expect, (uintptr_t)stream) && expect != (uintptr_t)stream) {
        *busy = true;
        return nullptr;
    }
    return ptr;
}

// Default choice when the arch table has no entry: pick the shape whose
// workgroup count best fills the chip without starving each wave of work.
// (The reference's heuristic ignores the CU count altogether and leaves half of
// a 256-CU part idle on 4096^2 -- SURVEY.md Appendix C.)
// SiLU-mul epilogue: a wave must hold the gate and the up tile of an output tile -> even n-tiles per wave
bool is_shared(const SolutionEntry &e) { return e.shape.am == kWideAm && e.shape.wm == 5; } // gemm_shared.hpp (plain / bias epilogue only)
bool is_batch(const SolutionEntry &e) { return e.shape.am == 0 && e.shape.wm == 2; }      // gemm_batch.hpp (17 <= M <= 128; reaches the default path through the arch table)
enum : unsigned { kNeedK32 = 1u, kNeedQuantOut = 2u }; // restrictions of the native pipeline (entry_allows)
bool act_ok(const SolutionEntry &e) { return e.shape.nt % 2 == 0 && !is_shared(e); }
// SiLU-mul with this (kernel, K split): unsplit, the kernel's own epilogue does it (gate and up tile in one wave: act_ok); with a cross-workgroup
// K split the slabs hold the plain product and the REDUCE pass applies it (splitk_reduce_silu_kernel) -- any kernel, but a 16-bit output only
bool act_runs(const SolutionEntry &e, unsigned splitk, unsigned restrict_ = 0) {
    return splitk > 1 ? !(restrict_ & kNeedQuantOut) : act_ok(e);
}

// The workgroup tile of a kernel (rows x columns of C), whatever its kind (solution.h: the fields read differently per kind).
void entry_tile(const SolutionEntry &e, unsigned *bm, unsigned *bn) {
    const StreamShape &s = e.shape;
    const bool m32 = s.am == kWideAm || s.am == kNative32Am; // 32-row MFMA blocks: tile_m counts m32-blocks
    *bm = (m32 ? 32u : 16u) * (unsigned)s.mt;
    *bn = 16u * (unsigned)s.wn * (unsigned)s.nt;
}
uint64_t operand_bytes(const SolutionEntry &e, unsigned m, unsigned n, unsigned k) {
    return (uint64_t)n * k / 2 + (uint64_t)n * k / (e.fmt == kFmtNv ? 16 : 32) + 2ull * m * k + 2ull * m * n;
}
// The K split a (kernel, problem) may run with, given the split a table row / the caller's bucket asks for.  A row is measured at ONE M and
// serves a bucket (the last one open-ended): a split that filled the chip at M = 512 is pure overhead at M = 16375 -- the unsplit grid already
// covers the chip several times, and the fp32 slabs (splitk * m * n * 4 bytes, written and read once more by the reduce pass) outgrow the
// operands.  Rules:
//   * no split once the unsplit grid has >= 2 workgroups per CU;
//   * halve the split while its slabs are larger than everything the GEMM reads and writes (W + scales + A + C).
// tune_candidates applies the same rule, so a tuned row never names a split this function would take away at the M it was measured at.
unsigned guarded_splitk(const SolutionEntry &e, unsigned splitk, unsigned m, unsigned n, unsigned k, int num_cus) {
    if (splitk <= 1)
        return splitk;
    unsigned bm, bn;
    entry_tile(e, &bm, &bn);
    const uint64_t tiles = (uint64_t)((m + bm - 1) / bm) * ((n + bn - 1) / bn);
    if (tiles >= 2ull * (unsigned)num_cus)
        return 1;
    const uint64_t cap = operand_bytes(e, m, n, k);
    while (splitk > 1 && splitk_bytes(splitk, m, n) > cap)
        splitk >>= 1;
    return splitk;
}

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
    const double per_weight = (0.5 + 2.0 * s.mt / s.nt) * 1e-7; // unpack once per block + fragment loads per (m-tile, n-tile) pair
    const unsigned blocks = (m + 16 * s.mt - 1) / (16 * s.mt);
    const double wgs = (double)blocks * ((n / kTileN + s.nt * s.wn - 1) / (s.nt * s.wn));
    // VALU-bound: a CU that holds two workgroups takes twice as long, one that holds none idles
    const double rounds = (double)(((unsigned)wgs + num_cus - 1) / num_cus);
    return 2.0 + blocks * (double)n * (double)k * per_weight * rounds * num_cus / wgs;
}
// `splitk` K slices across workgroups (1 = none): each slice walks K / splitk, the grid is splitk times larger, and the fp32
// slabs cost a second launch plus one write and one read of splitk * m * n floats (fitted on the r02 sweeps: sq8192 M = 128
// 128x128 x4 modelled 28.4 us / measured 28.8; down M = 128 128x128 x8 72.7 / 75.3).
// Step costs of the large-M kernels, FITTED (round 4: tools/fit_cost_model.py) on every candidate the in-library tuner timed while the
// built-in table was rebuilt (profiles/r04_table_candidates.csv.gz: 92 shapes x 10 M x 4 families): t1 = the time of one k-tile step of one
// workgroup on a full chip, resident = the workgroups a CU effectively overlaps; median |log error| of the fit 6-9 % per kernel.
struct StepCost {
    int a_type, fmt, kind, tile_m, nt, d, pf, kg;
    float t1, resident, err;
};
const StepCost kStepCost[] = {
#include "cost_gfx950.inc"
};
const StepCost *step_cost(const SolutionEntry &e) {
    const StreamShape &s = e.shape;
    const int kind = s.am == kTiledAm ? 8 : 12, kg = (s.am == kWideAm && s.wm == 3) ? 2 : 1, pf = s.am == kWideAm ? s.pa : 1;
    for (const StepCost &c : kStepCost)
        if (c.a_type == e.a_type && c.fmt == e.fmt && c.kind == kind && c.tile_m == s.mt && c.nt == s.nt && c.d == s.d && c.pf == pf && c.kg == kg)
            return &c;
    return nullptr;
}
double tiled_cost_us(const SolutionEntry &e, unsigned m, unsigned n, unsigned k, int num_cus, unsigned splitk = 1) {
    const StreamShape &s = e.shape;
    const bool wide = s.am == kWideAm;
    const unsigned kg = (wide && s.wm == 3) ? 2u : 1u; // K groups inside the workgroup
    const unsigned bm = (wide ? 32u : 16u) * s.mt, per_wg = s.nt * s.wn;
    const double wgs = (double)((m + bm - 1) / bm) * (double)((n / kTileN + per_wg - 1) / per_wg) * splitk;
    const unsigned ks = span_tiles_for_k(k), nspans = k / (kTileK * ks), parts = splitk * kg;
    const double steps = (double)((nspans + parts - 1) / parts) * ks; // k-tiles the longest slice walks
    const double reduce = splitk > 1 ? 1.5 + (double)splitk * m * n * 8.0 / 5e6 : 0.0;
    if (const StepCost *c = step_cost(e)) {
        // rounds: dispatch is dynamic, so a grid a little over a whole number of rounds pays for part of the next round only when many
        // rounds average it out; half way between the two readings fits the data best
        const double r = wgs / (num_cus * (double)c->resident), rounds = r <= 1.0 ? 1.0 : 0.5 * (r + (double)(unsigned long)(r + 0.999999));
        return 2.0 + steps * c->t1 * rounds + reduce;
    }
    // a kernel without a fitted row (a shape added after the last fit): the round-2 hand fit
    const int acc = s.mt * s.nt; // accumulator tiles per wave: 8 = 64x128 / 128x64, 16 = 64x256 / 128x128
    double t1 = s.mt == 4 && s.nt == 2 ? 0.71 : s.mt == 4 && s.nt == 4 ? 1.31 : s.mt == 8 && s.nt == 2 ? 1.135
              : s.mt == 8 && s.nt == 1 ? 0.94 : s.mt == 1 && s.nt == 4 ? 0.80 : s.mt == 2 && s.nt == 4 ? 0.97
              : s.mt == 4 && s.nt == 5 ? 1.43 : s.mt == 8 && s.nt == 4 ? 1.92 : 0.09 * acc + 0.2;
    if (wide)
        t1 *= 2.0 * (kg == 2 ? 1.8 : 1.0);
    if (e.fmt == kFmtMx)
        t1 *= 0.75; // no group-scale multiplies in the unpack
    const double resident = (acc <= 8 || (s.mt == 8 && s.nt == 2)) ? 1.14 : 1.0;
    double rounds = wgs / (num_cus * resident);
    if (rounds < 1.0)
        rounds = 1.0;
    return 2.0 + steps * t1 * rounds + reduce;
}

// *splitk_out (when given): the heuristic may answer with a K split across workgroups for the tiled kernels (needs scratch:
// callers without any pass nullptr and get the best kernel that needs none).
const SolutionEntry *heuristic(const Family &fam, unsigned m, unsigned n, unsigned k, bool need_pairs = false,
                               unsigned *splitk_out = nullptr, bool need_grouped = false) {
    if (splitk_out)
        *splitk_out = 1;
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
    if (m > 16 && need_grouped)
        return nullptr; // (grouped launches exist for the decode regime)
    if (m > 16) {
        const SolutionEntry *best = nullptr;
        double best_us = 1e30;
        for (int i = 0; i < fam.count; ++i) {
            const SolutionEntry &e = fam.entries[i];
            const StreamShape &s = e.shape;
            if (!entry_fits(e, m, k) || is_native_am(s.am) || s.am == kWideAm || (need_pairs && !act_ok(e)))
                continue; // (never the native-FP4 kernels: different accuracy class; the 32x32 kernels come from the arch table: the
                          //  cost model is good to ~10 % per kernel, and an argmin over twice the candidates loses more to that noise than it gains)
            double us;
            unsigned sk = 1;
            if (s.am == kTiledAm) { // (the 32x32 kernels were skipped above: they reach the default path through the arch table and its neighbours)
                us = tiled_cost_us(e, m, n, k, arch.num_cus);
                if (splitk_out) { // (SiLU-mul too: the reduce pass applies it) K-heavy / narrow problems leave most CUs idle without a K split
                    for (unsigned cand = 2; cand <= 8 && cand <= nspans; cand *= 2) {
                        const double c = tiled_cost_us(e, m, n, k, arch.num_cus, cand);
                        if (c < us)
                            us = c, sk = cand;
                    }
                }
                us -= 0.001 * s.d; // deeper ring on a tie
            } else {
                if (s.am != 0 || s.wn != 1 || s.wm != 1)
                    continue;
                us = stream_cost_us(e, m, n, k, arch.num_cus);
                // the swept winners: WK = 4, fragments requested 2 tiles ahead
                us *= 1.0 + 0.05 * (s.wk != 4) + 0.02 * (s.pa != 2);
                if (nspans < (unsigned)s.wk)
                    us *= (double)s.wk / nspans; // idle K waves
            }
            if (us < best_us) {
                best_us = us, best = &e;
                if (splitk_out)
                    *splitk_out = sk;
            }
        }
        if (best)
            return best;
    }
    if (m > 8 && ntiles >= 12u * arch.num_cus && !need_grouped) {
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
    // workgroup width: the widest of 16 / 32 / 64 columns that still leaves >= ~0.6 workgroups per CU (every swept winner at M = 5..16:
    // N = 4096 -> 16, 6144..8192 -> 32, 10240..28672 -> 64 columns; a wider tile shares the activation block among more columns)
    int want_nt = m <= 2 ? 1 : 4u * ntiles >= 5u * arch.num_cus ? 2 : 1;
    if (mid) {
        // ... refined in round 3: the width whose grid fills the most of the chip's workgroup slots, rounds counted whole, wider on a tie.  Reproduces
        // every pick of the rule above on the swept shapes and adds 48 columns for N = 10240 (214 workgroups instead of 160: qkv M = 16 13.8 -> 12.1 us)
        // and 112 / 224 for N = 28672 / 57344 (256 workgroups).
        double best_fill = 0.0;
        for (const int nt : {1, 2, 3, 4, 7}) {
            const unsigned wgs = (ntiles + nt - 1) / nt, rounds = (wgs + arch.num_cus - 1) / arch.num_cus;
            const double f = (double)wgs / ((double)rounds * arch.num_cus);
            if (f >= best_fill - 1e-9)
                best_fill = f > best_fill ? f : best_fill, want_nt = nt;
        }
    }
    const double target_waves = (double)arch.num_cus * (mid ? 4 : m > 2 ? 8 : 16);
    const SolutionEntry *best = nullptr;
    double best_score = -1e30;
    for (int i = 0; i < fam.count; ++i) {
        const SolutionEntry &e = fam.entries[i];
        if (!entry_fits(e, m, k) || (need_pairs && !act_ok(e)) || (need_grouped && !e.launch_grouped))
            continue;
        const StreamShape &s = e.shape;
        if (s.mt != want_mt || s.am == kTiledAm || is_native_am(s.am) || s.am == kWideAm || s.wm != 1)
            continue; // (the shared-activation-tile kernels, wm = 2, come from the arch table only)
        const unsigned wgs = (ntiles + s.wn * s.nt - 1) / (s.wn * s.nt);
        const unsigned busy_wk = nspans < (unsigned)s.wk ? nspans : (unsigned)s.wk;
        const double busy = (double)wgs * s.wn * busy_wk;
        double score = 0.0;
        // the smallest staged activation block that holds M; a LARGER staged block (the only one some span sizes have) is still
        // far better than fragment loads straight from L2 (5120 x 13824, KS = 4, M = 8: 20.6 us direct against ~13 staged)
        score -= am_rows(s.am) == want_am ? 0.0 : (am_rows(s.am) >= (int)m ? 1.0 : 4.0);
        score += 0.5 * (s.am >= kBfpAm); // bf16 x NVFP4, M <= 4: the fp16 pipeline unpacks cheaper
        // NVFP4, M <= 4: scale applied after the MFMA, cheaper still (gemm_decode.hpp); its 8-row form pays for bf16 only
        score += 0.5 * (s.am >= kDecodeAm && (am_rows(s.am) <= 4 || e.a_type == kDataTypeBf16));
        score -= 2.0 * (s.am >= kDecodeAm && am_rows(s.am) == 8 && e.a_type != kDataTypeBf16);
        score -= 1.0 * (s.nt != want_nt);
        // wave count: under-filling costs more than over-filling
        score -= busy < target_waves ? 3.0 * (1.0 - busy / target_waves) : 0.25 * (busy / target_waves - 1.0);
        // spans must divide evenly over the K waves, or some waves idle in the tail
        const unsigned per = (nspans + s.wk - 1) / s.wk;
        score -= 2.0 * (1.0 - (double)nspans / ((double)per * s.wk));
        // weight tiles in flight per wave: eight in every swept winner (NT x D = 1 x 8, 2 x 4, 4 x 2); deeper rings measured
        // ~1 us SLOWER at M = 8 / 16, and an unseen shape picked one on the old "deeper on a tie" rule
        // (12288 x 4096, M = 16: 11.4 us against 9.1)
        score -= 0.3 * ((s.nt * s.d > 8) ? 1.0 : 0.0) + 0.05 * ((s.nt * s.d < 8) ? 1.0 : 0.0);
        if (score > best_score)
            best_score = score, best = &e;
    }
    if (!best) { // relax the m-tile preference
        for (int i = 0; i < fam.count; ++i)
            if (entry_fits(fam.entries[i], m, k) && (!need_grouped || fam.entries[i].launch_grouped) && fam.entries[i].shape.am != kTiledAm && !is_batch(fam.entries[i]) &&
                !is_native_am(fam.entries[i].shape.am) && fam.entries[i].shape.am != kWideAm && (!need_pairs || act_ok(fam.entries[i])) &&
                (!best || fam.entries[i].shape.mt > best->shape.mt))
                best = &fam.entries[i];
    }
    return best;
}

unsigned entry_mfma(const Family &fam, const SolutionEntry &e) {
    if (e.shape.am == kNative32Am && e.shape.pa == 2) // activations quantised to MXFP4
        return e.a_type == kDataTypeFp16 ? kMfmaFp4ActFp16 : kMfmaFp4;
    if (e.shape.am == kNative32Am && e.shape.pa == 4) // activations quantised to MXFP6
        return e.a_type == kDataTypeFp16 ? kMfmaFp6ActFp16 : kMfmaFp6;
    if (is_native_am(e.shape.am))
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

// --- the opt-in native class: PETIT_SOLUTION_AUTO_NATIVE_MXFP8 / _MXFP4 ------------------------------------------------------
// Default pick inside the native-FP4 class (MXFP4 weights only), for callers that have opted into its accuracy by naming one of
// the two sentinels: arch table of the class first (tuned_native_gfx950.inc / tune-file rows that name a native kernel), else a
// small model: rounds the grid needs on the chip x time of one workgroup at the throughput its tile shape sustained on MI355X
// (bench cells of rounds 2-3: FP4 x FP4 128x256 with two workgroups per CU 3.3 PFLOP/s, 128x128 2.5; FP4 x FP8 64x256 2.3).
enum : int { kClassExact = 0, kClassNativeFp8 = 8, kClassNativeFp6 = 6, kClassNativeFp4 = 4 };
int entry_class(const SolutionEntry &e) {
    if (!is_native_am(e.shape.am))
        return kClassExact;
    if (e.shape.am == kNative32Am && e.shape.pa == 4)
        return kClassNativeFp6;
    return (e.shape.am == kNative32Am && e.shape.pa == 2) ? kClassNativeFp4 : kClassNativeFp8;
}
// restrictions the native pipeline puts on the kernel: bit 0 = pre-quantised activations (the 32x32x64 kernels' layout: kind 13
// only), bit 1 = quantising SiLU-mul epilogue (kind 13 with 128 x 256 workgroup tiles, four waves, no K split)
bool entry_allows(const SolutionEntry &e, unsigned restrict_) {
    const StreamShape &s = e.shape;
    if ((restrict_ & (kNeedK32 | kNeedQuantOut)) && s.am != kNative32Am)
        return false;
    if ((restrict_ & kNeedQuantOut) && !(s.nt == 4 && s.wn == 4 && s.wm == 1))
        return false;
    return true;
}
const SolutionEntry *heuristic_native(const Family &fam, int klass, unsigned m, unsigned n, unsigned k, bool need_pairs, bool have_slabs,
                                      unsigned *splitk_out, unsigned restrict_ = 0) {
    const ArchInfo &arch = arch_info(current_device());
    const unsigned nspans = k / (kTileK * span_tiles_for_k(k));
    const SolutionEntry *best = nullptr;
    double best_us = 1e30;
    *splitk_out = 1;
    for (int i = 0; i < fam.count; ++i) {
        const SolutionEntry &e = fam.entries[i];
        const StreamShape &s = e.shape;
        if (entry_class(e) != klass || !entry_fits(e, m, k) || s.wm == 2 || (need_pairs && !act_ok(e)) || !entry_allows(e, restrict_))
            continue; // (wm = 2: two waves along M, a measured loser kept as a tested instance; wm = 3: two K groups)
        const bool k32 = s.am == kNative32Am;
        const unsigned bm = (k32 ? 32u : 16u) * s.mt, bn = 16u * s.wn * s.nt;
        const bool fp4_rate = klass == kClassNativeFp4 || klass == kClassNativeFp6; // (e2m3 activations run at the e2m1 rate)
        const bool two = k32 && fp4_rate && s.mt * s.nt == 16 && s.d == 2; // Native32Cfg::kMinWavesPerSimd
        double tflops; // sustained by this tile shape when the chip is full
        if (k32 && fp4_rate)
            tflops = two ? 3300.0 : (s.mt * s.nt == 16 ? 2300.0 : 2500.0) + 50.0 * ((s.wk / 4 == 2) + (s.wk % 4 == 2)) + (s.wm == 3 ? 300.0 : 0.0);
        else if (k32)
            tflops = 2000.0;
        else
            tflops = (s.mt == 4 && s.nt >= 4) ? 2300.0 : 1900.0;
        const double wgs = (double)((m + bm - 1) / bm) * (double)((n + bn - 1) / bn);
        const double slots = (double)arch.num_cus * (two ? 2 : 1);
        for (unsigned sk = 1; sk <= 4 && sk <= nspans; sk *= 2) {
            if (sk > 1 && (!have_slabs || (restrict_ & kNeedQuantOut)))
                break;
            const double rounds = (double)(unsigned long)((wgs * sk + slots - 1) / slots);
            const double t_wg = 2.0 * bm * bn * ((double)k / sk) / (tflops * 1e6 / slots); // us: the workgroup's share of the chip rate
            const double us = 6.0 + rounds * t_wg + (sk > 1 ? 1.5 + (double)sk * m * n * 8.0 / 5e6 : 0.0);
            if (us < best_us)
                best_us = us, best = &e, *splitk_out = sk;
        }
    }
    return best;
}

// What solution_id = -1 resolves to for (device, dtypes, act, m, n, k): arch table first, heuristic second; NEVER a
// native-FP4 kernel (different accuracy class: a tune file that lists one is ignored for AUTO).  The choice is a pure
// function of its key (the tables are immutable after static init), so every thread keeps a small direct-mapped
// cache: the eager decode path pays a hash and a compare per call, not the table scans and the cost model.
struct AutoChoice {
    const SolutionEntry *entry;
    unsigned splitk;
};
// how far (2 |ln n/n'| + |ln k/k'|) a tabulated shape may lie from the problem and still lend it its kernel: a factor of ~2.7 in N or ~7 in K
constexpr double kNearestMaxDistance = 2.0;
bool nearest_disabled() { // $PETIT_AMD_NO_NEAREST=1: unseen shapes go straight to the formula heuristic (tools/check_heuristic.py compares the two)
    static const bool off = [] {
        const char *e = getenv("PETIT_AMD_NO_NEAREST");
        return e && *e && *e != '0';
    }();
    return off;
}
// $PETIT_AMD_NEAREST_K=1: an unseen shape takes the nearest tabulated shape's kernel blindly (round 4's behaviour; tools/check_heuristic.py compares)
int nearest_k() {
    static const int v = [] {
        const char *e = getenv("PETIT_AMD_NEAREST_K");
        const long x = e && *e ? strtol(e, nullptr, 10) : 3;
        return (int)(x < 1 ? 1 : x > 8 ? 8 : x);
    }();
    return v;
}
// a farther neighbour's kernel replaces a nearer one's only when its grid overhead on THIS problem (relative to the overhead it won with at home) is
// this much smaller (0.87 = 1 / 1.15 from a sweep over the held-out shapes, profiles/r05_heuristic.md)
double nearest_switch_gain() { // $PETIT_AMD_NEAREST_GAIN overrides (tools/check_heuristic.py sweeps it)
    static const double v = [] {
        const char *e = getenv("PETIT_AMD_NEAREST_GAIN");
        const double x = e && *e ? strtod(e, nullptr) : 0.0;
        return x > 0.0 && x <= 1.0 ? x : 0.87;
    }();
    return v;
}
// What a (kernel, K split) pays on a problem for not fitting it: the last round of workgroups that fills only part of the chip, the K slices
// that come out uneven, the columns of the last n-tile beyond N.  1.0 = a perfect fit; 0 = a kernel kind this does not describe (the decode /
// streaming kernels: their grids are not tile grids).  A grid below one round is NOT a misfit (the shape is small, whatever the kernel).
// A table row's kernel won at ITS shape with whatever overhead it has there; overhead(new) / overhead(home) says how well that win transfers.
double grid_overhead(const SolutionEntry &e, unsigned splitk, unsigned m, unsigned n, unsigned k, int num_cus) {
    const StreamShape &s = e.shape;
    const bool tiled = s.am == kTiledAm, wide = s.am == kWideAm && !is_shared(e), batch = is_batch(e);
    if (m <= 8) {
        // M <= 8 is one m-block and bandwidth-bound whatever the kernel kind (decode / streaming / shared-tile: a workgroup owns 16 nt wn columns; the tile
        // kinds: their BN), several workgroups share a CU: what does not transfer from a neighbour is how evenly the workgroups spread over the CUs.
        // (Three held-out logs: p90 1.08 -> 1.02 at M = 2, 1.11 -> 1.09 at 3-4, 1.18 -> 1.17 at 5-8; at 9-16 the same rule left p90 where it was and made
        // one case worse -- there the activation block starts to weigh and balance alone does not rank: the nearest row is taken as before.)
        if (is_shared(e) || is_native_am(s.am) || s.nt <= 0 || s.wn <= 0)
            return 0.0;
        const unsigned cols = 16u * (unsigned)s.nt * (unsigned)s.wn;
        const double r = (double)((n + cols - 1) / cols) * std::max(1u, splitk) / num_cus;
        return std::ceil(r - 1e-9) / r;
    }
    if (!tiled && !wide && !batch)
        return 0.0;
    unsigned bm, bn;
    entry_tile(e, &bm, &bn);
    const unsigned kp = batch ? (unsigned)s.wk : (wide && s.wm == 3) ? 2u : 1u; // K parts inside the workgroup
    const unsigned nspans = k / (kTileK * span_tiles_for_k(k));
    if (nspans == 0 || bm == 0 || bn == 0)
        return 0.0;
    const unsigned sk = std::max(1u, std::min(splitk, nspans >= kp ? nspans / kp : 1u)); // (the launchers drop empty slices)
    const unsigned parts = std::min(sk * kp, nspans);
    const StepCost *c = step_cost(e);
    const double resident = c ? (double)c->resident : 1.0;
    const double ntiles = (double)((n + bn - 1) / bn);
    const double r = (double)((m + bm - 1) / bm) * ntiles * sk / (num_cus * resident);
    const double rounds = r <= 1.0 ? 1.0 : 0.5 * (r + std::ceil(r - 1e-9)); // (as tiled_cost_us: dispatch is dynamic)
    const double q = rounds / std::max(r, 1.0);
    const double kq = (double)((nspans + parts - 1) / parts) * parts / nspans;
    const double waste = r >= 1.0 ? ntiles * bn / n : 1.0;
    return q * kq * waste;
}
AutoChoice choose_auto(const Family &fam, int dev, int a_type, int b_type, bool act, unsigned m, unsigned n, unsigned k,
                       int klass = kClassExact, unsigned restrict_ = 0) {
    struct Slot {
        uint64_t key0, key1, generation;
        AutoChoice val;
    };
    constexpr int kSlots = 64;
    static thread_local Slot cache[kSlots] = {};
    const uint64_t key0 = ((uint64_t)m << 32) | n;
    const uint64_t key1 = ((uint64_t)k << 32) | ((uint64_t)(restrict_ & 0x3) << 28) | ((uint64_t)(klass & 0xf) << 24) | ((uint64_t)(dev & 0xff) << 16) |
                          ((uint64_t)(a_type & 0xf) << 8) | ((uint64_t)(b_type & 0xf) << 4) | (act ? 2u : 0u) | 1u; // bit 0: slot in use
    const uint64_t generation = tuned_generation(); // bumped by petit_tune_* (hal.hip): run-time rows invalidate cached picks
    Slot &slot = cache[(key0 * 0x9E3779B97F4A7C15ull ^ key1 * 0xC2B2AE3D27D4EB4Full) >> 58];
    if (slot.key0 == key0 && slot.key1 == key1 && slot.generation == generation)
        return slot.val;
    AutoChoice c{nullptr, 1};
    const uint64_t tuned = tuned_solution(dev, a_type, b_type, m, n, k, klass);
    if (tuned) {
        c.entry = find_entry(fam, tuned);
        c.splitk = solution_splitk(tuned);
        if (c.entry && (entry_class(*c.entry) != klass || !entry_fits(*c.entry, m, k) || c.splitk == 0 ||
                        (act && !act_runs(*c.entry, c.splitk, restrict_)) || !entry_allows(*c.entry, restrict_)))
            c.entry = nullptr;
    }
    if (c.entry && klass == kClassExact && m > 512 && c.splitk == 1 && !act) {
        // Prefill at a ragged M.  The bucket's row was measured at ONE M (1024 / 2048 / 8192: whole multiples of every tile height), where its grid fills
        // the chip in whole rounds; at M = 2084 a 128 x 256 tile on N = 8192 needs 544 workgroups = 2.1 rounds of 256 and pays for three (measured: `o`
        // 846 TFLOP/s at M = 2084 between 1071 at 1024 and 982 at 4314).  The shape's rows of the other prefill buckets are measured kernels of this very
        // shape with other tile sizes.  `waste` = (rounds the grid takes x workgroup slots) / workgroups, rounds counted as the fitted cost model counts them
        // (half way between fractional and whole): when the row's kernel wastes > 8 % more here than at the M it was measured at, the sibling row that
        // wastes the least takes over if that is > 8 % less than the row's own.  Never at the measured M itself: a measurement beats this estimate.
        TunedEntry alt[24];
        const int n_alt = tuned_shape_rows(dev, a_type, b_type, n, k, klass, alt, 24);
        const int num_cus = arch_info(dev).num_cus;
        auto waste = [&](const SolutionEntry &e, unsigned mm) {
            unsigned bm, bn;
            entry_tile(e, &bm, &bn);
            const StepCost *sc = step_cost(e);
            const double slots = num_cus * (sc ? (double)sc->resident : 1.0);
            const double tiles = (double)((mm + bm - 1) / bm) * (double)((n + bn - 1) / bn), r = tiles / slots;
            const double rounds = r <= 1.0 ? 1.0 : 0.5 * (r + (double)(unsigned long)(r + 0.999999));
            return rounds * slots * bm * bn / ((double)mm * n); // (work paid for / work asked for: ragged edges count too)
        };
        unsigned hi = 0;
        for (int i = 0; i < n_alt; ++i)
            if (alt[i].solution == tuned && m >= alt[i].m_lo && m <= alt[i].m_hi)
                hi = alt[i].m_hi;
        const unsigned rep = hi == 0 ? m : hi > 4096 ? 8192u : hi == 4096 ? 2048u : hi; // the M the row was measured at (tools/make_tuned_inc.py BUCKET)
        const double own = waste(*c.entry, m);
        if (own > 1.08 * waste(*c.entry, rep)) {
            double best_w = own;
            for (int i = 0; i < n_alt; ++i) {
                if (alt[i].m_hi <= 512 || solution_splitk(alt[i].solution) != 1)
                    continue;
                const SolutionEntry *e = find_entry(fam, alt[i].solution);
                if (!e || !entry_fits(*e, m, k) || is_batch(*e))
                    continue;
                const double w = waste(*e, m);
                if (w < 0.92 * own && w < best_w)
                    best_w = w, c.entry = e;
            }
        }
    }
    if (!c.entry && !nearest_disabled()) {
        // no row for this shape: the rows of the nearest tabulated shapes (hal.h tuned_nearest_list) whose kernels can run this problem.  A
        // neighbour's winner was chosen for how ITS N, K and M fill the chip in whole rounds, which does not transfer (held-out shapes,
        // profiles/r05_heuristic.md: the nearest row taken blindly reads p90 1.2-1.35 at 17 <= M <= 4096, and the best of three neighbours'
        // kernels 1.00-1.14).  So each runnable neighbour gets the ratio grid_overhead(this problem) / grid_overhead(its own shape), and the
        // NEAREST one within 15 % of the best ratio wins: distance still decides between kernels that fit equally well.
        constexpr int kNeighbours = 3;
        TunedNeighbour nb[kNeighbours];
        const int found = tuned_nearest_list(dev, a_type, b_type, m, n, k, klass, kNearestMaxDistance, nb, nearest_k() < kNeighbours ? nearest_k() : kNeighbours);
        const unsigned nspans = k / (kTileK * span_tiles_for_k(k));
        const int num_cus = arch_info(dev).num_cus;
        struct Runnable {
            const SolutionEntry *e;
            unsigned sk;
            double ratio; // 0: unknown
        } run[kNeighbours];
        int n_run = 0;
        for (int i = 0; i < found; ++i) {
            const SolutionEntry *e = find_entry(fam, nb[i].solution);
            const unsigned row_sk = solution_splitk(nb[i].solution);
            if (!e || entry_class(*e) != klass || !entry_fits(*e, m, k) || row_sk == 0 || row_sk > nspans || (act && !act_runs(*e, row_sk, restrict_)) ||
                !entry_allows(*e, restrict_))
                continue;
            const unsigned sk = guarded_splitk(*e, row_sk, m, n, k, num_cus);
            if (act && !act_runs(*e, sk, restrict_))
                continue;
            const double here = klass == kClassExact ? grid_overhead(*e, sk, m, n, k, num_cus) : 0.0;
            const double home = here > 0.0 ? grid_overhead(*e, row_sk, m, nb[i].n, nb[i].k, num_cus) : 0.0;
            run[n_run++] = Runnable{e, sk, home > 0.0 ? here / home : 0.0};
            if (n_run == 1 && (run[0].ratio == 0.0 || nb[i].distance == 0.0))
                break; // nothing to compare (a kernel without a tile grid, the native classes), or not a neighbour at all
        }
        if (n_run) {
            double best_ratio = run[0].ratio;
            for (int i = 1; i < n_run; ++i)
                if (run[i].ratio > 0.0 && run[i].ratio < best_ratio)
                    best_ratio = run[i].ratio;
            // (the bandwidth-bound kernels of M <= 8 lie within a few per cent of each other: a CU imbalance of 8 % already decides)
            const double gain = m <= 8 ? std::max(nearest_switch_gain(), 1.0 / 1.08) : nearest_switch_gain();
            int pick = 0;
            for (int i = 0; i < n_run; ++i)
                if (run[i].ratio > 0.0 && run[i].ratio * gain <= best_ratio) {
                    pick = i;
                    break;
                }
            c.entry = run[pick].e, c.splitk = run[pick].sk;
        }
    }
    if (!c.entry)
        c.entry = klass == kClassExact ? heuristic(fam, m, n, k, act, &c.splitk)
                                       : heuristic_native(fam, klass, m, n, k, act, true, &c.splitk, restrict_);
    if (c.entry && c.splitk > 1) {
        // a row serves a whole M bucket: the split it was measured with is kept only while it still makes sense at THIS m (guarded_splitk)
        const unsigned sk = guarded_splitk(*c.entry, c.splitk, m, n, k, arch_info(dev).num_cus);
        if (sk != c.splitk && act && !act_runs(*c.entry, sk, restrict_)) {
            // SiLU-mul rode on the reduce pass of the split that just went away: a kernel whose own epilogue does it
            unsigned sk2 = 1;
            c.entry = klass == kClassExact ? heuristic(fam, m, n, k, true, nullptr)
                                           : heuristic_native(fam, klass, m, n, k, true, false, &sk2, restrict_);
            c.splitk = 1;
        } else {
            c.splitk = sk;
        }
    }
    slot = Slot{key0, key1, generation, c};
    return c;
}
// NVFP4 weights on the native class run on their MFMA-native image (nvnative.hip, "petit-cdna4-nv6/1").  Call sites that keep calling the reference's
// entry point with (b, scales) name the image by ATTACHING it to the packed weight pointer once at load time (petit_nvfp4_native_attach); the image
// stays the caller's memory.  Looked up only by native-class calls on NVFP4 weights (prefill-sized problems: a mutex and a hash probe).
struct ImageRegistry {
    std::mutex mu;
    std::unordered_map<const void *, const void *> map;
};
ImageRegistry &image_registry() {
    static ImageRegistry r;
    return r;
}
const void *attached_image(const void *b) {
    ImageRegistry &r = image_registry();
    std::lock_guard<std::mutex> lock(r.mu);
    const auto it = r.map.find(b);
    return it == r.map.end() ? nullptr : it->second;
}

int auto_class(uint64_t solution_id) {
    return solution_id == PETIT_SOLUTION_AUTO_NATIVE_MXFP8   ? kClassNativeFp8
           : solution_id == PETIT_SOLUTION_AUTO_NATIVE_MXFP6 ? kClassNativeFp6
           : solution_id == PETIT_SOLUTION_AUTO_NATIVE_MXFP4 ? kClassNativeFp4
                                                             : kClassExact;
}
bool is_auto_id(uint64_t solution_id) { return solution_id == PETIT_SOLUTION_AUTO || auto_class(solution_id) != kClassExact; }

// Prefill at a ragged M, second half: a grid a little over a whole number of rounds (M = 2084 on N = 8192 with 128 x 256 tiles: 544 workgroups = 2.125
// rounds of 256) pays most of a round for its last few tiles.  Rows are independent, so an AUTO call may run as TWO launches on the caller's stream: the
// bulk -- a whole number of m-tiles whose grid ends (nearly) on a round -- with the kernel picked for it, and the remaining rows as a problem of their own
// (a few dozen rows are a batched-decode problem: one more pass over W instead of a round of 128-row tiles).  Returns the bulk's rows, 0 = one launch.
// Estimates, not measurements (times in us): a round costs what the fitted step cost says (else 1 PFLOP/s worth of tiles), rounds are counted as the
// cost model counts them, the tail costs 8 us + max(W at 4.5 TB/s, its FLOPs at 0.8 PFLOP/s); the split must come out > 5 % ahead (measured where it fires: +6 ... +58 %, profiles/r05_row_split_ab.jsonl).
// $PETIT_AMD_NO_ROW_SPLIT=1 turns it off (A/B measurements).  Exact class, default pick only: an explicit id runs as named.
bool row_split_disabled() {
    static const bool off = [] {
        const char *e = getenv("PETIT_AMD_NO_ROW_SPLIT");
        return e && *e && *e != '0';
    }();
    return off;
}
unsigned plan_row_split(const SolutionEntry &e, unsigned splitk, unsigned m, unsigned n, unsigned k, int num_cus) {
    const StreamShape &s = e.shape;
    if (row_split_disabled() || m <= 512 || splitk != 1 || !(s.am == kTiledAm || s.am == kWideAm))
        return 0;
    unsigned bm, bn;
    entry_tile(e, &bm, &bn);
    const StepCost *sc = step_cost(e);
    const double slots = num_cus * (sc ? (double)sc->resident : 1.0);
    const unsigned nx = (n + bn - 1) / bn, ny = (m + bm - 1) / bm;
    const double r = (double)nx * ny / slots;
    if (r <= 1.0 || ny < 2)
        return 0;
    auto rounds = [](double x) { return x <= 1.0 ? 1.0 : 0.5 * (x + std::ceil(x - 1e-9)); };
    const double t_round = sc ? (k / 128.0) * (double)sc->t1 : 2.0 * bm * bn * (double)k * slots / 1.0e9;
    const double whole = rounds(r) * t_round;
    const double w_us = (double)n * k * 0.5625 / 4.5e6;
    double best = whole;
    unsigned best_rows = 0;
    const unsigned span = (unsigned)(slots / nx) + 2; // m-tiles of one round (+ slack): a longer tail is a prefill problem of its own, not a trim
    for (unsigned cut = 1; cut < ny && cut <= span; ++cut) {
        const unsigned ny1 = ny - cut, m1 = ny1 * bm, m2 = m - m1;
        const double tail = 8.0 + std::max(w_us, 2.0 * m2 * (double)n * k / 0.8e9);
        const double cost = rounds((double)nx * ny1 / slots) * t_round + tail;
        if (cost < best)
            best = cost, best_rows = m1;
    }
    return best < 0.95 * whole ? best_rows : 0;
}

// A process-wide opt-in for call sites that cannot name a sentinel (an unchanged SGLang / vLLM layer calls mul_mxfp4_a16(..., -1)):
// $PETIT_AMD_MXFP4_ACTIVATIONS = mxfp8 | mxfp6 | mxfp4, or petit_set_mxfp4_default_class(), makes PETIT_SOLUTION_AUTO on MXFP4 weights
// mean "the default pick of THAT native class" for m >= $PETIT_AMD_NATIVE_MIN_M (default 64: below it the exact kernels are HBM-bound and
// the 128-row native tiles buy nothing) -- whenever the call has the scratch the class needs; without it the exact default runs, as
// before.  Off by default: quantised activations are another accuracy class.
std::atomic<int> g_mxfp4_default_class{-1}; // -1: not read yet
unsigned native_min_m() {
    static const unsigned v = [] {
        const char *e = getenv("PETIT_AMD_NATIVE_MIN_M");
        const long x = e ? strtol(e, nullptr, 10) : 64;
        return (unsigned)(x < 1 ? 1 : x);
    }();
    return v;
}
int mxfp4_default_class() {
    int v = g_mxfp4_default_class.load(std::memory_order_relaxed);
    if (v < 0) {
        const char *e = getenv("PETIT_AMD_MXFP4_ACTIVATIONS");
        v = !e ? 0 : !strcmp(e, "mxfp8") ? kClassNativeFp8 : !strcmp(e, "mxfp6") ? kClassNativeFp6 : !strcmp(e, "mxfp4") ? kClassNativeFp4 : 0;
        g_mxfp4_default_class.store(v, std::memory_order_relaxed);
    }
    return v;
}
// the class PETIT_SOLUTION_AUTO stands for on this problem (kClassExact unless the process opted in, see above)
int auto_default_class(uint64_t solution_id, int b_type, unsigned m) {
    if (solution_id != PETIT_SOLUTION_AUTO || b_type != kDataTypeMxFp4e2m1 || m < native_min_m())
        return kClassExact;
    return mxfp4_default_class();
}

// An explicit id -> table entry.  The element_b nibble is forced to the entry point's format first, as the reference
// does (gemm_fp4_fp16_grid.cc:79-95): ids enumerated with b_type = FP4_E2M1 (what get_fp4_solutions(m, n, k, a, c)
// returns) therefore work with mul_mxfp4_a16; the block-floating-point staged kernels, which only exist for
// bf16 x NVFP4, map to their plain staged twins.
const SolutionEntry *find_explicit(const Family &fam, uint64_t id) {
    id = (id & ~((uint64_t)0xf << 28)) | ((uint64_t)fam.elem_b << 28); // (also: round 3's element nibble 3, "MXFP4 with scales in fp16's range", reads as MXFP4)
    const SolutionEntry *e = find_entry(fam, id);
    const unsigned am = (unsigned)(id >> 48) & 0xf;
    // NVFP4-only kernel kinds named on the MXFP4 entry point: the plain staged kernel with the same geometry
    if (!e && fam.elem_b != kElemBNvFp4 && am >= 5 && am <= 7)
        e = find_entry(fam, (id & ~((uint64_t)0xf << 48)) | ((uint64_t)(am - 4) << 48));
    if (!e && fam.elem_b != kElemBNvFp4 && (am == 4 || am == 14 || am == 15))
        e = find_entry(fam, (id & ~((uint64_t)0xf << 48)) | ((uint64_t)(am == 4 ? 1 : am == 14 ? 2 : 3) << 48));
    return e;
}

// PETIT_DTYPE_MXFP4_E2M1_F16RANGE in hints->b_type (round 3: "every e8m0 block scale lies in 114..140") is accepted and means plain MXFP4:
// the fp16 x MXFP4 kernels test the range themselves (Fp16Mx, device_common.hpp), so the value carries no information any more.
petit_solution_hints effective_hints(const petit_solution_hints *hints) {
    petit_solution_hints h = *hints;
    h.b_type = canonical_b_type(hints->b_type);
    return h;
}

} // namespace

// Candidates of a tuning run (tune.hip): every kernel of the class that can run (m, n, k) within `max_ws` bytes of scratch,
// with the K splits its kind supports.  Also the reference kernel the outputs are compared with (first element): the
// direct-path streaming kernel for the exact class (no staging, no K split: the simplest code path), the first 16x16x128
// (MXFP8) / 32x32x64 (MXFP4) native kernel for the native classes.
int tune_candidates(int a_type, int b_type, int klass, unsigned m, unsigned n, unsigned k, uint64_t max_ws, uint64_t *ids, uint64_t *needs,
                    int cap) {
    Family fam;
    b_type = canonical_b_type(b_type);
    if (!family_for(a_type, b_type, &fam) || !shape_ok(n, k) || m == 0)
        return 0;
    const unsigned nspans = k / (kTileK * span_tiles_for_k(k));
    const int num_cus = arch_info(current_device()).num_cus;
    int count = 0;
    auto push = [&](const SolutionEntry &e, unsigned sk, bool front) {
        const uint64_t need = workspace_need(e, sk, m, n, k);
        if (need > max_ws || count >= cap)
            return;
        const uint64_t id = make_solution_id(e.shape, fam.elem_b, entry_mfma(fam, e), sk);
        if (front && count) {
            ids[count] = ids[0], needs[count] = needs[0];
            ids[0] = id, needs[0] = need;
        } else {
            ids[count] = id, needs[count] = need;
        }
        ++count;
    };
    bool have_ref = false;
    for (int i = 0; i < fam.count; ++i) {
        const SolutionEntry &e = fam.entries[i];
        const StreamShape &s = e.shape;
        if (entry_class(e) != klass || !entry_fits(e, m, k))
            continue;
        const bool is_ref = !have_ref && (klass == kClassExact ? (s.am == 0 && s.wm == 1 && s.pa == 1) : true);
        have_ref |= is_ref;
        // prefill: the streaming kernels re-read W once per 16-64 rows -- tens of milliseconds per launch at M = 8192, never a winner above
        // M = 512 (0 of the 1104 measured rows there) -- so only the one that serves as the reference output is run
        if (!is_ref && m > 512 && s.am >= 0 && !is_batch(e))
            continue;
        // the batched-decode kernels beyond their regime (16-128-row workgroups that each stream their whole column block: not beyond M = 1024 / eight
        // m-blocks) are not candidates at all -- unsplit either (the cap used to sit below the push and only removed their K splits)
        static const unsigned batch_max_m = [] { // $PETIT_AMD_BATCH_MAX_M: experiments with the batched-decode kernels beyond their regime
            const char *v = getenv("PETIT_AMD_BATCH_MAX_M");
            return v && *v ? (unsigned)strtoul(v, nullptr, 10) : 1024u; // (measured: the 128 x 128 form wins `o` / `down` at M = 512 by 3-10 %: profiles/r05_summary.md)
        }();
        if (!is_ref && is_batch(e) && (m > batch_max_m || m > 8u * 16u * (unsigned)s.mt))
            continue;
        push(e, 1, is_ref);
        // K splits: the large-M kernels and the streaming kernels (direct and staged) take any split; the decode / shared-tile
        // kernels none
        // (measured: down 8192 x 28672 at M = 16, staged 16 x 64 tiles with a K split of 2: 29.0 us against 30.5 unsplit -- every CU
        // then pulls half of the activations)
        const bool splittable = s.am == kTiledAm || s.am == kWideAm || is_native_am(s.am) || (s.am >= 0 && s.am < kDecodeAm && s.wm == 1) || is_batch(e);
        if (!splittable)
            continue;
        for (unsigned sk = 2; sk <= 8 && sk <= nspans; sk *= 2)
            if (guarded_splitk(e, sk, m, n, k, num_cus) == sk) // (a row must never name a split that choose_auto would take away again)
                push(e, sk, false);
    }
    return count;
}

// bulk + tail planning (gemm_impl): a dry run walks a call down to its launch -- kernel, split and scratch resolved, every refusal reported -- and stops there
static thread_local bool tl_in_row_split = false, tl_dry_run = false;

int gemm_impl(int b_type, unsigned *c, const unsigned *a, const unsigned *b, const unsigned *scales,
              const float *global_scale, unsigned m, unsigned n, unsigned k,
              const petit_solution_hints *hints, uint64_t solution_id, const petit_epilogue *epilogue,
              void *call_ws, uint64_t call_ws_bytes, void *stream, const NativeIo *io) {
    if (epilogue && ((epilogue->activation != PETIT_ACTIVATION_NONE && epilogue->activation != PETIT_ACTIVATION_SILU_MUL) ||
                     epilogue->reserved != 0))
        return kErrBadArgument; // reject what a newer caller might ask for
    const bool act = epilogue && epilogue->activation == PETIT_ACTIVATION_SILU_MUL;
    const unsigned a_format = io ? io->a_format : 0u, out_format = io ? io->out_format : 0u;
    if ((a_format != 0 && a_format != 8 && a_format != 6 && a_format != 4) || (out_format != 0 && out_format != 8 && out_format != 6 && out_format != 4))
        return kErrBadArgument;
    if (m == 0 || n == 0 || k == 0)
        return kOk; // gemm_fp4_fp16_grid.cc:42-44
    if (!hints || !c || !a || !b || !scales || !global_scale || (!call_ws && call_ws_bytes))
        return kErrBadArgument;
    if ((uintptr_t)call_ws & (kWorkspaceAlign - 1))
        return kErrBadArgument; // f32x4 slabs and 16-byte activation loads: the scratch contract is 256-byte alignment (petit_amd.h)
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
    // M: every kernel addresses A and C per workgroup (64-bit base + a 32-bit offset inside at most 256 rows), so the exact kernels take any M
    // up to the tables' last bucket (prefill chunks of 16375 x 57344 included); beyond it, refuse rather than wrap a grid dimension
    if (m > kMaxM)
        return kErrProblemShape;
    // the native pipeline: pre-quantised activations / quantised SiLU-mul output (MXFP4 weights, 32x32x64 kernels only)
    const unsigned restrict_ = (a_format ? kNeedK32 : 0u) | (out_format ? kNeedQuantOut : 0u);
    if (out_format && !act)
        return kErrBadArgument; // (the quantised output is the SiLU-mul epilogue's)
    if (out_format && (n % 512 != 0 || ((uintptr_t)c & 15)))
        return kErrProblemShape; // the consumer's K = n / 2 must be a whole number of 256-column producer tiles
    if (a_format && ((uintptr_t)a & 15))
        return kErrBadArgument;

    const int dev = current_device();
    const bool is_auto = is_auto_id(solution_id);
    int klass = auto_class(solution_id);
    if (const int dflt = (restrict_ == 0) ? auto_default_class(solution_id, b_type, m) : kClassExact) {
        // the process-wide default class: taken when the scratch of this call (its own, else the registered one) covers the class's pick
        const AutoChoice chn = choose_auto(fam, dev, hints->a_type, b_type, act, m, n, k, dflt, 0);
        if (chn.entry) {
            const uint64_t need_n = workspace_need(*chn.entry, chn.splitk, m, n, k);
            bool busy = false;
            if (call_ws ? call_ws_bytes >= need_n : registered_workspace(dev, stream, need_n, &busy) != nullptr)
                klass = dflt;
        }
    }
    // NVFP4 weights: the native class runs on the weights' MFMA-native image (e4m3 group scales are not E8M0 block scales: nvnative.hip), handed
    // over by petit_gemm_nvfp4_native or attached to `b` beforehand; without one the call is refused, never served by another accuracy class
    const void *nv_image = b_type != kDataTypeFp4e2m1 ? nullptr : (io && io->image) ? io->image : klass != kClassExact ? attached_image(b) : nullptr;
    if (klass != kClassExact && b_type == kDataTypeFp4e2m1 && !nv_image)
        return kErrKernelShape;
    // the native kernels read the quantised activations (k-tile major, up to m * k bytes) through ONE 32-bit buffer descriptor, and the
    // quantiser's grid has one row per activation row
    if ((klass != kClassExact || a_format) && ((uint64_t)m * k >= (1ull << 32) || m > 65535u))
        return kErrProblemShape;
    if (restrict_ && is_auto && klass == kClassExact)
        return kErrKernelShape; // quantised I/O is the native class's: name it (a sentinel or an explicit native id)
    if (a_format && klass != kClassExact && (unsigned)klass != a_format)
        return kErrKernelShape; // activations quantised to one format, kernel class of the other
    const SolutionEntry *entry = nullptr;
    unsigned splitk = 1;
    if (is_auto) {
        // $PETIT_AMD_AUTOTUNE=1: a problem no table knows is tuned once, here, before its first real launch (tune.hip)
        if (klass == kClassExact && !act && autotune_enabled() && tuned_solution(dev, hints->a_type, b_type, m, n, k, kClassExact) == 0) {
            // candidates are limited to the scratch THIS call can use: its own, else the registered workspace if it serves this stream
            void *tws = call_ws;
            uint64_t tws_bytes = call_ws ? call_ws_bytes : 0;
            if (!tws) {
                bool busy = false;
                const uint64_t reg = g_workspace[dev].bytes.load();
                tws = reg ? registered_workspace(dev, stream, reg, &busy) : nullptr;
                tws_bytes = tws ? reg : 0;
            }
            autotune_on_first_sight(b_type, c, a, b, scales, global_scale, m, n, k, hints->a_type, tws, tws_bytes, stream);
        }
        const AutoChoice ch = choose_auto(fam, dev, hints->a_type, b_type, act, m, n, k, klass, restrict_);
        entry = ch.entry, splitk = ch.splitk;
        if (!entry)
            return kErrKernelShape;
        // (io == nullptr: the entry points that take petit_native_args refuse PETIT_SOLUTION_AUTO, so petit_gemm_auto_row_split and
        // petit_gemm_workspace_bytes_ex, which see hints only, describe exactly the calls that get here)
        if (klass == kClassExact && !io && !tl_in_row_split && !autotune_enabled()) {
            if (const unsigned m1 = plan_row_split(*entry, splitk, m, n, k, arch_info(dev).num_cus)) {
                // bulk + tail (plan_row_split): two default-pick calls on row ranges of A and C, same stream, same scratch (the launches are ordered).
                // BOTH are resolved (kernel, split, scratch) before either is launched: a tail that cannot run must not leave C half written or one launch
                // in a stream capture -- the call then runs as the single launch it would have been.
                const size_t c_row = (act ? n / 2 : n) * sizeof(uint16_t), a_row = (size_t)k * sizeof(uint16_t);
                unsigned *const c2 = (unsigned *)((char *)c + m1 * c_row);
                const unsigned *const a2 = (const unsigned *)((const char *)a + m1 * a_row);
                tl_in_row_split = true;
                tl_dry_run = true;
                const bool both = gemm_impl(b_type, c, a, b, scales, global_scale, m1, n, k, hints, solution_id, epilogue, call_ws, call_ws_bytes, stream, io) == kOk &&
                                  gemm_impl(b_type, c2, a2, b, scales, global_scale, m - m1, n, k, hints, solution_id, epilogue, call_ws, call_ws_bytes, stream, io) == kOk;
                tl_dry_run = false;
                int rc = kOk;
                if (both) {
                    rc = gemm_impl(b_type, c, a, b, scales, global_scale, m1, n, k, hints, solution_id, epilogue, call_ws, call_ws_bytes, stream, io);
                    if (rc == kOk)
                        rc = gemm_impl(b_type, c2, a2, b, scales, global_scale, m - m1, n, k, hints, solution_id, epilogue, call_ws, call_ws_bytes, stream, io);
                }
                tl_in_row_split = false;
                if (both)
                    return rc; // (a failure here is a launch error of the device: nothing a different plan would have avoided)
            }
        }
    } else {
        entry = find_explicit(fam, solution_id);
        if (!entry)
            return kErrKernelShape;
        if (!entry_fits(*entry, m, k))
            return kErrProblemShape;
        splitk = solution_splitk(solution_id);
        if (splitk == 0)
            return kErrKernelShape;
        if (act && !act_runs(*entry, splitk, restrict_))
            return kErrKernelShape; // unsplit: needs an even number of n-tiles per wave; split: a 16-bit output (the reduce pass applies SiLU-mul)
        if (!entry_allows(*entry, restrict_) || (a_format && (unsigned)entry_class(*entry) != a_format))
            return kErrKernelShape;
    }

    if (is_native_am(entry->shape.am) && ((uint64_t)m * k >= (1ull << 32) || m > 65535u))
        return kErrProblemShape; // (an explicit native id: the same descriptor range as above)
    if (is_native_am(entry->shape.am) && b_type == kDataTypeFp4e2m1) {
        if (!nv_image)
            nv_image = attached_image(b); // (an explicit native id)
        if (!nv_image)
            return kErrKernelShape; // an explicit native id on NVFP4 weights that have no image attached
        if (nv6_elem_bytes(n, k) >= (1ull << 32))
            return kErrProblemShape; // (the image's element part is read through one 32-bit buffer descriptor)
    }

    GemmArgs args{};
    args.c = c, args.a = a, args.w = b, args.s = scales, args.gs = global_scale;
    if (is_native_am(entry->shape.am) && b_type == kDataTypeFp4e2m1)
        args.w = nv_image, args.s = (const char *)nv_image + nv6_elem_bytes(n, k);
    args.m = m, args.n = n, args.k = k;
    args.bias = epilogue ? epilogue->bias : nullptr;
    args.qa = a_format ? (const void *)a : nullptr, args.qa_format = a_format, args.out_format = out_format;
    const bool have_qa = a_format != 0;
    uint64_t need = workspace_need(*entry, splitk, m, n, k, have_qa);
    if (need) {
        void *ws = nullptr;
        if (call_ws) {
            if (call_ws_bytes < need && !is_auto)
                return kErrBadArgument; // too small for the kernel the caller named
            ws = call_ws_bytes >= need ? call_ws : nullptr;
        } else {
            bool busy = false;
            ws = registered_workspace(dev, stream, need, &busy);
            if (busy && !is_auto)
                return kErrBadArgument; // the registered workspace is bound to another stream: pass one per call
        }
        if (!ws && klass != kClassExact && is_auto && splitk > 1) {
            // native default pick with a K split, scratch (per call or registered) covers the activations only, or they came quantised: the
            // same kernel unsplit -- what petit_gemm_resolve_solution reports for the same arguments.  SiLU-mul rode on the reduce pass of
            // the split (act_runs): unsplit it is the kernel's own epilogue's job, which needs gate and up tile in one wave (act_ok) -- a
            // row like the 64 x 320 kernel (five n-tiles per wave) x split 4 cannot, and returned kOk with C unwritten: re-pick
            if (act && !act_ok(*entry)) {
                unsigned sk1 = 1;
                const SolutionEntry *e1 = heuristic_native(fam, klass, m, n, k, true, /*have_slabs=*/false, &sk1, restrict_);
                if (!e1)
                    return kErrKernelShape;
                entry = e1;
            }
            const uint64_t need1 = workspace_need(*entry, 1, m, n, k, have_qa);
            void *ws1 = nullptr;
            if (!need1) {
                splitk = 1, need = 0;
            } else if (call_ws) {
                ws1 = call_ws_bytes >= need1 ? call_ws : nullptr;
            } else {
                bool busy = false;
                ws1 = registered_workspace(dev, stream, need1, &busy);
            }
            if (ws1)
                splitk = 1, need = need1, ws = ws1;
        }
        if (!ws && need) {
            if (!is_auto || klass != kClassExact)
                return kErrKernelShape; // explicit id (or the native class) that needs scratch nobody provided
            // AUTO without scratch: the best kernel that needs none (not the K-split pick minus its split: a tiled kernel
            // chosen FOR its split leaves most of the chip idle without it)
            entry = heuristic(fam, m, n, k, act);
            splitk = 1;
            if (!entry || workspace_need(*entry, 1, m, n, k))
                return kErrKernelShape; // (unreachable: the heuristic never picks a native kernel)
        }
        args.workspace = (float *)ws;
    }
    // SiLU-mul: in the kernel's epilogue unsplit; by the reduce pass over plain slabs with a cross-workgroup K split
    args.act = (act && splitk == 1) ? 1u : 0u;
    args.reduce_act = (act && splitk > 1) ? 1u : 0u;
    if (tl_dry_run)
        return kOk;
    int rc = entry->launch(args, splitk, (hipStream_t)stream);
    if (rc == kErrSplitCollapsed) {
        // K is too short for the split the id (or the table row) names: the kernel runs as one part, so SiLU-mul is its own epilogue's job
        if (!act_ok(*entry)) {
            if (!is_auto)
                return kErrKernelShape;
            if (klass != kClassExact) { // native class: the class's best kernel whose own epilogue applies SiLU-mul, within the scratch at hand
                unsigned sk1 = 1;
                const SolutionEntry *e1 = heuristic_native(fam, klass, m, n, k, true, /*have_slabs=*/false, &sk1, restrict_);
                if (!e1 || workspace_need(*e1, 1, m, n, k, have_qa) > workspace_need(*entry, splitk, m, n, k, have_qa))
                    return kErrKernelShape;
                entry = e1;
            } else {
                entry = heuristic(fam, m, n, k, true);
                if (!entry || workspace_need(*entry, 1, m, n, k))
                    return kErrKernelShape;
            }
        }
        args.act = 1u, args.reduce_act = 0u;
        rc = entry->launch(args, 1, (hipStream_t)stream);
    }
    return rc;
}

} // namespace petit_amd

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
    ImageRegistry &r = image_registry();
    std::lock_guard<std::mutex> lock(r.mu);
    if (image)
        r.map[b] = image;
    else
        r.map.erase(b);
    return kOk;
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
    const int a_type = (mfma == kMfmaBf16 || mfma == kMfmaFp8 || mfma == kMfmaFp4 || mfma == kMfmaFp6) ? kDataTypeBf16 : kDataTypeFp16;
    const int b_type = (elem_b == kElemBMxFp4 || elem_b == 3u) ? kDataTypeMxFp4e2m1 : kDataTypeFp4e2m1; // (3: round 3's fp16-range nibble)
    Family fam;
    const SolutionEntry *e = family_for(a_type, b_type, &fam) ? find_explicit(fam, id) : nullptr;
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
    if (s.am == kNative32Am) {
        const int wm = s.wm == 2 ? 2 : 1, kgrp = s.wm == 3 ? 2 : 1, lw = s.wm == 4 ? 1 : 0;
        snprintf(buf, len, "native32 %sx%s (activations -> %s) ks%d mb%d np%d waves%dx%d kgroups%d%s d%d kt%d pf%d splitk%u  (wg tile %dx%d, %d threads, 32x32x64 scaled mfma)",
                 a_type == kDataTypeBf16 ? "bf16" : "fp16", b_type == kDataTypeMxFp4e2m1 ? "mxfp4" : "nvfp4-image(e2m3)", s.pa == 2 ? "mxfp4" : s.pa == 4 ? "mxfp6" : "mxfp8", s.ks, s.mt / wm, s.nt / 2, wm, s.wn, kgrp, lw ? " +loader" : "", s.d,
                 s.wk / 4, s.wk % 4, solution_splitk(id), 32 * s.mt, 16 * s.wn * s.nt, 64 * (s.wn * wm * kgrp + lw));
        return kOk;
    }
    if (s.am == kWideAm && s.wm == 5) {
        snprintf(buf, len, "shared32 %sx%s ks%d nb%d splitk%u  (wg tile %dx%d, 256 threads: 4 waves along M, W unpacked once into LDS, 32x32x16 mfma)",
                 a_type == kDataTypeBf16 ? "bf16" : "fp16", b_type == kDataTypeMxFp4e2m1 ? "mxfp4" : "nvfp4", s.ks, s.nt / 2, solution_splitk(id), 32 * s.mt,
                 16 * s.nt);
        return kOk;
    }
    if (s.am == kWideAm) {
        snprintf(buf, len, "wide32 %sx%s ks%d mb%d np%d waves%d kgroups%d d%d pf%d splitk%u  (wg tile %dx%d, %d threads, 32x32x16 mfma%s)",
                 a_type == kDataTypeBf16 ? "bf16" : "fp16", b_type == kDataTypeMxFp4e2m1 ? "mxfp4" : "nvfp4", s.ks,
                 s.mt, s.nt / 2, s.wn, s.wm == 3 ? 2 : 1, s.d, s.pa, solution_splitk(id), 32 * s.mt, 16 * s.wn * s.nt, 64 * s.wn * (s.wm == 3 ? 2 : 1),
                 s.wm == 6 ? ", fragments unpacked one group ahead, accumulators in AGPRs" : "");
        return kOk;
    }
    if (s.am == kTiledAm) {
        snprintf(buf, len, "tiled %sx%s ks%d mt%d ntw%d waves%d d%d splitk%u  (wg tile %dx%d, %d threads)",
                 a_type == kDataTypeBf16 ? "bf16" : "fp16", b_type == kDataTypeMxFp4e2m1 ? "mxfp4" : "nvfp4", s.ks,
                 s.mt, s.nt, s.wn, s.d, solution_splitk(id), 16 * s.mt, 16 * s.wn * s.nt, 64 * s.wn);
        return kOk;
    }
    if (s.am == 0 && s.wm == 2) {
        const int da = s.pa == 2 ? 1 : s.pa == 4 ? 2 : s.pa == 8 ? 4 : 0; // loader wave per K part, activation tiles DA steps ahead (0: none)
        snprintf(buf, len, "batch %sx%s ks%d mt%d nt%d wn%d wk%d d%d da%d splitk%u  (wg tile %dx%d, %d threads: %d K parts reduced in LDS, activation tiles shared by %d waves%s)",
                 a_type == kDataTypeBf16 ? "bf16" : "fp16", b_type == kDataTypeMxFp4e2m1 ? "mxfp4" : "nvfp4", s.ks, s.mt, s.nt, s.wn, s.wk, s.d, da,
                 solution_splitk(id), 16 * s.mt, 16 * s.wn * s.nt, 64 * (s.wn + (da ? 1 : 0)) * s.wk, s.wk, s.wn, da ? ", a loader wave per part" : "");
        return kOk;
    }
    snprintf(buf, len, "stream %sx%s ks%d mt%d nt%d wn%d wk%d d%d am%d splitk%u  (wg tile %dx%d, %d threads)",
             a_type == kDataTypeBf16 ? "bf16" : "fp16", b_type == kDataTypeMxFp4e2m1 ? "mxfp4" : "nvfp4", s.ks,
             s.mt, s.nt, s.wn, s.wk, s.d, am_rows(s.am), solution_splitk(id), 16 * s.mt, 16 * s.wn * s.nt,
             64 * s.wn * s.wk);
    if (s.wm == 2 && s.am < kDecodeAm) // (the 8-row decode kernel also carries warp_partition_m = 2: solution.h)
        strncat(buf, " shared-a", len - strlen(buf) - 1);
    if (s.am >= kDecodeAm)
        strncat(buf, " scale-after-mfma", len - strlen(buf) - 1);
    else if (s.am >= kBfpAm)
        strncat(buf, " bfp16", len - strlen(buf) - 1);
    if (s.pa > 1) {
        char t[16];
        snprintf(t, sizeof(t), " pa%d", s.pa);
        strncat(buf, t, len - strlen(buf) - 1);
    }
    return kOk;
}

} // extern "C"
