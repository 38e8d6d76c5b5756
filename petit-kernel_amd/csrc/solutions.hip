// solutions.hip -- the kernel tables of libpetit_amd.so seen from the host: the four (activation type, weight format) families, ids <-> table entries,
// what an entry can run.  Replaces fp4/solution_map.cc, fp4/gen_solution_list.cc (the reference's build-time kernel list) and the id handling of
// fp4/gemm_fp4_fp16_grid.cc:79-95.
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

// Default choice when the arch table has no entry: pick the shape whose
// workgroup count best fills the chip without starving each wave of work.
// (The reference's heuristic ignores the CU count altogether and leaves half of
// a 256-CU part idle on 4096^2 -- SURVEY.md Appendix C.)
// SiLU-mul epilogue: a wave must hold the gate and the up tile of an output tile -> even n-tiles per wave
// SiLU-mul with this (kernel, K split): unsplit, the kernel's own epilogue does it (gate and up tile in one wave: act_ok); with a cross-workgroup
// K split the slabs hold the plain product and the REDUCE pass applies it (splitk_reduce_silu_kernel) -- any kernel, but a 16-bit output only
bool act_runs(const SolutionEntry &e, unsigned splitk, unsigned restrict_) {
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

} // namespace petit_amd
