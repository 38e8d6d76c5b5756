// dispatch.h -- declarations shared by the host-side translation units of libpetit_amd.so (round 6: csrc/api.hip, 1 650 lines, became five):
//   solutions.hip  the family tables, ids <-> table entries, what an entry can run
//   cost.hip       the cost model and the formula heuristic behind the arch tables
//   pick.hip       what PETIT_SOLUTION_AUTO (and the native-class sentinels) resolve to: arch table, neighbours, row split
//   dispatch.hip   scratch memory, the tuner's candidate list, gemm_impl -- the dispatcher behind every GEMM entry point
//   api.hip        the C ABI of include/petit_amd.h;  describe.hip: petit_describe_solution / error strings
// Not installed; the public surface is include/petit_amd.h.
#pragma once

#include <atomic>
#include <cstdint>

#include "../../include/petit_amd.h"
#include "gemm_native32.hpp"
#include "gemm_stream.hpp"
#include "hal.h"
#include "layout.h"
#include "petit_internal.h"
#include "solution.h"

namespace petit_amd {

// --- solutions.hip
struct Family {
    const SolutionEntry *entries;
    int count;
    unsigned elem_b, mfma;
};
bool family_for(int a_type, int b_type, Family *out);
bool shape_ok(unsigned n, unsigned k);
bool problem_in_range(unsigned m, unsigned n, unsigned k);
bool entry_fits(const SolutionEntry &e, unsigned m, unsigned k);
unsigned entry_mfma(const Family &fam, const SolutionEntry &e);
uint64_t entry_id(const Family &fam, const SolutionEntry &e);
const SolutionEntry *find_entry(const Family &fam, uint64_t id);
const SolutionEntry *find_explicit(const Family &fam, uint64_t id);
petit_solution_hints effective_hints(const petit_solution_hints *hints);
extern std::atomic<int> g_native_enabled; // -1: $PETIT_AMD_NATIVE_FP4 not read yet
bool native_enabled();
// the accuracy classes: exact, or the native block-scaled MFMA with activations quantised to MXFP8 / MXFP6 / MXFP4
enum : int { kClassExact = 0, kClassNativeFp8 = 8, kClassNativeFp6 = 6, kClassNativeFp4 = 4 };
int entry_class(const SolutionEntry &e);
enum : unsigned { kNeedK32 = 1u, kNeedQuantOut = 2u }; // restrictions of the native pipeline (entry_allows)
bool entry_allows(const SolutionEntry &e, unsigned restrict_);
inline bool is_shared(const SolutionEntry &e) { return e.shape.am == kWideAm && e.shape.wm == 5; } // gemm_shared.hpp (plain / bias epilogue only)
inline bool is_batch(const SolutionEntry &e) { return e.shape.am == 0 && e.shape.wm == 2; }      // gemm_batch.hpp (17 <= M <= 128; reaches the default path through the arch table)
// SiLU-mul epilogue: a wave must hold the gate and the up tile of an output tile -> even n-tiles per wave
inline bool act_ok(const SolutionEntry &e) { return e.shape.nt % 2 == 0 && !is_shared(e); }
bool act_runs(const SolutionEntry &e, unsigned splitk, unsigned restrict_ = 0);
void entry_tile(const SolutionEntry &e, unsigned *bm, unsigned *bn);
uint64_t operand_bytes(const SolutionEntry &e, unsigned m, unsigned n, unsigned k);

// --- cost.hip
struct StepCost {
    int a_type, fmt, kind, tile_m, nt, d, pf, kg;
    float t1, resident, err;
};
const StepCost *step_cost(const SolutionEntry &e);
unsigned guarded_splitk(const SolutionEntry &e, unsigned splitk, unsigned m, unsigned n, unsigned k, int num_cus);
double stream_cost_us(const SolutionEntry &e, unsigned m, unsigned n, unsigned k, int num_cus);
double tiled_cost_us(const SolutionEntry &e, unsigned m, unsigned n, unsigned k, int num_cus, unsigned splitk = 1);
const SolutionEntry *heuristic(const Family &fam, unsigned m, unsigned n, unsigned k, bool need_pairs = false, unsigned *splitk_out = nullptr,
                               bool need_grouped = false);

// --- pick.hip
const SolutionEntry *heuristic_native(const Family &fam, int klass, unsigned m, unsigned n, unsigned k, bool need_pairs, bool have_slabs,
                                      unsigned *splitk_out, unsigned restrict_ = 0);
struct AutoChoice {
    const SolutionEntry *entry;
    unsigned splitk;
};
double grid_overhead(const SolutionEntry &e, unsigned splitk, unsigned m, unsigned n, unsigned k, int num_cus);
AutoChoice choose_auto(const Family &fam, int dev, int a_type, int b_type, bool act, unsigned m, unsigned n, unsigned k, int klass = kClassExact,
                       unsigned restrict_ = 0);
int auto_class(uint64_t solution_id);
bool is_auto_id(uint64_t solution_id);
unsigned plan_row_split(const SolutionEntry &e, unsigned splitk, unsigned m, unsigned n, unsigned k, int num_cus);
unsigned plan_row_split_native(const SolutionEntry &e, int klass, unsigned splitk, unsigned m, unsigned n, unsigned k, int num_cus);
extern std::atomic<int> g_mxfp4_default_class; // -1: $PETIT_AMD_MXFP4_ACTIVATIONS not read yet
int mxfp4_default_class();
int auto_default_class(uint64_t solution_id, int b_type, unsigned m);

// --- dispatch.hip
constexpr int kMaxDevices = 64;
constexpr uintptr_t kWorkspaceAlign = 256;
struct Workspace {
    std::atomic<void *> ptr{nullptr};
    std::atomic<uint64_t> bytes{0};
    std::atomic<uintptr_t> stream{kUnbound};
    static constexpr uintptr_t kUnbound = ~(uintptr_t)0;
};
extern Workspace g_workspace[kMaxDevices];
int current_device();
uint64_t splitk_bytes(unsigned splitk, unsigned m, unsigned n);
uint64_t workspace_need(const SolutionEntry &e, unsigned splitk, unsigned m, unsigned n, unsigned k, bool have_qa = false);
void *registered_workspace(int dev, void *stream, uint64_t need, bool *busy);
// the MFMA-native images attached to packed NVFP4 weight pointers (petit_nvfp4_native_attach)
int attach_image(const void *b, const void *image);
const void *attached_image(const void *b);

} // namespace petit_amd
