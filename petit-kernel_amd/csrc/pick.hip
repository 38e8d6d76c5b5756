// pick.hip -- what PETIT_SOLUTION_AUTO and the native-class sentinels resolve to: the measured arch table first (hal.h), then the rows of the nearest
// tabulated shapes ranked by how their kernels' grids fit THIS problem, then the formula heuristic (cost.hip); sibling rows and the bulk + tail plan for
// ragged prefill M.  Decided once per (thread, problem) and cached.  Replaces fp4/algo_chooser.cc:64-132, which scans a 234-entry map on every call.
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

// (heuristic_native: the opt-in native class -- the arch table of the class first, else a small model)
const SolutionEntry *heuristic_native(const Family &fam, int klass, unsigned m, unsigned n, unsigned k, bool need_pairs, bool have_slabs,
                                      unsigned *splitk_out, unsigned restrict_) {
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
AutoChoice choose_auto(const Family &fam, int dev, int a_type, int b_type, bool act, unsigned m, unsigned n, unsigned k, int klass, unsigned restrict_) {
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
        // (round 6: not when the bulk + tail plan fires for the row's OWN kernel -- its bulk is whole rounds of the measured winner again; a sibling is the
        // second-best kernel of another M.  `o` at M = 4314, NVFP4: the 256 x 256 row -> 4096 rows in two full rounds + 218 rows of batched decode)
        if (own > 1.08 * waste(*c.entry, rep) && plan_row_split(*c.entry, 1, m, n, k, num_cus) == 0) {
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

// The same plan for the NATIVE class (round 6): a sentinel call whose grid of 128-row tiles ends a little past a whole number of rounds
// (`o` at M = 2084: 544 workgroups on 512 slots) runs its first m1 rows -- whole m-tiles, whole rounds -- in the class, and the remaining few dozen rows through
// the EXACT default pick (a batched-decode kernel: one more pass over W; the class has no small-M kernel, a 36-row tail would pay a 128-row tile at a fraction
// of the chip).  The tail rows are computed exactly -- never less accurately than the class promises; the quantised-activation scratch is k-tile major, so each
// part quantises its own rows (stream order lets them share the scratch).  Not with pre-quantised input / quantised output (one layout for all rows), and not
// when the image came per call without the packed tensors (petit_gemm_nvfp4_native).
unsigned plan_row_split_native(const SolutionEntry &e, int klass, unsigned splitk, unsigned m, unsigned n, unsigned k, int num_cus) {
    const StreamShape &s = e.shape;
    if (row_split_disabled() || m <= 512 || splitk != 1 || s.am != kNative32Am)
        return 0;
    unsigned bm, bn;
    entry_tile(e, &bm, &bn);
    const bool two = s.wm == 1 && s.mt * s.nt == 16 && s.d == 2; // Native32Cfg: the 128 x 256 two-tile-ring forms fit two workgroups per CU
    const double slots = num_cus * (two ? 2.0 : 1.0);
    const unsigned nx = (n + bn - 1) / bn, ny = (m + bm - 1) / bm;
    const double r = (double)nx * ny / slots;
    if (r <= 1.0 || ny < 2)
        return 0;
    auto rounds = [](double x) { return x <= 1.0 ? 1.0 : 0.5 * (x + std::ceil(x - 1e-9)); };
    const double tflops = klass == kClassNativeFp8 ? 2300.0 : klass == kClassNativeFp6 ? 2800.0 : 3300.0; // what the class sustains on a full chip (bench cells)
    const double t_round = 2.0 * bm * bn * (double)k * slots / (tflops * 1e6);
    const double whole = rounds(r) * t_round;
    const double w_us = (double)n * k * 0.5625 / 4.5e6;
    double best = whole;
    unsigned best_rows = 0;
    const unsigned span = (unsigned)(slots / nx) + 2;
    for (unsigned cut = 1; cut < ny && cut <= span; ++cut) {
        const unsigned ny1 = ny - cut, m1 = ny1 * bm, m2 = m - m1;
        if (m2 > 128)
            break; // (a longer tail is a compute-bound problem of its own: it stays in the class.  Measured, profiles/r06_native_row_split_ab.jsonl: tails of 36 rows
                   // at M = 2084 gain 6-25 % on `o` / `down`, both weight formats, all three activation formats; tails of 218 rows at M = 4314 -3 ... +1 %)
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

} // namespace petit_amd
