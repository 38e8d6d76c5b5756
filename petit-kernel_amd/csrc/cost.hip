// cost.hip -- the cost model of the large-M kernels (fitted on the in-library tuner's logs: cost_gfx950.inc) and the formula heuristic that picks a kernel
// when neither the arch table nor a tabulated neighbour knows the problem.  Replaces fp4/algo_chooser.cc:64-132 (ChooseDefaultFp4Fp16Solution), which
// ignores the CU count and leaves half of a 256-CU part idle on 4096^2.
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
double tiled_cost_us(const SolutionEntry &e, unsigned m, unsigned n, unsigned k, int num_cus, unsigned splitk) {
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
const SolutionEntry *heuristic(const Family &fam, unsigned m, unsigned n, unsigned k, bool need_pairs, unsigned *splitk_out, bool need_grouped) {
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

} // namespace petit_amd
