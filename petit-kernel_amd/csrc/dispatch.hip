// dispatch.hip -- the dispatcher behind every GEMM entry point (gemm_impl), the scratch-memory rules and the tuner's candidate list.
// Replaces fp4/gemm_fp4_fp16_grid.cc:11-77 (Dispatcher, GemmFp4Fp16GridImpl), which has no scratch, no launch-error check and no fused epilogue.
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

// Scratch memory.  A call that needs scratch (fp32 slabs of a cross-workgroup K split, the quantised activations of the
// native-FP4 path) takes it, in this order, from
//   1. the per-call workspace handed to petit_gemm_*_ws (caller-owned, stream-ordered by construction: what the Python
//      layer does with torch's caching allocator, and what concurrent streams / graphs must use);
//   2. the workspace registered per device with petit_set_workspace.  One buffer cannot serve two streams at once, so it
//      BINDS to the first stream that uses it; a call from any other stream is refused (PETIT_ERROR_BAD_ARGUMENT) until
//      petit_set_workspace is called again -- never a silent race.
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
uint64_t workspace_need(const SolutionEntry &e, unsigned splitk, unsigned m, unsigned n, unsigned k, bool have_qa) {
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

// NVFP4 weights on the native class run on their MFMA-native image (nvnative.hip, "petit-cdna4-nv6/1").  Call sites that keep calling the reference's
// entry point with (b, scales) name the image by ATTACHING it to the packed weight pointer once at load time (petit_nvfp4_native_attach); the image
// stays the caller's memory.  Looked up only by native-class calls on NVFP4 weights (prefill-sized problems: a mutex and a hash probe).
struct ImageRegistry {
    std::mutex mu;
    std::unordered_map<const void *, const void *> map;
};
static ImageRegistry &image_registry() {
    static ImageRegistry r;
    return r;
}
const void *attached_image(const void *b) {
    ImageRegistry &r = image_registry();
    std::lock_guard<std::mutex> lock(r.mu);
    const auto it = r.map.find(b);
    return it == r.map.end() ? nullptr : it->second;
}
int attach_image(const void *b, const void *image) {
    ImageRegistry &r = image_registry();
    std::lock_guard<std::mutex> lock(r.mu);
    if (image)
        r.map[b] = image;
    else
        r.map.erase(b);
    return kOk;
}

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
    // The table's own row for this problem is always a candidate: 110 rows of the 129 ... 4096 buckets name batched-decode kernels beyond the eight-m-block cap above (narrow N --
    // 576 ... 3072 columns -- where W stays in L2 and many small workgroups fill the chip; measured winners of round 5), and a row the tuner cannot time can never be
    // challenged by a newer kernel (tools/adopt_rows.py compares against the row's own time in the same session).
    if (const uint64_t row = tuned_solution(current_device(), a_type, b_type, m, n, k, klass)) {
        bool have = false;
        for (int i = 0; i < count; ++i)
            have |= ids[i] == row;
        const SolutionEntry *e = have ? nullptr : find_explicit(fam, row);
        const unsigned sk = solution_splitk(row);
        if (e && sk && entry_class(*e) == klass && entry_fits(*e, m, k))
            push(*e, sk, false);
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
        // the native class at a ragged prefill M (plan_row_split_native, pick.hip): bulk in the class, the few dozen rows of the tail through the exact default
        if (klass != kClassExact && restrict_ == 0 && !(io && io->image) && !tl_in_row_split && !autotune_enabled()) {
            if (const unsigned m1 = plan_row_split_native(*entry, klass, splitk, m, n, k, arch_info(dev).num_cus)) {
                const size_t c_row = (act ? n / 2 : n) * sizeof(uint16_t), a_row = (size_t)k * sizeof(uint16_t);
                unsigned *const c2 = (unsigned *)((char *)c + m1 * c_row);
                const unsigned *const a2 = (const unsigned *)((const char *)a + m1 * a_row);
                tl_in_row_split = true;
                tl_dry_run = true;
                const bool both = gemm_impl(b_type, c, a, b, scales, global_scale, m1, n, k, hints, solution_id, epilogue, call_ws, call_ws_bytes, stream, io) == kOk &&
                                  gemm_impl(b_type, c2, a2, b, scales, global_scale, m - m1, n, k, hints, PETIT_SOLUTION_AUTO, epilogue, call_ws, call_ws_bytes, stream, nullptr) == kOk;
                tl_dry_run = false;
                int rc = kOk;
                if (both) {
                    rc = gemm_impl(b_type, c, a, b, scales, global_scale, m1, n, k, hints, solution_id, epilogue, call_ws, call_ws_bytes, stream, io);
                    if (rc == kOk)
                        rc = gemm_impl(b_type, c2, a2, b, scales, global_scale, m - m1, n, k, hints, PETIT_SOLUTION_AUTO, epilogue, call_ws, call_ws_bytes, stream, nullptr);
                }
                tl_in_row_split = false;
                if (both)
                    return rc;
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
