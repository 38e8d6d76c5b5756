// tune.hip -- tune-and-persist inside the library: time every kernel that can run a problem ON THE USER'S DEVICE, check each
// candidate's output, record the winner in the run-time arch table (hal.hip) and optionally in a tune file.
//
// Replaces the reference's `bench_matmul -algo tune` (tools/benchmarks/matmul/main.cc:269-325: enumerate GemmGetSolutions,
// time each with hipEvents, print the best id) -- there a benchmark binary whose result the user has to carry back into
// his call sites by hand, here a library entry point (petit_gemm_tune), its Python wrapper (petit_kernel.tune) and an opt-in
// "tune on first sight" mode ($PETIT_AMD_AUTOTUNE=1) that feed PETIT_SOLUTION_AUTO directly; rows persist through
// petit_tune_save / $PETIT_AMD_TUNE_FILE and are loaded on the first call of a later process (hal.hip load_override).
//
// Method (the one tools/benchlib.py uses, in C): every launch of a timed sample reads a DIFFERENT copy of the weights so that
// the 256 MB Infinity Cache cannot serve them (the caller passes the copies, or the tuner clones the caller's one copy
// into a rotation of its own); one untimed sample first; median of `samples` hipEvent-timed samples; a candidate whose first
// sample is 1.5x behind the leader is dropped early.  A candidate is only timed after its output matched the class's
// reference kernel (dispatch.hip tune_candidates) element by element: a kernel that miscomputes at this shape is never ranked.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/petit_amd.h"
#include "layout.h"
#include "hal.h"
#include "petit_internal.h"

namespace petit_amd {
namespace {

// |x - ref| <= tol * max(floor, |ref|) for every element (inf saturated to the largest finite value), NaN must agree; *bad counts violations.  floor = the rms of the reference
// output (tune_problem): two exact kernels differ by f32 summation order and one 16-bit rounding -- a fraction of the VALUE where the value
// is large, of the output's rms where the terms cancel (with max(1, |ref|) as the floor every large-M kernel failed against the streaming
// reference on K = 28672 with MXFP4 block scales up to 2^8: rms ~ 5e4, cancelling elements off by ~1 -- and the tuner crowned a streaming kernel
// five times slower than the default; tools/refresh_table.py found it).
__global__ __launch_bounds__(256) void compare_outputs_kernel(const unsigned short *x, const unsigned short *ref, size_t count, int is_bf16,
                                                              float tol, float floor_, unsigned *bad) {
    unsigned local = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        float a, b;
        if (is_bf16) {
            const unsigned ua = (unsigned)x[i] << 16, ub = (unsigned)ref[i] << 16;
            a = __builtin_bit_cast(float, ua), b = __builtin_bit_cast(float, ub);
        } else {
            const unsigned short ha = x[i], hb = ref[i];
            a = (float)__builtin_bit_cast(_Float16, ha), b = (float)__builtin_bit_cast(_Float16, hb);
        }
        // +-inf saturates to the type's largest finite value before the comparison: an fp16 output near 65504 rounds to inf under one summation order
        // and stays finite under another (both exact kernels), and with "inf must agree" every large-M kernel lost to the streaming reference on
        // fp16 x MXFP4 problems with K >= 12800, whose synthetic outputs reach that range -- 33 prefill rows of the table named a kernel ten times
        // slower than the tiled ones (round 5, found when the rows were re-measured).  NaN must still agree.
        const float top = is_bf16 ? 3.3895314e38f : 65504.0f;
        const bool na = a != a, nb = b != b;
        a = fminf(fmaxf(a, -top), top), b = fminf(fmaxf(b, -top), top);
        if (na != nb || (!na && !(fabsf(a - b) <= tol * fmaxf(floor_, fabsf(b)))))
            ++local;
    }
    if (local)
        atomicAdd(bad, local);
}

// sum of squares (and count) of the finite elements of a 16-bit matrix: out[0] += sum x^2, out[1] += count
__global__ __launch_bounds__(256) void sumsq_kernel(const unsigned short *x, size_t count, int is_bf16, float *out) {
    float sq = 0.f, cnt = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        float a;
        if (is_bf16) {
            const unsigned ua = (unsigned)x[i] << 16;
            a = __builtin_bit_cast(float, ua);
        } else {
            const unsigned short ha = x[i];
            a = (float)__builtin_bit_cast(_Float16, ha);
        }
        if (__builtin_isfinite(a))
            sq += a * a * (1.0f / 1048576.0f), cnt += 1.f; // (scaled: 4 M elements of 6e4 squared stay inside f32)
    }
    for (int off = 32; off; off >>= 1)
        sq += __shfl_xor(sq, off), cnt += __shfl_xor(cnt, off);
    if ((threadIdx.x & 63u) == 0) {
        atomicAdd(out, sq);
        atomicAdd(out + 1, cnt);
    }
}

// A tuning run inside a serving process must not disturb a stream capture in progress on ANOTHER stream (the library can only ask about its
// own).  Measured (tools/probes/capture_legal.hip, ROCm 7.2): while any stream captures in global mode -- what torch.cuda.graph uses -- a
// thread's hipStreamSynchronize / hipEventSynchronize / hipEventQuery on an unrelated stream, hipMalloc, hipFree and hipHostMalloc all fail
// with hipErrorStreamCaptureUnsupported AND invalidate that capture; with the thread's capture mode exchanged to
// hipStreamCaptureModeRelaxed around them they all succeed and the capture survives (only hipMemcpy and hipDeviceSynchronize stay illegal:
// the tuner uses neither).  RelaxedCaptureMode is that exchange, for the duration of a run.
struct RelaxedCaptureMode {
    hipStreamCaptureMode prev = hipStreamCaptureModeRelaxed;
    bool ok;
    RelaxedCaptureMode() { ok = hipThreadExchangeStreamCaptureMode(&prev) == hipSuccess; } // prev: the thread's previous mode
    ~RelaxedCaptureMode() {
        if (ok)
            (void)hipThreadExchangeStreamCaptureMode(&prev);
    }
};

// Device memory of one tuning run.  Two sources:
//  * the pool the caller reserved with petit_tune_reserve (a bump allocator over it): no hipMalloc / hipFree of hundreds of megabytes
//    inside a GEMM call, and it works when the process's allocator already owns the whole device;
//  * hipMalloc for whatever the pool does not cover; everything is freed on every exit path.
struct TunePool {
    std::mutex busy;              // one tuning run at a time per device
    void *base = nullptr;
    uint64_t bytes = 0;
    hipEvent_t e0 = nullptr, e1 = nullptr; // created with the pool, so that a run creates nothing
    unsigned *host = nullptr;     // 4 pinned words: results come back without a pageable staging copy
};
constexpr int kMaxTuneDevices = 64;
TunePool g_pool[kMaxTuneDevices];

struct DeviceBuffers {
    std::vector<void *> ptrs;
    char *pool = nullptr;
    uint64_t pool_left = 0;
    void *alloc(size_t bytes) {
        const uint64_t need = (bytes + 255) & ~(uint64_t)255;
        if (pool && need <= pool_left) {
            void *p = pool;
            pool += need, pool_left -= need;
            return p;
        }
        void *p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        ptrs.push_back(p);
        return p;
    }
    ~DeviceBuffers() {
        for (void *p : ptrs)
            (void)hipFree(p);
    }
};
struct EventPair { // a run's two events: the pool's, or its own (destroyed on every exit path)
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bool own = false;
    bool create() {
        own = true;
        if (hipEventCreate(&e0) != hipSuccess) {
            e0 = nullptr;
            return false;
        }
        if (hipEventCreate(&e1) != hipSuccess) {
            e1 = nullptr;
            return false;
        }
        return true;
    }
    ~EventPair() {
        if (own && e0)
            (void)hipEventDestroy(e0);
        if (own && e1)
            (void)hipEventDestroy(e1);
    }
};
int tune_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxTuneDevices)
        return 0;
    return dev;
}

bool env_on(const char *name) {
    const char *e = getenv(name);
    return e && *e && *e != '0';
}

} // namespace

void tune_bucket(unsigned m, uint64_t solution, unsigned *m_lo, unsigned *m_hi) {
    // the M buckets of the built-in table (tools/make_tuned_inc.py): 1, 2, 3-4, 5-8, 9-16, 17-32, 33-48, 49-64, 65-128, 129-256, 257-512, 513-1024,
    // 1025-4096, 4097+ (round 5: the prefill buckets above 512 are their own rows -- a pick measured at M = 512 says little about M = 16375)
    unsigned lo = 1, hi = 1;
    while (hi < m && hi < 1024)
        lo = hi + 1, hi *= 2;
    if (lo == 33) { // (round 6) 33-48 / 49-64: 48-row tiles (MT = 3 batched-decode instances) serve the lower half
        if (m <= 48)
            hi = 48;
        else
            lo = 49;
    }
    if (m > 4096)
        lo = 4097, hi = kMaxM;
    else if (m > 1024)
        lo = 1025, hi = 4096;
    // a kernel that stages at most AM activation rows cannot serve a bucket that reaches past AM: the row covers m alone
    static const int kRowsOfCode[16] = {0, 1, 2, 4, 1, 1, 2, 4, 0, 0, 8, 16, 0, 0, 2, 4};
    int rows = kRowsOfCode[(solution >> 48) & 0xf];
    if (((solution >> 48) & 0xf) == 15 && ((solution >> 36) & 0xf) == 2)
        rows = 8; // the 8-row decode kernel (solution.h)
    if (rows && (unsigned)rows < hi)
        lo = hi = m;
    *m_lo = lo, *m_hi = hi;
}

int tune_problem(const TuneRequest &rq, uint64_t *best_solution, float *best_us) {
    if (!rq.c || !rq.a || !rq.gs || !rq.b || !rq.s || rq.n_copies == 0 || rq.m == 0)
        return kErrBadArgument;
    hipStream_t stream = (hipStream_t)rq.stream;
    const RelaxedCaptureMode relaxed; // a capture on some OTHER stream survives this run (see above)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone)
        return kErrBadArgument; // tuning synchronises and allocates: never inside a graph capture
    constexpr int kCap = 1024;
    std::vector<uint64_t> ids(kCap), needs(kCap);
    const uint64_t max_ws = rq.own_workspace ? (uint64_t)1 << 30 : rq.ws_bytes;
    const int count = tune_candidates(rq.a_type, rq.b_type, rq.klass, rq.m, rq.n, rq.k, max_ws, ids.data(), needs.data(), kCap);
    if (count == 0)
        return kErrKernelShape;

    // the reserved pool of this device, when there is one and no other run holds it
    TunePool &pool = g_pool[tune_device()];
    std::unique_lock<std::mutex> pool_lock(pool.busy, std::try_to_lock);
    const bool have_pool = pool_lock.owns_lock() && pool.base;
    DeviceBuffers mem;
    if (have_pool)
        mem.pool = (char *)pool.base, mem.pool_left = pool.bytes;
    const size_t out_elems = (size_t)rq.m * rq.n;
    unsigned short *c_ref = (unsigned short *)mem.alloc(out_elems * 2);
    unsigned *bad = (unsigned *)mem.alloc(sizeof(unsigned));
    if (!c_ref || !bad)
        return kErrLaunch;
    void *ws = rq.ws;
    uint64_t ws_bytes = rq.ws_bytes;
    if (rq.own_workspace) {
        uint64_t want = 0;
        for (int i = 0; i < count; ++i)
            want = std::max(want, needs[i]);
        ws = want ? mem.alloc(want) : nullptr;
        ws_bytes = ws ? want : 0;
    }
    // the rotation: the caller's copies, or clones of his single copy up to rotate_bytes (all-or-nothing per clone)
    std::vector<const void *> wb(rq.b, rq.b + rq.n_copies), ws_(rq.s, rq.s + rq.n_copies);
    const size_t w_bytes = (size_t)rq.n * rq.k / 2, s_bytes = (size_t)rq.n * rq.k / (is_mx_type(rq.b_type) ? 32 : 16);
    if (rq.n_copies == 1 && rq.rotate_bytes > w_bytes + s_bytes) {
        const size_t clones = std::min<size_t>(63, rq.rotate_bytes / (w_bytes + s_bytes));
        for (size_t i = 0; i < clones; ++i) {
            void *w2 = mem.alloc(w_bytes), *s2 = w2 ? mem.alloc(s_bytes) : nullptr;
            if (!w2 || !s2)
                break;
            if (hipMemcpyAsync(w2, rq.b[0], w_bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess ||
                hipMemcpyAsync(s2, rq.s[0], s_bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess)
                break;
            wb.push_back(w2), ws_.push_back(s2);
        }
    }
    const size_t copies = wb.size();
    // the native class on NVFP4 weights runs on the weights' MFMA-native image (nvnative.hip): one per copy of the rotation, built here
    std::vector<const void *> images;
    if (rq.klass != 0 && rq.b_type == kDataTypeFp4e2m1) {
        const size_t img_bytes = nv6_image_bytes(rq.n, rq.k);
        for (size_t i = 0; i < copies; ++i) {
            void *img = mem.alloc(img_bytes);
            if (!img || nv6_image(img, wb[i], ws_[i], rq.n, rq.k, stream) != kOk)
                return kErrLaunch;
            images.push_back(img);
        }
    }

    const petit_solution_hints hints{rq.a_type, rq.b_type, rq.a_type, 0};
    auto run = [&](uint64_t id, void *out, size_t copy) {
        const NativeIo io{0u, 0u, images.empty() ? nullptr : images[copy % copies]};
        return gemm_impl(rq.b_type, (unsigned *)out, (const unsigned *)rq.a, (const unsigned *)wb[copy % copies],
                         (const unsigned *)ws_[copy % copies], rq.gs, rq.m, rq.n, rq.k, &hints, id, nullptr, ws, ws_bytes, rq.stream,
                         images.empty() ? nullptr : &io);
    };
    // reference output: candidate 0 (tune_candidates puts the class's reference kernel first), and its rms: the floor of the comparison
    float *stats = (float *)mem.alloc(2 * sizeof(float));
    float stack_stats[2] = {0.f, 0.f};
    unsigned stack_bad = 1;
    // results come back into the pool's pinned words when there is a pool (a copy into pageable memory is staged by the runtime)
    float *host_stats = have_pool && pool.host ? reinterpret_cast<float *>(pool.host) : stack_stats;
    unsigned *host_bad = have_pool && pool.host ? pool.host + 2 : &stack_bad;
    host_stats[0] = host_stats[1] = 0.f;
    if (!stats || run(ids[0], c_ref, 0) != kOk || hipMemsetAsync(stats, 0, 2 * sizeof(float), stream) != hipSuccess)
        return kErrLaunch;
    hipLaunchKernelGGL(sumsq_kernel, dim3(512), dim3(256), 0, stream, c_ref, out_elems, rq.a_type == kDataTypeBf16 ? 1 : 0, stats);
    if (hipMemcpyAsync(host_stats, stats, 2 * sizeof(float), hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess)
        return kErrLaunch;
    const float ref_rms = host_stats[1] > 0.f ? sqrtf(host_stats[0] / host_stats[1]) * 1024.0f : 1.0f;
    const float cmp_floor = ref_rms > 0.f ? ref_rms : 1.0f;

    EventPair ev;
    if (have_pool && pool.e0 && pool.e1)
        ev.e0 = pool.e0, ev.e1 = pool.e1;
    else if (!ev.create())
        return kErrLaunch;
    const hipEvent_t e0 = ev.e0, e1 = ev.e1;
    // $PETIT_AMD_TUNE_LOG=<file>: append "a_type,b_type,klass,m,n,k,solution,us_median,samples" per timed candidate (what tools/build_table.py
    // turns into the heuristic's fit / check data: a full sweep's worth of timings for the price of a tuning run)
    static FILE *const tune_log = [] {
        const char *path = getenv("PETIT_AMD_TUNE_LOG");
        return path && *path ? fopen(path, "a") : nullptr;
    }();
    const float tol = rq.tolerance > 0.f ? rq.tolerance : 2e-2f;
    const unsigned samples = rq.samples ? rq.samples : 5;
    uint64_t best = 0;
    float best_t = 1e30f;
    int rc = kOk;
    size_t rot = 0;
    for (int i = 0; i < count && rc == kOk; ++i) {
        const uint64_t id = ids[i];
        // 1. the output check
        if (hipMemsetAsync(bad, 0, sizeof(unsigned), stream) != hipSuccess || run(id, rq.c, 0) != kOk)
            continue; // (a candidate the launcher refuses -- e.g. a grid limit -- is simply not ranked)
        hipLaunchKernelGGL(compare_outputs_kernel, dim3(512), dim3(256), 0, stream, (const unsigned short *)rq.c, c_ref, out_elems,
                           rq.a_type == kDataTypeBf16 ? 1 : 0, tol, cmp_floor, bad);
        *host_bad = 1;
        if (hipMemcpyAsync(host_bad, bad, sizeof(unsigned), hipMemcpyDeviceToHost, stream) != hipSuccess ||
            hipStreamSynchronize(stream) != hipSuccess) {
            rc = kErrLaunch;
            break;
        }
        if (*host_bad)
            continue;
        // 2. timing: one untimed sample sizes the batch, then `samples` timed ones
        unsigned launches = rq.launches;
        std::vector<float> us;
        for (unsigned sm = 0; sm <= samples; ++sm) {
            const unsigned batch = launches ? launches : (sm == 0 ? 2 : 4); // (the untimed sample only sizes the batch)
            (void)hipEventRecord(e0, stream);
            for (unsigned l = 0; l < batch; ++l)
                if (run(id, rq.c, rot++) != kOk)
                    rc = kErrLaunch;
            (void)hipEventRecord(e1, stream);
            if (hipEventSynchronize(e1) != hipSuccess)
                rc = kErrLaunch;
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            const float per = ms * 1e3f / (float)batch;
            if (rc != kOk)
                break;
            if (sm == 0) { // untimed: choose the batch so that a sample lasts ~0.3 ms (at least 4, at most 256 launches)
                if (!launches) // (2 launches per sample are enough once a launch takes a millisecond: prefill problems)
                    launches = (unsigned)std::min(256.0f, std::max(per >= 1000.0f ? 2.0f : 4.0f, 300.0f / std::max(per, 0.5f)));
                continue;
            }
            us.push_back(per);
            if (sm == 1 && per > 1.5f * best_t)
                break; // hopeless: not worth four more samples
        }
        if (rc != kOk || us.empty())
            continue;
        std::sort(us.begin(), us.end());
        const float med = us[us.size() / 2];
        if (tune_log) // $PETIT_AMD_TUNE_LOG: every candidate that passed its check, with the samples it got (1 = dropped as hopeless)
            fprintf(tune_log, "%d,%d,%d,%u,%u,%u,%llx,%.3f,%zu\n", rq.a_type, rq.b_type, rq.klass, rq.m, rq.n, rq.k, (unsigned long long)id, med, us.size());
        if (us.size() >= samples / 2 + 1 && med < best_t)
            best_t = med, best = id;
    }
    if (tune_log)
        fflush(tune_log);
    if (rc != kOk)
        return rc;
    if (!best)
        return kErrKernelShape;
    if (best_solution)
        *best_solution = best;
    if (best_us)
        *best_us = best_t;
    if (rq.persist) {
        TunedEntry e{};
        e.a_type = rq.a_type, e.b_type = rq.b_type, e.n = rq.n, e.k = rq.k, e.solution = best;
        if (rq.m_lo && rq.m_hi >= rq.m_lo)
            e.m_lo = rq.m_lo, e.m_hi = rq.m_hi;
        else
            tune_bucket(rq.m, best, &e.m_lo, &e.m_hi);
        tuned_insert(e);
    }
    return kOk;
}

// $PETIT_AMD_AUTOTUNE=1: PETIT_SOLUTION_AUTO tunes a problem whose (dtypes, N, K, M bucket) no table knows, once, on first sight, with the
// caller's own buffers (the output is overwritten by candidates and then by the real call) and the scratch of THAT call (so the winner
// is a kernel this caller can run).  Everything else -- the reference output, clones of the caller's weights for the rotation -- comes
// out of the pool reserved with petit_tune_reserve, else from hipMalloc; the run happens with the thread's capture mode relaxed, so a
// capture in progress on another stream survives it (RelaxedCaptureMode above).  Once per key and process for successes and hard failures (a
// sighting while the caller's own stream is capturing does not count; transient launch / allocation failures get a few retries).  The
// row lands in the run-time table and, when $PETIT_AMD_TUNE_FILE is set, in that file (merged with what other processes saved, hal.hip
// tuned_save) for the next process.
bool autotune_enabled() {
    static const bool on = env_on("PETIT_AMD_AUTOTUNE");
    return on;
}
void autotune_on_first_sight(int b_type, unsigned *c, const unsigned *a, const unsigned *b, const unsigned *scales, const float *gs, unsigned m,
                             unsigned n, unsigned k, int a_type, void *ws, uint64_t ws_bytes, void *stream) {
    const int dev = tune_device();
    if (!tuned_lookup_enabled(dev))
        return; // the table is switched off or belongs to another arch: a tuned row would never be looked up
    const void *bp = b, *sp = scales;
    TuneRequest rq{};
    rq.c = c, rq.a = a, rq.b = &bp, rq.s = &sp, rq.n_copies = 1, rq.gs = gs;
    rq.m = m, rq.n = n, rq.k = k, rq.a_type = a_type, rq.b_type = canonical_b_type(b_type), rq.klass = 0;
    rq.own_workspace = false, rq.ws = ws, rq.ws_bytes = ws ? ws_bytes : 0; // the candidates: what THIS caller's scratch can run
    rq.stream = stream, rq.persist = true;
    rq.rotate_bytes = (size_t)384 << 20; // past the 256 MB Infinity Cache (as far as the pool reaches)
    // A key is tuned ONCE per process: a success whose row some later lookup does not find (e.g. a row split by a later insert) is not repeated, and a
    // transient failure (out of memory for the reference output / the rotation, a device error) gets kMaxAttempts tries PER KEY -- not a process-wide
    // budget that one persistently failing key could use up for everybody.
    constexpr int kMaxAttempts = 3;
    struct Key {
        int a_type, b_type;
        unsigned n, k, m_lo;
        int attempts;   // tries so far
        bool done;      // tuned, failed for good, or being tuned right now
    };
    static std::mutex seen_mutex;
    static std::vector<Key> seen;
    unsigned lo, hi;
    tune_bucket(m, 0, &lo, &hi);
    // a first sighting INSIDE a stream capture (serving stacks often meet their M buckets while capturing graphs) cannot tune -- and must not
    // burn the key: the first eager call of the same problem tunes it
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing((hipStream_t)stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        return;
    }
    auto same = [&](const Key &f) { return f.a_type == a_type && f.b_type == rq.b_type && f.n == n && f.k == k && f.m_lo == lo; };
    {
        std::lock_guard<std::mutex> lock(seen_mutex);
        bool found = false;
        for (Key &f : seen)
            if (same(f)) {
                if (f.done)
                    return;
                f.done = true, ++f.attempts, found = true; // (a retry: this thread has it now)
                break;
            }
        if (!found)
            seen.push_back(Key{a_type, rq.b_type, n, k, lo, 1, true});
    }
    uint64_t best = 0;
    float us = 0.f;
    const int rc = tune_problem(rq, &best, &us);
    if (rc == kErrLaunch) {
        std::lock_guard<std::mutex> lock(seen_mutex);
        for (Key &f : seen)
            if (same(f) && f.attempts < kMaxAttempts)
                f.done = false; // the next call of this problem tries again
    }
    if (rc != kOk)
        return;
    const char *path = getenv("PETIT_AMD_TUNE_FILE");
    if (path && *path)
        (void)tuned_save(path);
}

} // namespace petit_amd

using namespace petit_amd;

extern "C" {

int petit_gemm_tune(unsigned *c, const unsigned *a, const float *global_scale, unsigned m, unsigned n, unsigned k,
                    const petit_solution_hints *hints, const petit_tune_params *params, void *workspace, uint64_t workspace_bytes,
                    void *stream, uint64_t *best_solution, float *best_us) {
    if (!hints || !params || params->struct_bytes != sizeof(petit_tune_params) || hints->c_type != hints->a_type)
        return kErrBadArgument;
    if (params->klass != 0 && params->klass != 8 && params->klass != 6 && params->klass != 4)
        return kErrBadArgument;
    if (((uintptr_t)workspace & 255) || (!workspace && workspace_bytes))
        return kErrBadArgument;
    TuneRequest rq{};
    rq.c = c, rq.a = a, rq.b = params->b, rq.s = params->scales, rq.n_copies = params->n_copies, rq.gs = global_scale;
    rq.m = m, rq.n = n, rq.k = k, rq.a_type = hints->a_type, rq.b_type = canonical_b_type(hints->b_type), rq.klass = params->klass;
    rq.ws = workspace, rq.ws_bytes = workspace_bytes, rq.own_workspace = false, rq.stream = stream;
    rq.launches = params->launches, rq.samples = params->samples, rq.tolerance = params->tolerance;
    rq.persist = params->persist != 0, rq.m_lo = params->m_lo, rq.m_hi = params->m_hi;
    rq.rotate_bytes = params->rotate_bytes;
    return tune_problem(rq, best_solution, best_us);
}

int petit_tune_insert(const petit_solution_hints *hints, unsigned n, unsigned k, unsigned m_lo, unsigned m_hi, uint64_t solution) {
    if (!hints || m_lo == 0 || m_hi < m_lo || n == 0 || k == 0 || solution == 0 || solution == PETIT_SOLUTION_AUTO)
        return kErrBadArgument;
    char buf[8];
    if (petit_describe_solution(solution, buf, sizeof(buf)) != kOk)
        return kErrKernelShape; // not a kernel of this build
    TunedEntry e{};
    e.a_type = hints->a_type, e.b_type = hints->b_type, e.n = n, e.k = k, e.m_lo = m_lo, e.m_hi = m_hi, e.solution = solution;
    tuned_insert(e);
    return kOk;
}

int petit_tune_reserve(void *device_ptr, uint64_t bytes) {
    if (((uintptr_t)device_ptr & 255) || (!device_ptr && bytes))
        return kErrBadArgument;
    TunePool &pool = g_pool[tune_device()];
    std::lock_guard<std::mutex> lock(pool.busy); // (waits for a run in progress)
    pool.base = device_ptr, pool.bytes = device_ptr ? bytes : 0;
    if (device_ptr && !pool.e0) { // the per-device helpers of a run, created here so that a run creates nothing
        if (hipEventCreate(&pool.e0) != hipSuccess)
            pool.e0 = nullptr;
        if (hipEventCreate(&pool.e1) != hipSuccess)
            pool.e1 = nullptr;
        void *h = nullptr;
        if (hipHostMalloc(&h, 4 * sizeof(unsigned), hipHostMallocDefault) == hipSuccess)
            pool.host = (unsigned *)h;
        else
            (void)hipGetLastError();
    }
    return kOk;
}

int petit_tune_save(const char *path) { return tuned_save(path) ? kOk : kErrBadArgument; }

uint64_t petit_tune_generation(void) { return tuned_generation(); }

} // extern "C"
