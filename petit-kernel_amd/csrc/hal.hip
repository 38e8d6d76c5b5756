// hal.hip -- arch record + tuned-solution table (see hal.h).
#include "hal.h"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/file.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "petit_internal.h"

namespace petit_amd {
namespace {

constexpr int kMaxDevices = 64;
ArchInfo g_arch[kMaxDevices];
std::once_flag g_arch_once[kMaxDevices];

ArchInfo datasheet_mi355x() {
    ArchInfo a{};
    snprintf(a.name, sizeof(a.name), "gfx950");
    a.num_cus = 256, a.lds_bytes_per_cu = 160 * 1024, a.max_waves_per_cu = 32;
    a.clock_khz = 2400000, a.mem_clock_khz = 0, a.mem_bus_bits = 8192;
    a.hbm_peak_gbs = 8000.0, a.bf16_peak_tflops = 2500.0, a.fp4_peak_tflops = 10000.0;
    a.native_fp4 = true;
    return a;
}

void query_arch(int dev) {
    ArchInfo a = datasheet_mi355x();
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) == hipSuccess) {
        // gcnArchName looks like "gfx950:sramecc+:xnack-"
        size_t n = strcspn(p.gcnArchName, ":");
        if (n >= sizeof(a.name))
            n = sizeof(a.name) - 1;
        memcpy(a.name, p.gcnArchName, n);
        a.name[n] = 0;
        if (p.multiProcessorCount > 0)
            a.num_cus = p.multiProcessorCount;
        if (p.maxSharedMemoryPerMultiProcessor > 0)
            a.lds_bytes_per_cu = (int)p.maxSharedMemoryPerMultiProcessor;
        if (p.maxThreadsPerMultiProcessor > 0)
            a.max_waves_per_cu = p.maxThreadsPerMultiProcessor / 64;
        if (p.clockRate > 0)
            a.clock_khz = p.clockRate;
        a.mem_clock_khz = p.memoryClockRate;
        if (p.memoryBusWidth > 0)
            a.mem_bus_bits = p.memoryBusWidth;
        a.native_fp4 = strcmp(a.name, "gfx950") == 0;
        // dense MFMA peaks: 4096 (bf16) / 16384 (fp4) FLOP per clock per CU
        a.bf16_peak_tflops = (double)a.num_cus * a.clock_khz * 1e3 * 4096.0 / 1e12;
        a.fp4_peak_tflops = a.native_fp4 ? 4.0 * a.bf16_peak_tflops : 0.0;
    }
    g_arch[dev] = a;
}

const TunedEntry kBuiltin[] = {
#include "tuned_gfx950.inc"
    {0, 0, 0, 0, 0, 0, 0} // terminator
};
// the opt-in native class (PETIT_SOLUTION_AUTO_NATIVE_*): same columns, ids of kind 9 / 13
const TunedEntry kBuiltinNative[] = {
#include "tuned_native_gfx950.inc"
    {0, 0, 0, 0, 0, 0, 0} // terminator
};

// rows from $PETIT_AMD_TUNE_FILE (read once) and rows added at run time, newest first; guarded by g_rows_mutex (the lookup
// runs only when a thread's default-pick cache misses, pick.hip choose_auto)
std::vector<TunedEntry> g_override;
std::once_flag g_override_once;
std::mutex g_rows_mutex;
std::atomic<uint64_t> g_generation{1};
std::atomic<size_t> g_override_count{0}; // rows in g_override: the lookups skip the mutex while there are none (the usual case)

// rows written by round 3 may carry b_type 8 / element nibble 3 ("MXFP4 with scales in fp16's range"): plain MXFP4 now (petit_internal.h)
TunedEntry canonical_row(TunedEntry e) {
    e.b_type = canonical_b_type(e.b_type);
    if (((e.solution >> 28) & 0xf) == 3)
        e.solution = (e.solution & ~((uint64_t)0xf << 28)) | ((uint64_t)2 << 28);
    return e;
}

// the rows of a tune file, in file order (a row naming a native-FP4 kernel lands in the native class by its id: never a plain default)
std::vector<TunedEntry> read_rows(const char *path) {
    std::vector<TunedEntry> rows;
    FILE *f = fopen(path, "r");
    if (!f)
        return rows;
    char line[256];
    while (fgets(line, sizeof(line), f)) {
        if (line[0] == '#' || line[0] == '\n')
            continue;
        TunedEntry e{};
        unsigned long long sol = 0;
        if (sscanf(line, "%d %d %u %u %u %u %llx", &e.a_type, &e.b_type, &e.n, &e.k, &e.m_lo, &e.m_hi, &sol) == 7 && e.m_lo >= 1 && e.m_hi >= e.m_lo) {
            e.solution = sol;
            // A file saved by a build of rounds 1-4 has open-ended rows "257 .. 2^20" measured at M = 512; file rows are looked up before the built-in
            // table, so such a row would hide the prefill buckets (513-1024, 1025-4096, 4097+) this build measures separately.  An open-ended row keeps
            // the bucket its lower end lies in: 257 .. 2^20 -> 257 .. 512; a row that starts in the last bucket stays open-ended.
            if (e.m_hi >= kMaxM) {
                unsigned end = 1;
                while (end < e.m_lo && end < 1024)
                    end *= 2;
                end = e.m_lo > 4096 ? kMaxM : e.m_lo > 1024 ? 4096u : end;
                e.m_hi = end;
            }
            rows.push_back(canonical_row(e));
        }
    }
    fclose(f);
    return rows;
}

void load_override() {
    const char *path = getenv("PETIT_AMD_TUNE_FILE");
    if (!path || !*path)
        return;
    std::vector<TunedEntry> rows = read_rows(path);
    std::lock_guard<std::mutex> lock(g_rows_mutex);
    g_override.insert(g_override.end(), rows.begin(), rows.end());
    g_override_count.store(g_override.size(), std::memory_order_release);
}

bool same_problem(const TunedEntry &a, const TunedEntry &b) {
    return a.a_type == b.a_type && a.b_type == b.b_type && a.n == b.n && a.k == b.k && solution_class(a.solution) == solution_class(b.solution);
}
// put `e` in front of `rows`; a row of the same problem and class that it overlaps keeps the part of its M range that `e` does not cover
// (a single-M row inserted into a 5-8 bucket leaves 5-6 and 8 with the bucket's kernel, not with nothing)
void insert_row(std::vector<TunedEntry> &rows, const TunedEntry &e) {
    std::vector<TunedEntry> out;
    out.reserve(rows.size() + 3);
    out.push_back(e);
    for (const TunedEntry &o : rows) {
        if (!same_problem(o, e) || o.m_hi < e.m_lo || o.m_lo > e.m_hi) {
            out.push_back(o);
            continue;
        }
        if (o.m_lo < e.m_lo) {
            TunedEntry left = o;
            left.m_hi = e.m_lo - 1;
            out.push_back(left);
        }
        if (o.m_hi > e.m_hi) {
            TunedEntry right = o;
            right.m_lo = e.m_hi + 1;
            out.push_back(right);
        }
    }
    rows.swap(out);
}

bool matches(const TunedEntry &e, int a_type, int b_type, unsigned m, unsigned n, unsigned k, int klass) {
    return e.a_type == a_type && e.b_type == b_type && e.n == n && e.k == k && m >= e.m_lo && m <= e.m_hi &&
           solution_class(e.solution) == klass;
}

// The built-in tables, indexed once: the generator sorts rows by (a_type, b_type, N, K, ...), so the rows of one (dtypes, shape) are one
// contiguous run.  A lookup is a map probe + a scan of that run (10-13 rows; x3 activation formats in the native table) instead of a scan
// of 3680 / 2760 rows with the M test last (eager prefill with varying token counts misses the per-thread pick cache on every
// call, and the miss path cost tens of microseconds next to 10-400 us kernels).  shapes[] also serves the nearest-shape search: one
// distance per distinct shape, not per row.
struct ShapeRun {
    int a_type, b_type;
    unsigned n, k;
    const TunedEntry *first, *last; // [first, last)
};
struct TableIndex {
    std::vector<ShapeRun> shapes;
    std::map<std::tuple<int, int, unsigned, unsigned>, int> by_key;
    explicit TableIndex(const TunedEntry *rows) {
        for (const TunedEntry *e = rows; e->solution;) {
            const TunedEntry *f = e;
            while (f->solution && f->a_type == e->a_type && f->b_type == e->b_type && f->n == e->n && f->k == e->k)
                ++f;
            const auto key = std::make_tuple(e->a_type, e->b_type, e->n, e->k);
            auto it = by_key.find(key);
            if (it == by_key.end()) {
                by_key.emplace(key, (int)shapes.size());
                shapes.push_back(ShapeRun{e->a_type, e->b_type, e->n, e->k, e, f});
            } else { // (a hand-edited table whose rows of one shape are not contiguous: never generated, but stay correct -- widen the run)
                ShapeRun &r = shapes[it->second];
                r.first = std::min(r.first, e), r.last = std::max(r.last, f);
            }
            e = f;
        }
    }
    const ShapeRun *find(int a_type, int b_type, unsigned n, unsigned k) const {
        const auto it = by_key.find(std::make_tuple(a_type, b_type, n, k));
        return it == by_key.end() ? nullptr : &shapes[it->second];
    }
};
const TableIndex &builtin_index(int klass) {
    static const TableIndex exact(kBuiltin), native(kBuiltinNative);
    return klass == 0 ? exact : native;
}

} // namespace

const ArchInfo &arch_info(int device) {
    if (device < 0 || device >= kMaxDevices)
        device = 0;
    std::call_once(g_arch_once[device], query_arch, device);
    return g_arch[device];
}

int solution_class(uint64_t solution) {
    const unsigned kind = (unsigned)(solution >> 48) & 0xf, mfma = (unsigned)(solution >> 32) & 0x7;
    if (kind != 9 && kind != 13)
        return 0;
    return mfma == 6 ? 4 : mfma == 4 ? 6 : 8; // mfma_type nibble: 2 = MXFP8 activations, 4 = MXFP6, 6 = MXFP4 (solution.h)
}

uint64_t tuned_generation() { return g_generation.load(std::memory_order_acquire); }

void tuned_insert(const TunedEntry &row) {
    const TunedEntry e = canonical_row(row);
    std::call_once(g_override_once, load_override);
    {
        std::lock_guard<std::mutex> lock(g_rows_mutex);
        insert_row(g_override, e);
        g_override_count.store(g_override.size(), std::memory_order_release);
    }
    g_generation.fetch_add(1, std::memory_order_acq_rel);
}

bool tuned_save(const char *path) {
    std::call_once(g_override_once, load_override);
    if (!path || !*path)
        return false;
    const std::string target(path), lock_path = target + ".lock", tmp = target + ".tmp." + std::to_string((long)getpid());
    // advisory lock shared by every process that saves to this file (kept for the read-merge-rename sequence; released by close)
    const int lock_fd = open(lock_path.c_str(), O_CREAT | O_RDWR, 0644);
    if (lock_fd >= 0)
        (void)flock(lock_fd, LOCK_EX);
    std::vector<TunedEntry> rows = read_rows(path); // what other processes have saved meanwhile (empty when the file does not exist yet)
    {
        std::lock_guard<std::mutex> lock(g_rows_mutex);
        for (auto it = g_override.rbegin(); it != g_override.rend(); ++it) // oldest first, so that this process's newest rows end up in front
            insert_row(rows, *it);
    }
    bool ok = false;
    if (FILE *f = fopen(tmp.c_str(), "w")) {
        fprintf(f, "# a_type b_type n k m_lo m_hi solution   (petit-kernel_amd tune file; $PETIT_AMD_TUNE_FILE)\n");
        for (const TunedEntry &e : rows)
            fprintf(f, "%d %d %u %u %u %u %llx\n", e.a_type, e.b_type, e.n, e.k, e.m_lo, e.m_hi, (unsigned long long)e.solution);
        ok = fclose(f) == 0 && rename(tmp.c_str(), path) == 0;
        if (!ok)
            (void)unlink(tmp.c_str());
    }
    if (lock_fd >= 0)
        close(lock_fd);
    return ok;
}

static bool tuned_disabled_by_env() {
    static const bool disabled = [] { // $PETIT_AMD_NO_TUNED=1: heuristic only (tools/check_heuristic.py measures what that costs)
        const char *e = getenv("PETIT_AMD_NO_TUNED");
        return e && *e && *e != '0';
    }();
    return disabled;
}
bool tuned_lookup_enabled(int device) {
    // tuned ids are only meaningful on the arch they were measured on
    return strcmp(arch_info(device).name, "gfx950") == 0 && !tuned_disabled_by_env();
}

uint64_t tuned_solution(int device, int a_type, int b_type, unsigned m, unsigned n, unsigned k, int klass) {
    if (!tuned_lookup_enabled(device))
        return 0;
    b_type = canonical_b_type(b_type);
    std::call_once(g_override_once, load_override);
    if (g_override_count.load(std::memory_order_acquire)) {
        std::lock_guard<std::mutex> lock(g_rows_mutex);
        for (const TunedEntry &e : g_override)
            if (matches(e, a_type, b_type, m, n, k, klass))
                return e.solution;
    }
    if (const ShapeRun *run = builtin_index(klass).find(a_type, b_type, n, k))
        for (const TunedEntry *e = run->first; e != run->last; ++e)
            if (matches(*e, a_type, b_type, m, n, k, klass))
                return e->solution;
    return 0;
}

int tuned_shape_rows(int device, int a_type, int b_type, unsigned n, unsigned k, int klass, TunedEntry *out, int cap) {
    if (!tuned_lookup_enabled(device) || cap <= 0)
        return 0;
    b_type = canonical_b_type(b_type);
    int count = 0;
    std::call_once(g_override_once, load_override);
    if (g_override_count.load(std::memory_order_acquire)) {
        std::lock_guard<std::mutex> lock(g_rows_mutex);
        for (const TunedEntry &e : g_override)
            if (e.a_type == a_type && e.b_type == b_type && e.n == n && e.k == k && solution_class(e.solution) == klass && count < cap)
                out[count++] = e;
    }
    if (const ShapeRun *run = builtin_index(klass).find(a_type, b_type, n, k))
        for (const TunedEntry *e = run->first; e != run->last && count < cap; ++e)
            if (e->a_type == a_type && e->b_type == b_type && e->n == n && e->k == k && solution_class(e->solution) == klass)
                out[count++] = *e;
    return count;
}

int tuned_nearest_list(int device, int a_type, int b_type, unsigned m, unsigned n, unsigned k, int klass, double max_distance, TunedNeighbour *out, int cap) {
    if (!tuned_lookup_enabled(device) || n == 0 || k == 0 || cap <= 0)
        return 0;
    b_type = canonical_b_type(b_type);
    auto span_class = [](unsigned kk) { return kk % 1024 == 0 ? 8 : kk % 512 == 0 ? 4 : 2; }; // (layout.h span_tiles_for_k: a kernel is built for one)
    const int ks = span_class(k);
    int count = 0;
    // keep the `cap` nearest, one per tabulated shape, nearest first (insertion into a tiny sorted array)
    auto offer = [&](const TunedEntry &e, double d) {
        for (int i = 0; i < count; ++i)
            if (out[i].n == e.n && out[i].k == e.k) { // (a run-time row and a built-in row of the same shape: the first offered wins -- run-time rows are offered first)
                return;
            }
        int pos = count;
        while (pos > 0 && out[pos - 1].distance > d)
            --pos;
        if (pos >= cap)
            return;
        for (int i = std::min(count, cap - 1); i > pos; --i)
            out[i] = out[i - 1];
        out[pos] = TunedNeighbour{e.solution, e.n, e.k, d};
        if (count < cap)
            ++count;
    };
    auto distance = [&](unsigned en, unsigned ek) { return 2.0 * std::fabs(std::log((double)n / en)) + std::fabs(std::log((double)k / ek)); };
    std::call_once(g_override_once, load_override);
    if (g_override_count.load(std::memory_order_acquire)) {
        std::lock_guard<std::mutex> lock(g_rows_mutex);
        for (const TunedEntry &e : g_override) {
            if (e.a_type != a_type || e.b_type != b_type || m < e.m_lo || m > e.m_hi || solution_class(e.solution) != klass || span_class(e.k) != ks)
                continue;
            const double d = distance(e.n, e.k);
            if (d < max_distance)
                offer(e, d);
        }
    }
    for (const ShapeRun &run : builtin_index(klass).shapes) {
        if (run.a_type != a_type || run.b_type != b_type || span_class(run.k) != ks)
            continue;
        const double d = distance(run.n, run.k);
        if (d >= max_distance || (count == cap && d >= out[count - 1].distance))
            continue;
        for (const TunedEntry *e = run.first; e != run.last; ++e)
            if (m >= e->m_lo && m <= e->m_hi && solution_class(e->solution) == klass) {
                offer(*e, d);
                break;
            }
    }
    return count;
}

uint64_t tuned_nearest(int device, int a_type, int b_type, unsigned m, unsigned n, unsigned k, int klass, double max_distance, unsigned *n_found,
                       unsigned *k_found) {
    TunedNeighbour one;
    if (tuned_nearest_list(device, a_type, b_type, m, n, k, klass, max_distance, &one, 1) == 0)
        return 0;
    if (n_found)
        *n_found = one.n;
    if (k_found)
        *k_found = one.k;
    return one.solution;
}

} // namespace petit_amd
