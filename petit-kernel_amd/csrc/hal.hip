// hal.hip -- arch record + tuned-solution table (see hal.h).
#include "hal.h"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
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

std::vector<TunedEntry> g_override;
std::once_flag g_override_once;

void load_override() {
    const char *path = getenv("PETIT_AMD_TUNE_FILE");
    if (!path || !*path)
        return;
    FILE *f = fopen(path, "r");
    if (!f)
        return;
    char line[256];
    while (fgets(line, sizeof(line), f)) {
        if (line[0] == '#' || line[0] == '\n')
            continue;
        TunedEntry e{};
        unsigned long long sol = 0;
        if (sscanf(line, "%d %d %u %u %u %u %llx", &e.a_type, &e.b_type, &e.n, &e.k, &e.m_lo, &e.m_hi, &sol) == 7) {
            e.solution = sol;
            if (((sol >> 48) & 0xf) == 9)
                continue; // a native-FP4 kernel is never a default (own accuracy class): the row is ignored
            g_override.push_back(e);
        }
    }
    fclose(f);
}

bool matches(const TunedEntry &e, int a_type, int b_type, unsigned m, unsigned n, unsigned k) {
    return e.a_type == a_type && e.b_type == b_type && e.n == n && e.k == k && m >= e.m_lo && m <= e.m_hi;
}

} // namespace

const ArchInfo &arch_info(int device) {
    if (device < 0 || device >= kMaxDevices)
        device = 0;
    std::call_once(g_arch_once[device], query_arch, device);
    return g_arch[device];
}

uint64_t tuned_solution(int device, int a_type, int b_type, unsigned m, unsigned n, unsigned k) {
    // tuned ids are only meaningful on the arch they were measured on
    if (strcmp(arch_info(device).name, "gfx950") != 0)
        return 0;
    static const bool disabled = [] { // $PETIT_AMD_NO_TUNED=1: heuristic only (tools/check_heuristic.py measures what that costs)
        const char *e = getenv("PETIT_AMD_NO_TUNED");
        return e && *e && *e != '0';
    }();
    if (disabled)
        return 0;
    std::call_once(g_override_once, load_override);
    for (const TunedEntry &e : g_override)
        if (matches(e, a_type, b_type, m, n, k))
            return e.solution;
    for (const TunedEntry *e = kBuiltin; e->solution; ++e)
        if (matches(*e, a_type, b_type, m, n, k))
            return e->solution;
    return 0;
}

} // namespace petit_amd
