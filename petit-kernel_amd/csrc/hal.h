// hal.h -- "hardware abstraction" of this build: the per-device arch record and
// the tuned-solution table consulted for solution_id = -1.
//
// The reference's lib/hal is a hipMalloc/hipMemcpy wrapper for its gtests
// (lib/hal/device.h:8-34) and has no arch knowledge at all: its only arch
// switches are two #ifs and a `major*10+minor <= 90` test
// (lib/pybind/fp4.cc:24-34), and its default-solution heuristic is blind to
// the CU count (fp4/algo_chooser.cc:64-132).  BASELINE.json asks for arch
// tables, so the name is reused for what is new here:
//   * ArchInfo   -- CU count, LDS size, clocks, queried once per device;
//   * the tuned table -- (a_type, b_type, N, K, M range) -> solution id, the
//     persisted result of tools/tune.py sweeps on a real MI355X
//     (tuned_gfx950.inc), overridable at run time through the text file named
//     by $PETIT_AMD_TUNE_FILE (same columns, one entry per line).
#pragma once

#include <stdint.h>

namespace petit_amd {

struct ArchInfo {
    char name[32];         // "gfx950"
    int num_cus;           // 256 on MI355X
    int lds_bytes_per_cu;  // 163840
    int max_waves_per_cu;  // 32
    int clock_khz;         // peak engine clock
    int mem_clock_khz;
    int mem_bus_bits;
    double hbm_peak_gbs;   // spec: 8000 GB/s on MI355X
    double bf16_peak_tflops; // dense: 2500
    double fp4_peak_tflops;  // dense: 10000
    bool native_fp4;       // v_mfma_scale_f32_*_f8f6f4 + v_cvt_scalef32_*_fp4
};

// Cached after the first call per device; never fails (falls back to the
// MI355X datasheet record when the runtime query does).
const ArchInfo &arch_info(int device);

struct TunedEntry {
    int a_type, b_type; // DataType values (petit_internal.h)
    unsigned n, k;
    unsigned m_lo, m_hi; // inclusive range of M this entry covers
    uint64_t solution;   // full id incl. split-K nibble
};

// 0 when the table has no entry for this problem.  klass 0: the exact kernels (what PETIT_SOLUTION_AUTO may run); 8 / 4: the
// opt-in native class with MXFP8 / MXFP4 activations (its own table: a row naming a native kernel is never seen by klass 0).
// Lookup order: rows added at run time (tuned_insert: petit_gemm_tune, $PETIT_AMD_AUTOTUNE), $PETIT_AMD_TUNE_FILE, built-in.
uint64_t tuned_solution(int device, int a_type, int b_type, unsigned m, unsigned n, unsigned k, int klass = 0);
// No row for (n, k)?  The row of the NEAREST tabulated shape of the same dtypes, class and span size whose M range holds m:
// distance 2 |ln(n / n')| + |ln(k / k')| (the kernel choice follows the column count -- how the grid fills the chip -- more than the
// reduction length), 0 when nothing lies within `max_distance`.  Measured on ten shapes kept out of the table (profiles/r04_heuristic.md):
// the neighbour's kernel is within 1 % of the best kernel in the median, 14 % at the 90th percentile; the formula-based heuristic 5 % / 24 %.
uint64_t tuned_nearest(int device, int a_type, int b_type, unsigned m, unsigned n, unsigned k, int klass, double max_distance, unsigned *n_found = nullptr,
                       unsigned *k_found = nullptr);
// The `cap` nearest tabulated shapes (one entry per shape, nearest first, each with the row that holds m), for callers that want to rank them:
// pick.hip choose_auto scores the neighbours' kernels for THIS problem's grid instead of taking the nearest blindly.
struct TunedNeighbour {
    uint64_t solution;
    unsigned n, k;
    double distance;
};
int tuned_nearest_list(int device, int a_type, int b_type, unsigned m, unsigned n, unsigned k, int klass, double max_distance, TunedNeighbour *out, int cap);
// Every built-in / run-time row of ONE tabulated shape and class (all M buckets), for callers that want to weigh a bucket's row against its neighbours'
// (pick.hip choose_auto: tile quantisation at a ragged prefill M).  Returns the count written (<= cap).
int tuned_shape_rows(int device, int a_type, int b_type, unsigned n, unsigned k, int klass, TunedEntry *out, int cap);
// class of a solution id: 0 exact, 8 / 4 native with MXFP8 / MXFP4 activations
int solution_class(uint64_t solution);
// Add (or replace) a row at run time; thread safe; bumps tuned_generation() so cached default picks are re-derived.
void tuned_insert(const TunedEntry &e);
uint64_t tuned_generation();
// false when tuned_solution() can never answer for this device ($PETIT_AMD_NO_TUNED=1, or not the arch the tables were measured on):
// tuning on first sight would then repeat on every call and its result would never be used
bool tuned_lookup_enabled(int device);
// Write every run-time and tune-file row (not the built-in table) in the $PETIT_AMD_TUNE_FILE format; false on I/O error.  Several
// processes may share one file (TP ranks with $PETIT_AMD_AUTOTUNE=1): the rows already in the file are merged in (this process's rows win
// where they overlap) under an advisory lock on "<path>.lock", and the file is replaced by rename(), never rewritten in place.
bool tuned_save(const char *path);

} // namespace petit_amd
