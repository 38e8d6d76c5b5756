// describe.hip -- human-readable text for return codes and solution ids (petit_error_string, petit_describe_solution).
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

using namespace petit_amd;

extern "C" {

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
