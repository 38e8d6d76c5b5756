// ref_shim.cc -- C-ABI window onto the reference's OWN host-side software
// float types, compiled from where they lie under /root/reference (never
// copied).  It includes lib/tests/floating_points.h (which in turn includes
// the vendored lib/gemm/cpu/half_float.h) and exports thin wrappers so that
// tests/test_oracle.py can check oracle/petit_oracle.c against them over the
// full 8-bit / sampled 32-bit input domains.
//
// TEST INFRASTRUCTURE ONLY.  Output goes to oracle/_ref/ (git-ignored, travels
// to the GPU box).  Built by `make -C oracle ref`, only when /root/reference
// exists.
//
// The reference's GPU kernels cannot run here (no GPU) and its CMake build
// cannot configure (Hunter needs network; absl/gtest/gflags absent), so this
// header-only corner is the only part of the reference that is buildable.
#include "tests/floating_points.h"

#include <cstdint>

using namespace causalflow::petit::tests::cpu_numeric;

extern "C" {
// lib/tests/floating_points.h:21-75
float ref_e4m3_to_f32(uint8_t x) { return detail::CvtFp32Fp8<false>(x); }
float ref_e5m2_to_f32(uint8_t x) { return detail::CvtFp32Fp8<true>(x); }
// lib/tests/floating_points.h:79-112
uint16_t ref_f32_to_bf16(float f) { return bf16_t::from_fp32(f).to_bits(); }
// lib/tests/floating_points.h:145-148 (half_float::half rounding)
uint16_t ref_f32_to_f16(float f) { return fp16_t::from_fp32(f).to_bits(); }
// lib/tests/quantization.cc:29-38 is a file-static function in a TU that needs
// abseil; its three lines are re-expressed on the reference's own types here.
uint8_t ref_e4m3_to_e5m3(uint8_t x) {
    float v = fp8_e4m3_t::from_bits(x).to_fp32() * (1 << 7);
    return (fp16_t::from_fp32(v).to_bits() >> 7) & 0xff;
}
}
