// mfma_scale_rate.hip -- sustained issue rate of the MFMAs this library uses, measured on the chip (all 256 CUs busy,
// one wave per SIMD, 8 independent accumulators, no memory traffic): the practical MFMA roofline under the
// power/clock management of a full-chip MFMA load.  Inline asm so that the loop is exactly 8 MFMAs + a branch.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define LOOP8(STMT) STMT(0) STMT(1) STMT(2) STMT(3) STMT(4) STMT(5) STMT(6) STMT(7)

template <int KIND> __global__ __launch_bounds__(256) void k(float *out, int iters) {
    i32x8 a8, b8;
    i32x4 a4, b4;
    for (int r = 0; r < 8; ++r) { a8[r] = 0x38383838 + threadIdx.x * r; b8[r] = 0x38383838 ^ (threadIdx.x + r); }
    for (int r = 0; r < 4; ++r) { a4[r] = 0x22222222 + threadIdx.x * r; b4[r] = 0x22222222 ^ (threadIdx.x + r); }
    const int s = 0x7f7f7f7f;
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == 0) {        // A fp4, B fp8: what gemm_native.hpp issues
#define S(i) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0] cbsz:4" : "+v"(acc[i]) : "v"(a4), "v"(b8), "v"(s));
            LOOP8(S)
#undef S
        } else if constexpr (KIND == 1) { // A fp8, B fp8
#define S(i) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+v"(acc[i]) : "v"(a8), "v"(b8), "v"(s));
            LOOP8(S)
#undef S
        } else if constexpr (KIND == 2) { // A fp4, B fp4
#define S(i) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0] cbsz:4 blgp:4" : "+v"(acc[i]) : "v"(a4), "v"(b4), "v"(s));
            LOOP8(S)
#undef S
        } else {                          // v_mfma_f32_16x16x32_bf16 (the dequant kernels)
#define S(i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a4), "v"(b4));
            LOOP8(S)
#undef S
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15");
    f32x4 t = acc[0];
    for (int i = 1; i < 8; ++i) t += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = t[0] + t[1] + t[2] + t[3];
}

template <int KIND> static void run(const char *name, float *d, int cus, double flop_per_mfma) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<KIND>), dim3(cus), dim3(256), 0, 0, d, iters);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double mfmas = (double)cus * 4 * iters * 8;
        if (rep == 2)
            printf("%-40s %8.3f ms  %7.1f TFLOP/s  %.2f ns per MFMA per SIMD (%.1f cycles at 2.4 GHz)\n", name, ms,
                   mfmas * flop_per_mfma / ms / 1e9, ms * 1e6 / (iters * 8.0), ms * 1e6 / (iters * 8.0) * 2.4);
    }
}
int main() {
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    float *d;
    (void)hipMalloc(&d, (size_t)cus * 256 * 4);
    printf("%s, %d CUs, clock %d MHz\n", p.gcnArchName, cus, p.clockRate / 1000);
    run<0>("mfma_scale 16x16x128  A fp4, B fp8", d, cus, 2.0 * 16 * 16 * 128);
    run<1>("mfma_scale 16x16x128  A fp8, B fp8", d, cus, 2.0 * 16 * 16 * 128);
    run<2>("mfma_scale 16x16x128  A fp4, B fp4", d, cus, 2.0 * 16 * 16 * 128);
    run<3>("mfma 16x16x32 bf16", d, cus, 2.0 * 16 * 16 * 32);
    return 0;
}
