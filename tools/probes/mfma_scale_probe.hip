// mfma_scale_probe.hip -- pins down the operand layout of v_mfma_scale_f32_16x16x128_f8f6f4 on gfx950
// for A = FP4 (cbsz 4) and B = FP8 e4m3 (blgp 0): which k a given nibble / byte is, which lane group
// pairs with which, and how the per-lane E8M0 scale bytes are selected.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// a_words[lane*8 + r], b_words[lane*8 + r], scales: sa[lane], sb[lane]; out[lane*4 + i]
template <int OPA, int OPB> __global__ void k(const int *a_words, const int *b_words, const int *sa, const int *sb, float *out) {
    const int l = threadIdx.x;
    i32x8 a, b;
    for (int r = 0; r < 8; ++r) { a[r] = a_words[l * 8 + r]; b[r] = b_words[l * 8 + r]; }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 4 /*A fp4*/, 0 /*B fp8 e4m3*/, OPA, sa[l], OPB, sb[l]);
    for (int i = 0; i < 4; ++i) out[l * 4 + i] = c[i];
}
extern "C" void run_mfma_scale(int opa, int opb, const int *a, const int *b, const int *sa, const int *sb, float *out, void *st) {
    hipStream_t s = (hipStream_t)st;
    if (opa == 0 && opb == 0) hipLaunchKernelGGL((k<0, 0>), dim3(1), dim3(64), 0, s, a, b, sa, sb, out);
    if (opa == 1 && opb == 0) hipLaunchKernelGGL((k<1, 0>), dim3(1), dim3(64), 0, s, a, b, sa, sb, out);
    if (opa == 2 && opb == 0) hipLaunchKernelGGL((k<2, 0>), dim3(1), dim3(64), 0, s, a, b, sa, sb, out);
    if (opa == 3 && opb == 0) hipLaunchKernelGGL((k<3, 0>), dim3(1), dim3(64), 0, s, a, b, sa, sb, out);
    if (opa == 0 && opb == 1) hipLaunchKernelGGL((k<0, 1>), dim3(1), dim3(64), 0, s, a, b, sa, sb, out);
    if (opa == 0 && opb == 3) hipLaunchKernelGGL((k<0, 3>), dim3(1), dim3(64), 0, s, a, b, sa, sb, out);
}
