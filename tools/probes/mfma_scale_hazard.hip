// mfma_scale_hazard.hip -- how many wait states does gfx950 need between a VALU write of a VGPR and a
// v_mfma_scale_f32_16x16x128_f8f6f4 that reads it as its scale operand?  (hipcc 7.2 inserts `s_nop 1`.)
// out[N][lane*4+i] for N = 0..9 wait states; the scale register holds 2^3 ("stale") and is overwritten
// with 2^0 right before the MFMA: a result 8x too large means the MFMA saw the stale value.
#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int N, int MODE> __device__ f32x4 one(i32x4 a, i32x8 b, int sa, int stale, int fresh) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int s;
    if constexpr (MODE == 0) {          // scale_b written by v_mov
        asm volatile("v_mov_b32 %[s], %[stale]\n\ts_nop 7\n\ts_nop 7\n\t"
                     "v_mov_b32 %[s], %[fresh]\n\ts_nop %[n]\n\t"
                     "v_mfma_scale_f32_16x16x128_f8f6f4 %[acc], %[a], %[b], %[acc], %[sa], %[s] op_sel_hi:[0,0,0] cbsz:4\n\t"
                     "s_nop 7\n\ts_nop 7\n\ts_nop 7"
                     : [acc] "+v"(acc), [s] "=&v"(s)
                     : [a] "v"(a), [b] "v"(b), [sa] "v"(sa), [stale] "v"(stale), [fresh] "v"(fresh), [n] "i"(N));
    } else if constexpr (MODE == 1) {   // scale_b written by v_accvgpr_read
        asm volatile("v_mov_b32 %[s], %[stale]\n\ts_nop 7\n\ts_nop 7\n\t"
                     "v_accvgpr_read_b32 %[s], %[fresh]\n\ts_nop %[n]\n\t"
                     "v_mfma_scale_f32_16x16x128_f8f6f4 %[acc], %[a], %[b], %[acc], %[sa], %[s] op_sel_hi:[0,0,0] cbsz:4\n\t"
                     "s_nop 7\n\ts_nop 7\n\ts_nop 7"
                     : [acc] "+v"(acc), [s] "=&v"(s)
                     : [a] "v"(a), [b] "v"(b), [sa] "v"(sa), [stale] "v"(stale), [fresh] "a"(fresh), [n] "i"(N));
    } else if constexpr (MODE == 2) {   // scale_a written by v_mov
        asm volatile("v_mov_b32 %[s], %[stale]\n\ts_nop 7\n\ts_nop 7\n\t"
                     "v_mov_b32 %[s], %[fresh]\n\ts_nop %[n]\n\t"
                     "v_mfma_scale_f32_16x16x128_f8f6f4 %[acc], %[a], %[b], %[acc], %[s], %[sa] op_sel_hi:[0,0,0] cbsz:4\n\t"
                     "s_nop 7\n\ts_nop 7\n\ts_nop 7"
                     : [acc] "+v"(acc), [s] "=&v"(s)
                     : [a] "v"(a), [b] "v"(b), [sa] "v"(sa), [stale] "v"(stale), [fresh] "v"(fresh), [n] "i"(N));
    } else {                            // back-to-back: MFMA #1 reads s, then s is overwritten (WAR), MFMA #2 reads new s
        f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
        asm volatile("v_mov_b32 %[s], %[stale]\n\ts_nop 7\n\ts_nop 7\n\t"
                     "v_mfma_scale_f32_16x16x128_f8f6f4 %[acc2], %[a], %[b], %[acc2], %[sa], %[s] op_sel_hi:[0,0,0] cbsz:4\n\t"
                     "v_mov_b32 %[s], %[fresh]\n\ts_nop %[n]\n\t"
                     "v_mfma_scale_f32_16x16x128_f8f6f4 %[acc], %[a], %[b], %[acc], %[sa], %[s] op_sel_hi:[0,0,0] cbsz:4\n\t"
                     "s_nop 7\n\ts_nop 7\n\ts_nop 7"
                     : [acc] "+v"(acc), [acc2] "+v"(acc2), [s] "=&v"(s)
                     : [a] "v"(a), [b] "v"(b), [sa] "v"(sa), [stale] "v"(stale), [fresh] "v"(fresh), [n] "i"(N));
        acc[1] = acc2[0]; // expect 8x (stale on purpose) for MFMA #1, 1x for MFMA #2
    }
    return acc;
}

template <int MODE> __global__ void k(float *out) {
    const int l = threadIdx.x;
    const i32x4 a = {0x22222222, 0x22222222, 0x22222222, 0x22222222};        // fp4 1.0
    const i32x8 b = {0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838}; // e4m3 1.0
    const int sa = 0x7f7f7f7f, stale = 127 + 3, fresh = 127;
    f32x4 r[10];
    r[0] = one<0, MODE>(a, b, sa, stale, fresh);
    r[1] = one<1, MODE>(a, b, sa, stale, fresh);
    r[2] = one<2, MODE>(a, b, sa, stale, fresh);
    r[3] = one<3, MODE>(a, b, sa, stale, fresh);
    r[4] = one<4, MODE>(a, b, sa, stale, fresh);
    r[5] = one<5, MODE>(a, b, sa, stale, fresh);
    r[6] = one<6, MODE>(a, b, sa, stale, fresh);
    r[7] = one<7, MODE>(a, b, sa, stale, fresh);
    r[8] = one<8, MODE>(a, b, sa, stale, fresh);
    r[9] = one<10, MODE>(a, b, sa, stale, fresh);
    for (int n = 0; n < 10; ++n)
        for (int i = 0; i < 4; ++i) out[(n * 64 + l) * 4 + i] = r[n][i];
}
int main() {
    float *d; hipMalloc(&d, 4 * 10 * 64 * 4 * 4);
    static float h[4][10][64][4];
    hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, d);
    hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, d + 10 * 256);
    hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, d + 20 * 256);
    hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, d + 30 * 256);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char *names[4] = {"scale_b <- v_mov", "scale_b <- v_accvgpr_read", "scale_a <- v_mov", "WAR after MFMA (acc[1] = first MFMA, expect 1024)"};
    for (int m = 0; m < 4; ++m) {
        printf("%s (expect 128 when the fresh scale is seen, 1024 when stale)\n", names[m]);
        for (int n = 0; n < 10; ++n) {
            int bad = 0;
            for (int l = 0; l < 64; ++l) for (int i = (m == 3 ? 2 : 0); i < 4; ++i) bad += h[m][n][l][i] != 128.f;
            printf("  wait states %2d: lane0 = %g %g %g %g   wrong elements: %d\n", n == 9 ? 11 : n + 1, h[m][n][0][0], h[m][n][0][1], h[m][n][0][2], h[m][n][0][3], bad);
        }
    }
    return 0;
}
