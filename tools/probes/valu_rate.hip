// valu_rate.hip -- issue cost (cycles per wave-instruction) of the instructions the unpack path is
// made of.  Each kernel issues 16 INDEPENDENT copies of one instruction per loop iteration through
// asm volatile (nothing can be hoisted or merged), timed with s_memtime inside the wave.
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ITERS 256
#define X16(S) S S S S S S S S S S S S S S S S

template <int OP> __global__ void k(unsigned *out, unsigned seed, float fs) {
    unsigned w = seed * (threadIdx.x + 1);
    float f0 = (float)w, f1 = fs;
    unsigned d0, d1;  // sinks (overwritten; volatile asm keeps every instance)
    unsigned long long dd;
    long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; ++it) {
        if constexpr (OP == 0) { X16(asm volatile("v_cvt_scalef32_pk_f32_fp4 %0, %1, %2" : "=v"(dd) : "v"(w), "v"(fs));) }
        if constexpr (OP == 1) { X16(asm volatile("v_cvt_scalef32_pk_bf16_fp4 %0, %1, %2" : "=v"(d0) : "v"(w), "v"(fs));) }
        if constexpr (OP == 2) { X16(asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(dd) : "v"(dd), "v"(dd));) }
        if constexpr (OP == 3) { X16(asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d0) : "v"(f0), "v"(f1));) }
        if constexpr (OP == 4) { X16(asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d0) : "v"(f0), "v"(f1), "v"(f1));) }
        if constexpr (OP == 5) { X16(asm volatile("v_cvt_scalef32_pk_f16_fp4 %0, %1, %2" : "=v"(d0) : "v"(w), "v"(fs));) }
        if constexpr (OP == 6) { X16(asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(d0) : "v"(w), "v"(w));) }
        if constexpr (OP == 7) { X16(asm volatile("v_cvt_f32_fp8 %0, %1" : "=v"(d0) : "v"(w));) }
        if constexpr (OP == 8) { X16(asm volatile("v_add_u32 %0, %1, %2" : "=v"(d0) : "v"(w), "v"(w));) }
        if constexpr (OP == 9) { X16(asm volatile("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(d0) : "v"(w), "v"(w), "v"(f1));) }
        if constexpr (OP == 10) { X16(asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(dd) : "v"(dd), "v"(dd), "v"(dd));) }
        if constexpr (OP == 11) { X16(asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(d0), "+v"(d1));) }
        if constexpr (OP == 12) { X16(asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(d0) : "v"(f0), "v"(f1));) }
        if constexpr (OP == 13) { X16(asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(d0) : "v"(w), "v"(w), "v"(w));) }
    }
    long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[blockIdx.x] = (unsigned)(t1 - t0);
    if (w == 0xdeadbeef) out[1000000] = d0 + d1 + (unsigned)dd;
}
extern "C" void run_valu_rate(int op, unsigned blocks, unsigned threads, unsigned *out, void *stream) {
    hipStream_t st = (hipStream_t)stream;
#define GO(O) if (op == O) hipLaunchKernelGGL(k<O>, dim3(blocks), dim3(threads), 0, st, out, 12345u, 1.5f);
    GO(0) GO(1) GO(2) GO(3) GO(4) GO(5) GO(6) GO(7) GO(8) GO(9) GO(10) GO(11) GO(12) GO(13)
}
