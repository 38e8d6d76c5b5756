// valu_rate.hip -- issue cost (cycles per wave-instruction on one SIMD) of the instructions the
// unpack path is made of.  One wave per block, 64 independent chains... measured with s_memtime.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

#define REP 64
template <int OP> __global__ void k(unsigned *out, unsigned seed, float fs) {
    unsigned w[8];
    f32x2 f[8];
    unsigned acc = 0;
    for (int i = 0; i < 8; ++i) { w[i] = seed * (threadIdx.x + 17 * i + 1); f[i] = f32x2{(float)w[i], fs}; }
    long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < REP; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if constexpr (OP == 0) { f[i] = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(w[i], fs, 0); w[i] += __builtin_bit_cast(unsigned, f[i].x); }
            if constexpr (OP == 1) { bf16x2 b = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w[i], fs, 1); w[i] += __builtin_bit_cast(unsigned, b); }
            if constexpr (OP == 2) { f[i] = f[i] * f32x2{fs, fs}; }
            if constexpr (OP == 3) { bf16x2 b = __builtin_convertvector(f[i], bf16x2); f[i].x += __builtin_bit_cast(float, b); }
            if constexpr (OP == 4) { f[i].x = __builtin_fmaf(f[i].x, fs, f[i].y); }
            if constexpr (OP == 5) { f16x2 h = __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(w[i], fs, 2); w[i] += __builtin_bit_cast(unsigned, h); }
            if constexpr (OP == 6) { f16x2 h = __builtin_bit_cast(f16x2, w[i]); h = h * h; w[i] = __builtin_bit_cast(unsigned, h); }
            if constexpr (OP == 7) { f[i].x = __builtin_amdgcn_cvt_f32_fp8((int)w[i], 1) + f[i].x; }
            if constexpr (OP == 8) { w[i] = w[i] + (w[i] >> 3); }
        }
    }
    long t1 = __builtin_readcyclecounter();
    for (int i = 0; i < 8; ++i) acc ^= w[i] ^ __builtin_bit_cast(unsigned, f[i].x) ^ __builtin_bit_cast(unsigned, f[i].y);
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = (unsigned)(t1 - t0); out[blockIdx.x * 2 + 1] = acc; }
}
extern "C" void run_valu_rate(int op, unsigned blocks, unsigned threads, unsigned *out, void *stream) {
    hipStream_t st = (hipStream_t)stream;
#define GO(O) if (op == O) hipLaunchKernelGGL(k<O>, dim3(blocks), dim3(threads), 0, st, out, 12345u, 1.5f);
    GO(0) GO(1) GO(2) GO(3) GO(4) GO(5) GO(6) GO(7) GO(8)
}
