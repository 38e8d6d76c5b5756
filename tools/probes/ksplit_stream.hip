// tools/probes/ksplit_stream.hip -- can a CU stream BOTH FP4 operands of a 128 x 128 tile straight into registers (no LDS) at the rate the 32x32x64
// block-scaled MFMA consumes them?  Four waves per workgroup, each owns the whole 128 x 128 accumulator tile (256 registers) over a QUARTER of K;
// per k-tile (128 k) a wave loads 8 KiB of activations + 8 KiB of weights (16 wave-loads of 1 KiB) and issues 32 MFMAs.  Every operand byte enters
// the CU once: 16 KiB per 128 x 128 x 128 block, the minimum for this tile; no LDS traffic, no barrier.  Measured against the LDS-staged kernel
// (gemm_native32.hpp), whose 1 x 4 wave layout re-reads the activation tile from LDS four times.
#include <hip/hip_runtime.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int R, int MFMA, int INTER>
__global__ __launch_bounds__(256, 1) void ksplit_kernel(const u32x4 *__restrict__ wbuf, const u32x4 *__restrict__ abuf, unsigned ktiles, float *sink) {
    const unsigned lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned b = blockIdx.x, xcd = b & 7u, idx = b >> 3; // 32 workgroups per XCD: 8 weight panels x 4 row blocks
    const unsigned bm = idx & 3u, bn = xcd * 8 + (idx >> 2);
    const unsigned per_wave = ktiles / 4, kt0 = wave * per_wave;
    const u32x4 *w = wbuf + ((size_t)bn * ktiles + kt0) * 512 + lane; // 8 KiB = 512 u32x4 per k-tile
    const u32x4 *a = abuf + ((size_t)bm * ktiles + kt0) * 512 + lane;
    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v)
                acc[i][j][v] = 0.f;
    u32x4 ra[R][8], rw[R][8];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const unsigned t = r;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            ra[r][i] = a[(t * 8 + i) * 64];
            rw[r][i] = w[(t * 8 + i) * 64];
        }
    }
    for (unsigned kt = 0; kt < per_wave; kt += R) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const unsigned nk = kt + r + R; // (past the slice: re-reads the slice's first tiles -- same bytes, no branch)
            const unsigned off = (nk < per_wave ? nk : nk - per_wave) * 512;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) {
                    const u32x4 x = ra[r][mb * 2 + q];
                    const i32x8 aop = {(int)x[0], (int)x[1], (int)x[2], (int)x[3], 0, 0, 0, 0};
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb) {
                        const u32x4 y = rw[r][nb * 2 + q];
                        const i32x8 wop = {(int)y[0], (int)y[1], (int)y[2], (int)y[3], 0, 0, 0, 0};
                        if constexpr (MFMA)
                            acc[mb][nb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wop, aop, acc[mb][nb], 4, 4, 0, 127, 0, 127);
                        else
                            asm volatile("" ::"v"(wop), "v"(aop));
                    }
                    if constexpr (INTER) { // refill a fragment as soon as its last MFMA has issued: a continuous request stream
                        ra[r][mb * 2 + q] = a[off + (mb * 2 + q) * 64];
                        if (mb == 3) {
#pragma unroll
                            for (int nb = 0; nb < 4; ++nb)
                                rw[r][nb * 2 + q] = w[off + (nb * 2 + q) * 64];
                        }
                        if constexpr (MFMA) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                            if (mb == 3)
                                __builtin_amdgcn_sched_group_barrier(0x020, 5, 0);
                            else
                                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                        }
                    }
                }
            }
            if constexpr (!INTER) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    ra[r][i] = a[off + i * 64];
                    rw[r][i] = w[off + i * 64];
                }
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v)
                s += acc[i][j][v];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i)
            s += (float)(ra[r][i][0] ^ rw[r][i][0]);
    if (s == 12345.678f)
        sink[0] = s;
}

extern "C" int ksplit_launch(int r, int mfma, int stag, const void *w, const void *a, unsigned ktiles, void *sink, void *stream) {
    hipStream_t st = (hipStream_t)stream;
#define L(R_, M_, S_)                                                                                                                     \
    if (r == R_ && mfma == M_ && stag == S_) {                                                                                                        \
        hipLaunchKernelGGL((ksplit_kernel<R_, M_, S_>), dim3(256), dim3(256), 0, st, (const u32x4 *)w, (const u32x4 *)a, ktiles, (float *)sink); \
        return (int)hipGetLastError();                                                                                                  \
    }
    L(2, 1, 0) L(3, 1, 0) L(2, 0, 0) L(3, 0, 0) L(2, 1, 1) L(3, 1, 1) L(2, 0, 1) L(3, 0, 1)
#undef L
    return -1;
}
