// l2_ingest.hip -- how fast can a CU pull data it shares with every other CU (the operand streams of the large-M kernels)?
// Each wave reads 1 KiB chunks (one buffer_load_dwordx4 per lane, the kernels' access shape) from a window of `window_bytes` that ALL
// workgroups walk (so it lives in L2 / Infinity Cache, like an activation tile or a weight panel shared by the m-blocks), keeping U loads
// in flight.  Reports nothing itself: time it from the host (run_l2_ingest.py) for waves-per-CU x U x window size.
#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int U>
__global__ __launch_bounds__(256) void ingest_kernel(const u32x4 *src, unsigned window_u4, unsigned iters, unsigned *sink) {
    const unsigned lane = threadIdx.x & 63u, wave = (blockIdx.x * 4 + (threadIdx.x >> 6));
    // every wave starts somewhere else in the window and strides through it chunk by chunk (64 u4 = 1 KiB per wave-load)
    unsigned pos = (wave * 977u * 64u) % window_u4;
    u32x4 acc = u32x4{0, 0, 0, 0};
    for (unsigned it = 0; it < iters; it += U) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            v[u] = src[pos + lane];
            pos += 64u * 61u; // a stride of 61 chunks: consecutive loads of a wave hit different channels
            if (pos >= window_u4)
                pos -= window_u4;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            acc ^= v[u];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u)
        sink[0] = 1; // never true in practice: keeps the loads alive
}

extern "C" int ingest_launch(int u, const void *src, unsigned window_bytes, unsigned iters, unsigned blocks, void *sink, void *stream) {
    const unsigned window_u4 = window_bytes / 16;
#define L(UU) hipLaunchKernelGGL(ingest_kernel<UU>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const u32x4 *)src, window_u4, iters, (unsigned *)sink)
    if (u == 2) L(2); else if (u == 4) L(4); else if (u == 8) L(8); else if (u == 16) L(16); else return 1;
#undef L
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

