// probes.hip -- tiny hardware-semantics probes run once on the MI355X box; results are
// recorded in DESIGN.md.  Not part of the product library.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;

// out[i*2..] = cvt_scalef32_pk_f32_fp4(word, scale[i]) for byte 0
__global__ void probe_cvt_scale(const unsigned *words, const float *scales, float *out_f32, unsigned *out_bf16, int n) {
    int i = threadIdx.x;
    if (i >= n) return;
    f32x2 f = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(words[i], scales[i], 0);
    out_f32[2 * i] = f.x;
    out_f32[2 * i + 1] = f.y;
    bf16x2 b = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(words[i], scales[i], 0);
    out_bf16[i] = __builtin_bit_cast(unsigned, b);
}

// raw buffer bounds semantics: is soffset part of the range check?
__global__ void probe_buffer_oob(const unsigned *buf, unsigned num_bytes, unsigned voff, unsigned soff, unsigned *out) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)buf, 0, (int)num_bytes, 0x00020000);
    out[threadIdx.x] = __builtin_amdgcn_raw_buffer_load_b32(r, voff + threadIdx.x * 4, soff, 0);
}

extern "C" void run_probe_cvt_scale(const unsigned *w, const float *s, float *of, unsigned *ob, int n, void *stream) {
    hipLaunchKernelGGL(probe_cvt_scale, dim3(1), dim3(64), 0, (hipStream_t)stream, w, s, of, ob, n);
}
extern "C" void run_probe_buffer_oob(const unsigned *buf, unsigned num_bytes, unsigned voff, unsigned soff, unsigned *out, void *stream) {
    hipLaunchKernelGGL(probe_buffer_oob, dim3(1), dim3(64), 0, (hipStream_t)stream, buf, num_bytes, voff, soff, out);
}

// launch-floor probe: an empty kernel at a given grid/block, and a pure streaming read
// (nt dwordx4, `per_thread` loads in flight per thread, grid-stride) summing into a sink.
__global__ void probe_empty(unsigned *sink) {
    if (threadIdx.x == 0xffffffffu) sink[0] = 1;
}
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
template <int PER> __global__ void probe_stream(const u32x4 *src, size_t n16, unsigned *sink) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    u32x4 acc = {0, 0, 0, 0};
    for (; i + (PER - 1) * stride < n16; i += PER * stride) {
        u32x4 v[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) v[j] = __builtin_nontemporal_load(src + i + j * stride);
#pragma unroll
        for (int j = 0; j < PER; ++j) acc ^= v[j];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[threadIdx.x] = acc.x;
}
// contiguous-per-wave variant: each wave reads `PER` consecutive KiB (like the GEMM's tiles)
template <int PER> __global__ void probe_stream_wave(const u32x4 *src, size_t n16, unsigned *sink) {
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned lane = threadIdx.x & 63;
    const size_t base = wave * (size_t)PER * 64 + lane;
    u32x4 acc = {0, 0, 0, 0};
    if (base + (PER - 1) * 64 < n16) {
        u32x4 v[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) v[j] = __builtin_nontemporal_load(src + base + j * 64);
#pragma unroll
        for (int j = 0; j < PER; ++j) acc ^= v[j];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[threadIdx.x] = acc.x;
}
// workgroup-interleaved variant: the G waves of a workgroup own G*PER consecutive KiB; at step j
// wave i reads KiB (j*G + i) of that block, so the workgroup's concurrent loads are contiguous.
template <int PER> __global__ void probe_stream_wgil(const u32x4 *src, size_t n16, unsigned *sink) {
    const unsigned G = blockDim.x >> 6, wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t base = (size_t)blockIdx.x * G * PER * 64 + (size_t)wv * 64 + lane;
    u32x4 acc = {0, 0, 0, 0};
    if (base + (size_t)(PER - 1) * G * 64 < n16) {
        u32x4 v[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) v[j] = __builtin_nontemporal_load(src + base + (size_t)j * G * 64);
#pragma unroll
        for (int j = 0; j < PER; ++j) acc ^= v[j];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[threadIdx.x] = acc.x;
}
extern "C" void run_probe_empty(unsigned grid, unsigned block, unsigned *sink, void *stream) {
    hipLaunchKernelGGL(probe_empty, dim3(grid), dim3(block), 0, (hipStream_t)stream, sink);
}
extern "C" void run_probe_stream(int kind, int per, unsigned grid, unsigned block, const void *src, size_t bytes,
                                 unsigned *sink, void *stream) {
    const u32x4 *s = (const u32x4 *)src;
    const size_t n16 = bytes / 16;
    hipStream_t st = (hipStream_t)stream;
#define GO(K, P)                                                                                        \
    if (kind == 0 && per == P) hipLaunchKernelGGL(probe_stream<P>, dim3(grid), dim3(block), 0, st, s, n16, sink); \
    if (kind == 1 && per == P) hipLaunchKernelGGL(probe_stream_wave<P>, dim3(grid), dim3(block), 0, st, s, n16, sink); \
    if (kind == 2 && per == P) hipLaunchKernelGGL(probe_stream_wgil<P>, dim3(grid), dim3(block), 0, st, s, n16, sink);
    GO(0, 1) GO(0, 2) GO(0, 4) GO(0, 8) GO(0, 16)
}
