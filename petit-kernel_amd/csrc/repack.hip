// repack.hip -- offline weight / scale shuffles into the petit-cdna4 layout
// (layout.h).  Replaces, for gfx950:
//   RepackNvFp4ToPetitFp4WeightsKernel  quantization_utils.cu:208-253 (+:729-746)
//   RepackFp4ScalesKernel (NV)          quantization_utils.cu:255-304 (+:748-760)
//   RepackFp4ScalesKernel (MX)          quantization_utils.cu:255-304 (+:762-773)
// These are one-time, pure byte-movement kernels: every thread produces one
// coalesced output vector and gathers its inputs; no nibble re-encode (the
// reference's PetitFormat, :183-206) is needed because gfx950 converts raw
// E2M1 in hardware.
#include <hip/hip_runtime.h>

#include "layout.h"
#include "petit_internal.h"

namespace petit_amd {

// Gather indices, shared by the device kernels and the host (offline) twins below.
// packed uint4 `o` of the weights <- native uint4 index
PETIT_HD size_t weight_src_u4(size_t o, unsigned k) {
    const unsigned row_u4 = k / 32; // uint4 per native row
    const unsigned lane = (unsigned)(o % 64);
    const size_t tile = o / 64;
    const unsigned kt = (unsigned)(tile % (k / kTileK));
    const unsigned nt = (unsigned)(tile / (k / kTileK));
    const unsigned r = lane % 16, g = lane / 16;
    return (size_t)(nt * 16 + r) * row_u4 + kt * 4 + g;
}
// packed scale item `o` (u16 of two e4m3 for NV, one e8m0 byte for MX) <- native item index
PETIT_HD size_t scale_src_item(size_t o, unsigned k, unsigned ks) {
    const unsigned row_items = k / 32;
    const unsigned spans = k / (kTileK * ks);
    const unsigned t = (unsigned)(o % ks);
    const size_t rec = o / ks;
    const unsigned lane = (unsigned)(rec % 64);
    const size_t sp_idx = rec / 64;
    const unsigned sp = (unsigned)(sp_idx % spans);
    const unsigned nt = (unsigned)(sp_idx / spans);
    const unsigned r = lane % 16, g = lane / 16;
    // NV: groups 8*(ks*sp+t) + 2g, +1 -> u16 index 4*(ks*sp+t) + g;  MX: block 4*(ks*sp+t) + g
    return (size_t)(nt * 16 + r) * row_items + 4 * (ks * sp + t) + g;
}

// One thread -> one packed uint4 (a lane's 32 k of one weight row).
__global__ __launch_bounds__(256) void repack_weights_kernel(
    uint4 *__restrict__ out, const uint4 *__restrict__ in, unsigned n, unsigned k) {
    const size_t total = (size_t)n * k / 32; // uint4 count
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total;
         o += (size_t)gridDim.x * blockDim.x)
        out[o] = in[weight_src_u4(o, k)];
}

// One thread -> one u16 of a span record: groups (2g, 2g+1) of one tile.
__global__ __launch_bounds__(256) void repack_nvscales_kernel(
    uint16_t *__restrict__ out, const uint16_t *__restrict__ in, unsigned n,
    unsigned k, unsigned ks) {
    const size_t total = (size_t)n * k / 32; // u16 count
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total;
         o += (size_t)gridDim.x * blockDim.x)
        out[o] = in[scale_src_item(o, k, ks)];
}

// One thread -> one byte of a span record.
__global__ __launch_bounds__(256) void repack_mxscales_kernel(
    uint8_t *__restrict__ out, const uint8_t *__restrict__ in, unsigned n,
    unsigned k, unsigned ks) {
    const size_t total = (size_t)n * k / 32;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total;
         o += (size_t)gridDim.x * blockDim.x)
        out[o] = in[scale_src_item(o, k, ks)];
}

static unsigned grid_for(size_t items) {
    size_t blocks = (items + 255) / 256;
    const size_t cap = 256 * 8; // 256 CUs x 8 blocks, grid-stride the rest
    return (unsigned)(blocks < cap ? (blocks ? blocks : 1) : cap);
}

int repack_weights(void *out, const void *in, unsigned k, unsigned n, hipStream_t stream) {
    if (n == 0 || k == 0)
        return kOk;
    if (n % kTileN || k % kTileK)
        return kErrProblemShape;
    const size_t items = (size_t)n * k / 32;
    hipLaunchKernelGGL(repack_weights_kernel, dim3(grid_for(items)), dim3(256), 0, stream,
                       (uint4 *)out, (const uint4 *)in, n, k);
    return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

int repack_nvscales(void *out, const void *in, unsigned k, unsigned n, hipStream_t stream) {
    if (n == 0 || k == 0)
        return kOk;
    if (n % kTileN || k % 256)
        return kErrProblemShape;
    const size_t items = (size_t)n * k / 32;
    hipLaunchKernelGGL(repack_nvscales_kernel, dim3(grid_for(items)), dim3(256), 0, stream,
                       (uint16_t *)out, (const uint16_t *)in, n, k,
                       (unsigned)span_tiles_for_k(k));
    return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

int repack_mxscales(void *out, const void *in, unsigned k, unsigned n, hipStream_t stream) {
    if (n == 0 || k == 0)
        return kOk;
    if (n % kTileN || k % 256)
        return kErrProblemShape;
    const size_t items = (size_t)n * k / 32;
    hipLaunchKernelGGL(repack_mxscales_kernel, dim3(grid_for(items)), dim3(256), 0, stream,
                       (uint8_t *)out, (const uint8_t *)in, n, k,
                       (unsigned)span_tiles_for_k(k));
    return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

// --- host twins: offline checkpoint conversion on the CPU -------------
// Same gather, plain loops over host memory; no GPU work is enqueued.  Not a fallback of the GEMM:
// the packed tensors they produce are consumed by the GPU kernels only.
int repack_weights_host(void *out, const void *in, unsigned k, unsigned n) {
    if (n == 0 || k == 0)
        return kOk;
    if (n % kTileN || k % kTileK)
        return kErrProblemShape;
    const size_t total = (size_t)n * k / 32;
    uint4 *o4 = (uint4 *)out;
    const uint4 *i4 = (const uint4 *)in;
    for (size_t o = 0; o < total; ++o)
        o4[o] = i4[weight_src_u4(o, k)];
    return kOk;
}

int repack_nvscales_host(void *out, const void *in, unsigned k, unsigned n) {
    if (n == 0 || k == 0)
        return kOk;
    if (n % kTileN || k % 256)
        return kErrProblemShape;
    const size_t total = (size_t)n * k / 32;
    const unsigned ks = (unsigned)span_tiles_for_k(k);
    for (size_t o = 0; o < total; ++o)
        ((uint16_t *)out)[o] = ((const uint16_t *)in)[scale_src_item(o, k, ks)];
    return kOk;
}

int repack_mxscales_host(void *out, const void *in, unsigned k, unsigned n) {
    if (n == 0 || k == 0)
        return kOk;
    if (n % kTileN || k % 256)
        return kErrProblemShape;
    const size_t total = (size_t)n * k / 32;
    const unsigned ks = (unsigned)span_tiles_for_k(k);
    for (size_t o = 0; o < total; ++o)
        ((uint8_t *)out)[o] = ((const uint8_t *)in)[scale_src_item(o, k, ks)];
    return kOk;
}

} // namespace petit_amd
