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

// --- ingest of tensors packed by the REFERENCE build (host side, offline) --------------------------------------------
// A checkpoint that was already repacked by the reference wheel (RepackNvFp4ToPetitFp4Weights / ...Scales,
// quantization_utils.cu:183-304,729-773) can be converted to this build's layout without going back to the native
// tensors.  The reference's packed formats, as its kernels define them:
//   weights  uint4 [K/64][N/32][64 lanes], lane = ((k%64)/16)*16 + n%16, word = 2*((n%32)/16) + (k%16)/8   (:20-87,208-253),
//            every u32 re-encoded by PetitFormat (:183-206): nibbles 0..3 at the 0x8e positions of bytes [1,3,0,2]
//            (sign at bit 15,31,7,23, magnitude three bits below), nibbles 4..7 bit-reversed into the 0x71 positions
//            (sign at bit 16,0,24,8), -0 stored as +0;
//   NV scales u16 [K/64][N/32][64], same lane, low byte = row n%32 < 16, high byte = the row 16 below; every e4m3 byte
//            rewritten as "e5m3" = fp16 bits of (scale * 2^7) >> 7 (:143-162);
//   MX scales u16 [K/64][N/32][32], index = ((k%64)/32)*16 + n%16, low/high byte as above, raw e8m0 (:165-181).
// Weights need n % 32 == 0, k % 128 == 0; NV scales n % 64 == 0 (the reference's launcher tiles 64 x 64, :755), MX n % 32.
namespace {

inline uint32_t unpetit_word(uint32_t r) { // inverse of PetitFormat: back to 8 raw E2M1 nibbles
    static const int sign_lo[4] = {15, 31, 7, 23}, sign_hi[4] = {16, 0, 24, 8};
    uint32_t v = 0;
    for (int i = 0; i < 4; ++i) {
        const uint32_t sgn = (r >> sign_lo[i]) & 1u, mag = (r >> (sign_lo[i] - 6)) & 7u;
        v |= ((sgn << 3) | mag) << (4 * i);
    }
    for (int i = 0; i < 4; ++i) {
        const uint32_t sgn = (r >> sign_hi[i]) & 1u, rev = (r >> (sign_hi[i] + 4)) & 7u;
        const uint32_t mag = ((rev & 1u) << 2) | (rev & 2u) | ((rev >> 2) & 1u);
        v |= ((sgn << 3) | mag) << (4 * (4 + i));
    }
    return v;
}

inline uint8_t e5m3_to_e4m3(uint8_t b) { // inverse of the reference's scale transform; valid (non-negative, finite) scales only
    // b << 7 is an fp16 holding scale * 2^7: exponent field e5 = b >> 3 (bias 15), mantissa top 3 bits = b & 7
    const unsigned e5 = b >> 3, m = b & 7u;
    if (e5 == 0) // fp16 subnormal: scale * 2^7 = m/8 * 2^-14 -> scale = m * 2^-24: below e4m3's smallest subnormal (2^-9): 0
        return 0;
    const int e = (int)e5 - 15 - 7; // unbiased exponent of the scale
    if (e >= -6)                    // e4m3 normal: bias 7
        return (uint8_t)(((unsigned)(e + 7) << 3) | m);
    // e4m3 subnormal range (2^-9 .. 2^-7 * 0.875): value = (8 + m) * 2^(e-3) must be a multiple of 2^-9
    const int shift = -6 - e; // 1..3
    return (uint8_t)(((8u + m) >> shift) & 7u);
}

} // namespace

int convert_reference_weights_host(void *out, const void *in, unsigned k, unsigned n) {
    if (n == 0 || k == 0)
        return kOk;
    if (n % 32 || k % kTileK)
        return kErrProblemShape;
    const uint32_t *src = (const uint32_t *)in;
    uint32_t *dst = (uint32_t *)out;
    for (unsigned r = 0; r < n; ++r)
        for (unsigned k8 = 0; k8 < k / 8; ++k8) {
            const unsigned kk = k8 * 8;
            const size_t tile = (size_t)(kk / 64) * (n / 32) + r / 32;
            const unsigned lane = ((kk % 64) / 16) * 16 + r % 16, word = 2 * ((r % 32) / 16) + (kk % 16) / 8;
            dst[packed_weight_word_index(k, r, k8)] = unpetit_word(src[(tile * 64 + lane) * 4 + word]);
        }
    return kOk;
}

int convert_reference_nvscales_host(void *out, const void *in, unsigned k, unsigned n) {
    if (n == 0 || k == 0)
        return kOk;
    if (n % 64 || k % 256)
        return kErrProblemShape;
    const uint8_t *src = (const uint8_t *)in;
    uint8_t *dst = (uint8_t *)out;
    for (unsigned r = 0; r < n; ++r)
        for (unsigned g = 0; g < k / 16; ++g) {
            const unsigned kk = g * 16;
            const size_t tile = (size_t)(kk / 64) * (n / 32) + r / 32;
            const unsigned lane = ((kk % 64) / 16) * 16 + r % 16;
            dst[packed_nvscale_byte_index(k, r, g)] = e5m3_to_e4m3(src[(tile * 64 + lane) * 2 + (r % 32) / 16]);
        }
    return kOk;
}

int convert_reference_mxscales_host(void *out, const void *in, unsigned k, unsigned n) {
    if (n == 0 || k == 0)
        return kOk;
    if (n % 32 || k % 256)
        return kErrProblemShape;
    const uint8_t *src = (const uint8_t *)in;
    uint8_t *dst = (uint8_t *)out;
    for (unsigned r = 0; r < n; ++r)
        for (unsigned b = 0; b < k / 32; ++b) {
            const unsigned kk = b * 32;
            const size_t tile = (size_t)(kk / 64) * (n / 32) + r / 32;
            const unsigned idx = ((kk % 64) / 32) * 16 + r % 16;
            dst[packed_mxscale_byte_index(k, r, b)] = src[(tile * 32 + idx) * 2 + (r % 32) / 16];
        }
    return kOk;
}

} // namespace petit_amd
