// nvnative.hip -- the MFMA-native image of NVFP4 weights ("petit-cdna4-nv6/1", layout.h), built ONCE at load time from the packed
// tensors the exact kernels read (petit_repack_nvfp4_weights / _scales), consumed by the WF = 6 instances of gemm_native32.hpp.
//
// Why an image and not a conversion inside the GEMM (VERDICT r05 item 1 asked for the in-kernel form first): NVFP4 is
//   w[n][k] = fp4(q[n][k]) x e4m3(s[n][k / 16]) x global_scale
// and the block-scaled MFMA takes ONE power-of-two scale per 32 k.  The e4m3 mantissa has to be multiplied into the elements, which the hardware
// converts cannot do (v_cvt_scalef32_pk_*_fp4 reads the scale's exponent only): fp4 -> f32 (1 op / 2 weights), x scale (1 / 2), -> fp6 (the
// 32-wide convert: ~1 / 16) plus the block maximum and the exponent arithmetic -- the ISA of nv6_image_kernel below is ~1.3 VALU lane-ops per weight.
// A 256 x 256 workgroup tile would spend ~340 wave-instructions per SIMD and k-tile on it next to 1024 (FP4 rate) / 2048 (FP8 rate) cycles of
// MFMA -- 0.7-1.3 of the MFMA time at ~4-8 cycles per instruction, under the same 1400 W cap that already holds the exact kernels (12 VALU per
// 8 weights, per WAVE) at 0.46 of the bf16 peak; gemm_shared.hpp is that organisation (unpack once per workgroup into LDS) for bf16 and gains
// only on gate_up.  Prefill is compute-bound, so the bytes of the image cost nothing there, and 288 GB of HBM3E hold it easily
// (Llama-3-70B at TP = 8: 4.9 GB packed NVFP4 + 6.8 GB image per GPU); decode (M <= 16) keeps reading the 4.5-bit packed tensors.
//
// Encoding of one 32-k block of one weight row (v_e = fp4 x e4m3, exact in f32):
//   E = floor(log2(max |v_e|)) - 2,  scale byte = E + 127 (127 for an all-zero block),  element_e = RNE_e2m3(v_e / 2^E)  (hardware:
//   v_cvt_scalef32_2xpk16_fp6_f32; host twin below, bit-identical).  The block maximum lands in [4, 8) and every product of an fp4
//   significand {1, 1.5} with an e4m3 significand {1 .. 1.875} below 8 is <= 7.5: the maximum never saturates.  What is lost: fp4 x e4m3 has
//   up to 6 significant bits, e2m3 keeps 4 (and a group whose own scale is 2^-d of the block's larger one keeps 4 - d): on weights quantised
//   by the checkpoint recipe 22-38 % of the elements move, by 2.3 % rms of the weight -- the weight's total quantisation error goes from 9.51 %
//   to 9.79 % of its rms (profiles/r06_nv6_reencode.md; FP8 e4m3 elements: 9.73 %, at 8 bits and the FP8 rate for every activation format).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>

#include "device_common.hpp"
#include "layout.h"
#include "petit_internal.h"

namespace petit_amd {

namespace {

PETIT_HD float fp4_value(unsigned code) {
    const unsigned mag = code & 7u;
    const float v = mag == 0 ? 0.f : mag == 1 ? 0.5f : mag == 2 ? 1.f : mag == 3 ? 1.5f : mag == 4 ? 2.f : mag == 5 ? 3.f : mag == 6 ? 4.f : 6.f;
    return (code & 8u) ? -v : v;
}
// e4m3fn byte -> f32 (0x7f / 0xff are NaN; subnormals 2^-9 .. 7 x 2^-9)
PETIT_HD float e4m3_value(unsigned b, bool *nan) {
    const unsigned e = (b >> 3) & 15u, m = b & 7u;
    *nan = (b & 0x7fu) == 0x7fu;
    float v;
    if (e == 0) {
        v = (float)m * (1.0f / 512.0f);
    } else {
        const unsigned bits = ((e + 120u) << 23) | (m << 20);
#if defined(__HIP_DEVICE_COMPILE__)
        v = __builtin_bit_cast(float, bits);
#else
        memcpy(&v, &bits, 4);
#endif
    }
    return (b & 0x80u) ? -v : v;
}
PETIT_HD unsigned f32_bits(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bit_cast(unsigned, x);
#else
    unsigned u;
    memcpy(&u, &x, 4);
    return u;
#endif
}
PETIT_HD float bits_f32(unsigned u) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bit_cast(float, u);
#else
    float x;
    memcpy(&x, &u, 4);
    return x;
#endif
}

// the block's values and its E8M0 byte from the lane's 16 bytes of nibbles (word j nibble i = element 8 j + i) and its two group scales
PETIT_HD unsigned block_values(const unsigned w[4], unsigned s_lo, unsigned s_hi, float v[32]) {
    bool nan_lo, nan_hi;
    const float sl = e4m3_value(s_lo, &nan_lo), sh = e4m3_value(s_hi, &nan_hi);
    float amax = 0.f;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 8; ++i) {
            const float x = fp4_value((w[j] >> (4 * i)) & 15u) * (j < 2 ? sl : sh);
            v[8 * j + i] = x;
            const float ax = x < 0.f ? -x : x;
            amax = ax > amax ? ax : amax;
        }
    if (nan_lo || nan_hi) { // outside the class's contract (the reference: "scales positive", README.md:25): the block reads as NaN-scaled zeros
        for (int e = 0; e < 32; ++e)
            v[e] = 0.f;
        return 0xffu;
    }
    if (amax == 0.f)
        return 127u;
    const unsigned ebits = (f32_bits(amax) >> 23) & 0xffu; // (amax >= 2^-10: never an f32 subnormal)
    return ebits - 2u;                                     // in 115 .. 136 for every e4m3 scale
}

// host twin of the hardware convert: RNE onto the e2m3 grid (subnormal step 1/8, normals 1 .. 7.5), saturating; the sign of zero is kept
inline unsigned e2m3_rne_host(float x) {
    const unsigned sign = std::signbit(x) ? 32u : 0u;
    const float ax = std::fabs(x);
    if (ax >= 7.5f)
        return sign | 31u;
    if (ax < 1.0f)
        return sign | (unsigned)std::nearbyint(ax * 8.0f); // (8 = code of 1.0)
    int e;
    (void)std::frexp(ax, &e); // ax = f 2^e, f in [0.5, 1)
    e -= 1;                   // floor(log2 ax): 0 .. 2
    const float step = std::ldexp(1.0f, e) / 8.0f;
    unsigned r = (unsigned)std::nearbyint(ax / step); // 8 .. 16
    if (r == 16)
        r = 8, e += 1;
    return sign | ((unsigned)(e + 1) << 3) | (r - 8u);
}

} // namespace

// One thread = one 32-k block of one weight row; consecutive threads = consecutive rows of an n32-block (16-byte stores side by side).
__global__ __launch_bounds__(256) void nv6_image_kernel(unsigned char *__restrict__ img, const u32x4 *__restrict__ pw, const unsigned char *__restrict__ ps,
                                                        unsigned n, unsigned k) {
    const unsigned ktiles = k / kTileK;
    const size_t total = (size_t)((n + 31) / 32) * ktiles * 4 * 32;
    unsigned char *const sc = img + nv6_elem_bytes(n, k);
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
        const unsigned r32 = (unsigned)(o % 32), b4 = (unsigned)((o / 32) % 4);
        const size_t tile = o / 128;
        const unsigned kt = (unsigned)(tile % ktiles), blk32 = (unsigned)(tile / ktiles);
        const unsigned row = blk32 * 32 + r32, blk = 4 * kt + b4, q = b4 >> 1, h = b4 & 1u;
        unsigned words[6] = {0u, 0u, 0u, 0u, 0u, 0u}, sbyte = 127u;
        if (row < n) {
            const u32x4 raw = pw[packed_weight_word_index(k, row, 4 * blk) / 4];
            const unsigned w[4] = {raw[0], raw[1], raw[2], raw[3]};
            const size_t si = packed_nvscale_byte_index(k, row, 2 * blk);
            float v[32];
            sbyte = block_values(w, ps[si], ps[si + 1], v);
#if defined(__HIP_DEVICE_COMPILE__)
            f32x16v ea, eb; // element 2 t = ea[t], 2 t + 1 = eb[t] (tools/probes/mfma32_fp6_probe.hip)
#pragma unroll
            for (int t = 0; t < 16; ++t)
                ea[t] = v[2 * t], eb[t] = v[2 * t + 1];
            const float scale = bits_f32((sbyte == 0xffu ? 127u : sbyte) << 23);
            const u32x6 qv = cvt_2xpk16_fp6_f32(ea, eb, scale); // (early-clobber form: device_common.hpp)
#pragma unroll
            for (int d = 0; d < 6; ++d)
                words[d] = qv[d];
#endif
        }
        const unsigned lane_img = h * 32 + r32;
        unsigned char *const tile_base = img + ((size_t)blk32 * ktiles + kt) * kNv6TileBytes;
        *reinterpret_cast<u32x4 *>(tile_base + q * 1024 + lane_img * 16) = u32x4{words[0], words[1], words[2], words[3]};
        uint2 tail;
        tail.x = words[4], tail.y = words[5];
        *reinterpret_cast<uint2 *>(tile_base + 2048 + lane_img * 16 + q * 8) = tail;
        sc[nv6_scale_offset(k, row, blk)] = (unsigned char)sbyte;
    }
}

int nv6_image(void *image, const void *pw, const void *ps, unsigned n, unsigned k, hipStream_t stream) {
    if (n == 0 || k == 0)
        return kOk;
    if (n % kTileN || k % 256)
        return kErrProblemShape;
    if (nv6_elem_bytes(n, k) >= (1ull << 32))
        return kErrProblemShape; // (the kernels address the element part through one 32-bit buffer descriptor)
    const size_t items = (size_t)((n + 31) / 32) * (k / kTileK) * 128;
    size_t blocks = (items + 255) / 256;
    if (blocks > 256 * 16)
        blocks = 256 * 16;
    hipLaunchKernelGGL(nv6_image_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (unsigned char *)image, (const u32x4 *)pw,
                       (const unsigned char *)ps, n, k);
    return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

// host twin (offline conversion; bit-identical to the device kernel: tests/test_gpu_parity.py::test_nv6_image_device_equals_host)
int nv6_image_host(void *image, const void *pw_, const void *ps_, unsigned n, unsigned k) {
    if (n == 0 || k == 0)
        return kOk;
    if (n % kTileN || k % 256)
        return kErrProblemShape;
    if (nv6_elem_bytes(n, k) >= (1ull << 32))
        return kErrProblemShape;
    unsigned char *const img = (unsigned char *)image;
    const unsigned *const pw = (const unsigned *)pw_;
    const unsigned char *const ps = (const unsigned char *)ps_;
    unsigned char *const sc = img + nv6_elem_bytes(n, k);
    memset(img, 0, nv6_elem_bytes(n, k));
    const unsigned n32 = (n + 31) / 32 * 32;
    for (unsigned row = 0; row < n32; ++row)
        for (unsigned blk = 0; blk < k / 32; ++blk) {
            const unsigned kt = blk / 4, q = (blk % 4) >> 1, h = blk & 1u;
            unsigned words[6] = {0u, 0u, 0u, 0u, 0u, 0u}, sbyte = 127u;
            if (row < n) {
                const unsigned *const w = pw + packed_weight_word_index(k, row, 4 * blk);
                const size_t si = packed_nvscale_byte_index(k, row, 2 * blk);
                float v[32];
                sbyte = block_values(w, ps[si], ps[si + 1], v);
                const float inv = std::ldexp(1.0f, 127 - (int)(sbyte == 0xffu ? 127u : sbyte));
                for (int e = 0; e < 32; ++e) {
                    const unsigned code = e2m3_rne_host(v[e] * inv);
                    const unsigned bit = 6u * (unsigned)e;
                    const uint64_t bits = (uint64_t)code << (bit % 32);
                    words[bit / 32] |= (unsigned)bits;
                    if (bit % 32 > 26)
                        words[bit / 32 + 1] |= (unsigned)(bits >> 32);
                }
            }
            unsigned char *const tile_base = img + ((size_t)(row / 32) * (k / kTileK) + kt) * kNv6TileBytes;
            const unsigned lane_img = h * 32 + row % 32;
            memcpy(tile_base + q * 1024 + lane_img * 16, words, 16);
            memcpy(tile_base + 2048 + lane_img * 16 + q * 8, words + 4, 8);
            sc[nv6_scale_offset(k, row, blk)] = (unsigned char)sbyte;
        }
    return kOk;
}

// Dense expansion of an image (test / debug aid, host): out[n][k] f32 = e2m3(element) x 2^(scale byte - 127), WITHOUT the global scale
int nv6_image_dequant_host(float *out, const void *image, unsigned n, unsigned k) {
    if (n % kTileN || k % 256)
        return kErrProblemShape;
    const unsigned char *const img = (const unsigned char *)image;
    const unsigned char *const sc = img + nv6_elem_bytes(n, k);
    for (unsigned row = 0; row < n; ++row)
        for (unsigned blk = 0; blk < k / 32; ++blk) {
            const unsigned kt = blk / 4, q = (blk % 4) >> 1, h = blk & 1u;
            const unsigned char *const tile_base = img + ((size_t)(row / 32) * (k / kTileK) + kt) * kNv6TileBytes;
            const unsigned lane_img = h * 32 + row % 32;
            unsigned words[6];
            memcpy(words, tile_base + q * 1024 + lane_img * 16, 16);
            memcpy(words + 4, tile_base + 2048 + lane_img * 16 + q * 8, 8);
            const unsigned sb = sc[nv6_scale_offset(k, row, blk)];
            const float scale = sb == 0xffu ? NAN : std::ldexp(1.0f, (int)sb - 127);
            for (unsigned e = 0; e < 32; ++e) {
                const unsigned bit = 6 * e;
                uint64_t wv = words[bit / 32];
                if (bit / 32 + 1 < 6)
                    wv |= (uint64_t)words[bit / 32 + 1] << 32;
                const unsigned c = (unsigned)(wv >> (bit % 32)) & 63u, ee = (c >> 3) & 3u, mm = c & 7u;
                const float mag = ee ? std::ldexp(1.0f + mm / 8.0f, (int)ee - 1) : mm / 8.0f;
                out[(size_t)row * k + 32 * blk + e] = ((c & 32u) ? -mag : mag) * scale;
            }
        }
    return kOk;
}

} // namespace petit_amd
