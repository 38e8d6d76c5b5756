// gemm_native32.hpp -- the native-FP4 path on the 32x32x64 block-scaled MFMA, with FP8, FP6 or FP4 activations.
//
// MXFP4 weights go RAW into v_mfma_scale_f32_32x32x64_f8f6f4 (zero unpack VALU); the 16-bit activations are quantised on
// the fly by a first small kernel to one of three block-scaled formats (one e8m0 scale per 32 k each):
//   MXFP8 (e4m3 elements): the instruction then runs at the FP8 rate, 5 PFLOP/s dense;
//   MXFP6 (e2m3 elements): the instruction runs at its FP4 rate as long as neither operand is 8-bit -- e4m3's three mantissa
//          bits over three binades (subnormals of 1/8 below) at the 10 PFLOP/s rate;
//   MXFP4 (e2m1 elements): FP4 x FP4, the rate MI355X quotes for its hardware FP4.
// OPT-IN, like gemm_native.hpp: quantising activations is a different accuracy class (e4m3 / e2m3: 2^-4 relative per element,
// e2m1: 2^-2), never chosen by solution_id = -1; ids carry mfma_type 2 (FP8 activations), 4 (FP6) or 6 (FP4).
// Exact-semantics + stated-tolerance tests: tests/test_gpu_parity.py.
//
// What differs from gemm_native.hpp (16x16x128, FP8 only):
//  * 32x32x64 instruction: two neighbouring n-tiles are merged in registers with two lane swaps per packed word
//    (merge_tiles, gemm_wide.hpp), P1 = the pair's 32 rows x k 0..63 of the tile, P2 = k 64..127 -- each IS the
//    instruction's FP4 operand (lane (row, h): 32 consecutive k), and the merged span-record byte IS its per-lane
//    E8M0 scale.  Half as many MFMAs per flop, half as many activation-fragment reads per flop.
//  * a k-step is two groups (P1, P2); a group issues MB*NP MFMAs on MB*NP different accumulators while the fragments of
//    the next group are read from LDS (no MFMA waits for the previous one, no fragment is waited for right after
//    its request -- the 16x16 kernel did both and sat at 0.30 of the FP8 rate).
// Operand layouts probed on gfx950 (tools/probes/mfma32_layout_probe.hip):
//   FP4 operand (A and B)  lane (row|col = l%32, h = l/32): regs 0-3, k = 32h + 8*reg + nibble   (natural)
//   FP8 operand (B)        lane (col, h): regs 0-3 k = 16h .. 16h+15, regs 4-7 k = 32 + 16h .. 32+16h+15
//   FP6 operand (B)        lane (col, h): 32 elements of 6 bits in regs 0-5, k = 32h + element   (tools/probes/mfma32_fp6_probe.hip)
//   scales                 the E8M0 byte of block b (k in [32b, 32b+32)) comes from lanes l/32 = b of the same row/column
#pragma once

#include "gemm_native.hpp"
#include "gemm_wide.hpp"

// measurement-only builds (tools/ablate_native32.sh): 1 = no activation DMA, 2 = no W refills, 4 = fragments read once,
// 8 = no MFMAs (operands kept alive), 16 = no output stores.  0 in every shipped library.
#ifndef PETIT_ABLATE_N32
#define PETIT_ABLATE_N32 0
#endif

namespace petit_amd {

// Workspace layout, K-TILE MAJOR: qa[K/128][M][16 ACT] bytes, then qs[K/128][M][4] E8M0 bytes -- the activation tile of a
// workgroup and k-tile (BM rows x 128 k) is ONE contiguous run, so every 1 KiB wave-load of the GEMM's direct-to-LDS staging
// reads eight whole cache lines (with the row-major form qa[M][K ACT / 8] a wave-load touched 16 half lines 4 KiB apart, and
// the activation DMA cost more than the weight stream: 8.9 vs 4.8 us of a 30.6 us launch at 8192^2, M = 512 --
// tools/ablate_native32.sh).  Rows >= M of the last m-block read the next tile's rows (or zeros past the end): such rows
// only feed output columns that are never stored.
//   ACT = 8: per 128-k tile 128 bytes in the order [0-15][32-47][16-31][48-63][64-79][96-111][80-95][112-127] (16-k units),
//            so that lane (m, h) finds its P1 operand at byte 32h and its P2 operand at byte 64 + 32h.
//   ACT = 4: per 128-k tile 64 bytes, natural nibble order: P1 operand at byte 16h, P2 at 32 + 16h.
//   ACT = 6 (MXFP6, e2m3 elements: the instruction still runs at the FP4 rate, the elements carry e4m3's three mantissa bits): a 32-k block
//            is 24 bytes = the 6 operand registers of one lane (element i at bits [6 i, 6 i + 6), tools/probes/mfma32_fp6_probe.hip).  Stored as
//            TWO images so that every access stays a 16-byte one: qa_lo[K/128][M][64] holds registers 0-3 of block b at byte 16 b (the FP4
//            geometry), qa_hi[K/128][M][32] registers 4-5 of blocks 0, 2, 1, 3 in that order (lane (m, h) reads ONE 16-byte unit h and finds
//            the tail of its P1 block in the first 8 bytes, of its P2 block in the last 8); then the scales as above.
template <int ACT> __host__ __device__ inline size_t native32_ws_bytes(unsigned m, unsigned k) {
    return (size_t)m * (k / 8 * ACT) + (size_t)m * (k / 32);
}

// One thread = 8 consecutive k of one row; the 4 threads of a quad share one 32-k block.
template <class AT, int ACT>
__global__ __launch_bounds__(256) void quantize_act32_kernel(const void *a, unsigned char *ws, unsigned m, unsigned k) {
    // grid = (ceil(K / 8 / (4 * 256)), M): blockIdx.y is the row -- no division in the address arithmetic -- and a thread owns
    // four 8-element columns 256 apart, all four loads requested before the first is used
    const unsigned row_bytes = k / 8 * ACT;
    unsigned char *qa = ws;
    unsigned char *qs = ws + (size_t)m * row_bytes;
    constexpr int kIlp = 4;
    const unsigned row = blockIdx.y, cols = k / 8;
    const u32x4 *const a_row = reinterpret_cast<const u32x4 *>(a) + (size_t)row * cols;
    u32x4 raws[kIlp];
#pragma unroll
    for (int j = 0; j < kIlp; ++j) {
        const unsigned c = (blockIdx.x * kIlp + j) * 256 + threadIdx.x;
        raws[j] = c < cols ? a_row[c] : u32x4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int j = 0; j < kIlp; ++j) {
        const unsigned c8 = (blockIdx.x * kIlp + j) * 256 + threadIdx.x; // 8-element column
        if (c8 >= cols) // (K / 8 is a multiple of 16: the 16 lanes of a k-tile leave together)
            break;
        const u32x4 raw = raws[j];
        float v[8];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const unsigned w = raw[d];
            if constexpr (AT::kType == kDataTypeBf16) {
                const unsigned lo = w << 16, hi = w & 0xffff0000u;
                v[2 * d] = __builtin_bit_cast(float, lo);
                v[2 * d + 1] = __builtin_bit_cast(float, hi);
            } else {
                const f16x2 p = __builtin_bit_cast(f16x2, w);
                v[2 * d] = (float)p[0];
                v[2 * d + 1] = (float)p[1];
            }
        }
        float amax = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            amax = fmaxf(amax, fabsf(v[i]));
        amax = fmaxf(amax, __shfl_xor(amax, 1));
        amax = fmaxf(amax, __shfl_xor(amax, 2));
        // E8M0 scale 2^(E - emax_elem) with E the exponent of the block maximum (OCP MX): emax_elem = 7 for e4m3 as
        // gemm_native.hpp uses it (maximum lands in [128, 256) <= 448), 2 for e2m1 (maximum in [4, 8), above 6 saturates)
        constexpr unsigned kEmax = ACT == 8 ? 7u : 2u;
        const unsigned ebits = (__builtin_bit_cast(unsigned, amax) >> 23) & 0xffu;
        unsigned sbyte = amax == 0.f ? 127u : (ebits > kEmax ? ebits - kEmax : 1u);
        sbyte = sbyte > 254u ? 254u : sbyte;
        const unsigned kt = c8 / 16, col16 = c8 % 16;
        if constexpr (ACT == 8) {
            const float inv = __builtin_bit_cast(float, (254u - sbyte) << 23); // 2^-(sbyte-127)
            int q0 = 0, q1 = 0;
            q0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[0] * inv, v[1] * inv, q0, false);
            q0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[2] * inv, v[3] * inv, q0, true);
            q1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[4] * inv, v[5] * inv, q1, false);
            q1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[6] * inv, v[7] * inv, q1, true);
            const unsigned u16 = col16 >> 1, half = col16 & 1u;
            const unsigned pos = (u16 & 4u) | ((u16 & 1u) << 1) | ((u16 >> 1) & 1u);
            uint2 o;
            o.x = (unsigned)q0, o.y = (unsigned)q1;
            *reinterpret_cast<uint2 *>(qa + ((size_t)kt * m + row) * 128 + pos * 16 + half * 8) = o;
        } else {
            // hardware RNE conversion to e2m1 with saturation: dst nibbles = cvt(src / scale)
            const float scale = __builtin_bit_cast(float, sbyte << 23); // 2^(sbyte-127)
            unsigned q = 0;
            q = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(q, v[0], v[1], scale, 0);
            q = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(q, v[2], v[3], scale, 1);
            q = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(q, v[4], v[5], scale, 2);
            q = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(q, v[6], v[7], scale, 3);
            *reinterpret_cast<unsigned *>(qa + ((size_t)kt * m + row) * 64 + col16 * 4) = q;
        }
        // the four scale bytes of a row's k-tile leave as ONE dword (byte stores are the slow path of the memory pipeline):
        // the 16 lanes of a k-tile are consecutive, the first lane of each quad holds the quad's byte
        const unsigned s1 = __shfl_down(sbyte, 4, 16), s2 = __shfl_down(sbyte, 8, 16), s3 = __shfl_down(sbyte, 12, 16);
        if (col16 == 0)
            *reinterpret_cast<unsigned *>(qs + ((size_t)kt * m + row) * 4) = sbyte | (s1 << 8) | (s2 << 16) | (s3 << 24);
    }
}

// MXFP6 (e2m3): one thread = one 32-k block of one row (the hardware converts 32 values at once: v_cvt_scalef32_pk32_fp6_{bf16,f16}, RNE of
// src / scale, saturating at 7.5).  grid = (ceil(K / 32 / 256), M); the four blocks of a k-tile are four consecutive lanes.
template <class AT>
__global__ __launch_bounds__(256) void quantize_act32_fp6_kernel(const void *a, unsigned char *ws, unsigned m, unsigned k) {
    const unsigned row = blockIdx.y, blocks = k / 32, blk = blockIdx.x * 256 + threadIdx.x;
    if (blk >= blocks) // (K / 32 is a multiple of 4: the lanes of a k-tile leave together)
        return;
    unsigned char *const qa_lo = ws, *const qa_hi = ws + (size_t)m * (k / 2), *const qs = ws + (size_t)m * (k / 8 * 6);
    const u32x4 *const src = reinterpret_cast<const u32x4 *>(a) + ((size_t)row * blocks + blk) * 4;
    u32x4 raw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        raw[j] = src[j];
    // block maximum on the 16-bit patterns (sign cleared: bf16 and fp16 magnitudes order like their bit patterns), exponent from its f32 value
    unsigned mx = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const unsigned w = raw[j][d] & 0x7fff7fffu;
            const unsigned lo = w & 0xffffu, hi = w >> 16;
            mx = mx > lo ? mx : lo;
            mx = mx > hi ? mx : hi;
        }
    float amax;
    if constexpr (AT::kType == kDataTypeBf16) {
        amax = __builtin_bit_cast(float, mx << 16);
    } else {
        const unsigned short hb = (unsigned short)mx;
        amax = (float)__builtin_bit_cast(_Float16, hb);
    }
    // OCP MX scale 2^(E - emax_elem), emax_elem = 2 for e2m3 (largest normal 7.5 = 1.875 x 2^2): the maximum lands in [4, 8), above 7.5 saturates
    const unsigned ebits = (__builtin_bit_cast(unsigned, amax) >> 23) & 0xffu;
    unsigned sbyte = amax == 0.f ? 127u : (ebits > 2u ? ebits - 2u : 1u);
    sbyte = sbyte > 254u ? 254u : sbyte;
    const float scale = __builtin_bit_cast(float, sbyte << 23);
    const u32x16 packed = u32x16{raw[0][0], raw[0][1], raw[0][2], raw[0][3], raw[1][0], raw[1][1], raw[1][2], raw[1][3],
                                 raw[2][0], raw[2][1], raw[2][2], raw[2][3], raw[3][0], raw[3][1], raw[3][2], raw[3][3]};
    const u32x6 q = cvt_pk32_fp6_16bit<AT::kType == kDataTypeBf16>(packed, scale); // (early-clobber form: device_common.hpp)
    const unsigned kt = blk / 4, b = blk % 4;
    const size_t tile_row = (size_t)kt * m + row;
    *reinterpret_cast<u32x4 *>(qa_lo + tile_row * 64 + 16 * b) = u32x4{q[0], q[1], q[2], q[3]};
    uint2 tail;
    tail.x = q[4], tail.y = q[5];
    *reinterpret_cast<uint2 *>(qa_hi + tile_row * 32 + 8 * ((b & 1u) * 2 + (b >> 1))) = tail;
    // the four scale bytes of a row's k-tile leave as one dword (the first lane of the four collects them)
    const unsigned s1 = __shfl_down(sbyte, 1, 4), s2 = __shfl_down(sbyte, 2, 4), s3 = __shfl_down(sbyte, 3, 4);
    if (b == 0)
        *reinterpret_cast<unsigned *>(qs + tile_row * 4) = sbyte | (s1 << 8) | (s2 << 16) | (s3 << 24);
}

// The SiLU-mul epilogue as a PRODUCER of quantised activations (out_format 8 / 4; 128 x 256 workgroup tiles only: NP = 2, four waves).
// A gated MLP is gate_up -> SiLU-mul -> down; with 16-bit hand-over the native path pays a quantiser launch in front of each GEMM
// (5-6 us each at M = 512, 15-19 % of `o`).  Here the gate_up kernel writes what `down`'s kernel reads: out[m][j], j < N/2, quantised
// per 32 columns to MXFP8 / MXFP4 with the quantiser's own rule (E8M0 scale from the block maximum), in the k-tile-major scratch
// layout described above -- a 256-column workgroup tile produces exactly ONE 128-column k-tile of the consumer for its 128 rows, a
// contiguous 8 / 16 KiB run plus 512 scale bytes.  The values are quantised from the f32 SiLU-mul result (one rounding, where
// "round to 16 bit, then quantise" has two).
//   lane (m, h) of wave wn holds, for each of the wave's two output tiles np, columns {0-3, 8-11} (h = 0) or {4-7, 12-15} (h = 1): a
//   32-column block is the wave's two tiles of one row, i.e. this lane's 16 values and its partner's (lane ^ 32) 16.  The block
//   maximum needs one cross-half swap; a second swap (of the packed codes) leaves lane h with ALL 16 columns of tile np = h, so each
//   lane stores 8 (FP4) / 16 (FP8) contiguous bytes.
template <class AT, int OUTF, int MB>
__device__ __forceinline__ void n32_silu_quant_epilogue(const f32x16 (&acc)[MB][2], const float gs, const void *bias, unsigned char *out,
                                                        const unsigned m_total, const unsigned n_half, const unsigned m_first,
                                                        const unsigned col0, const unsigned kt, const unsigned wn, const unsigned m_l,
                                                        const unsigned h, unsigned char *lds_scales, const unsigned m0, const unsigned tid) {
    constexpr unsigned kRowB = OUTF == 6 ? 64 : 16 * OUTF; // bytes per row and k-tile: 128 (FP8) / 64 (FP4; FP6: of the image of registers 0-3)
    unsigned char *const qs = out + (size_t)m_total * (n_half / 8 * OUTF);
    unsigned char *const out_hi = out + (size_t)m_total * (n_half / 2); // FP6: the image of registers 4-5 (native32_ws_bytes)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const unsigned m = m_first + mb * 32;
        f32x4 v[2][2];
        float amax = 0.f;
#pragma unroll
        for (int np = 0; np < 2; ++np)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const f32x16 &a = acc[mb][np];
                v[np][u] = silu_mul4<AT>(f32x4{a[4 * u], a[4 * u + 1], a[4 * u + 2], a[4 * u + 3]},
                                         f32x4{a[8 + 4 * u], a[8 + 4 * u + 1], a[8 + 4 * u + 2], a[8 + 4 * u + 3]}, gs, bias,
                                         col0 + np * 16 + u * 8 + 4 * h, n_half);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    amax = fmaxf(amax, fabsf(v[np][u][i]));
            }
        {
            const unsigned au = __builtin_bit_cast(unsigned, amax);
            const auto sw = __builtin_amdgcn_permlane32_swap(au, au, false, false);
            const unsigned s0 = sw[0], s1 = sw[1];
            amax = fmaxf(__builtin_bit_cast(float, s0), __builtin_bit_cast(float, s1));
        }
        // the quantiser's rule (quantize_act32_kernel): E8M0 scale 2^(E - emax_elem), E = exponent of the block maximum
        constexpr unsigned kEmax = OUTF == 8 ? 7u : 2u;
        const unsigned ebits = (__builtin_bit_cast(unsigned, amax) >> 23) & 0xffu;
        unsigned sbyte = amax == 0.f ? 127u : (ebits > kEmax ? ebits - kEmax : 1u);
        sbyte = sbyte > 254u ? 254u : sbyte;
        if (h == 0)
            lds_scales[(mb * 32 + m_l) * 4 + wn] = (unsigned char)sbyte;
        if constexpr (OUTF == 6) {
            // the block's 32 columns are this lane's 16 and its partner's (lane ^ 32) 16: each converts a 32-wide vector that holds its own values at
            // their column positions and zeros (code 0) elsewhere; the two results OR into the block's 6 registers.  v_cvt_scalef32_2xpk16_fp6_f32
            // interleaves its sources (element 2 t = a[t], 2 t + 1 = b[t]: tools/probes/mfma32_fp6_probe.hip); column np 16 + u 8 + 4 h + i.
            const float scale = __builtin_bit_cast(float, sbyte << 23);
            f32x16v ea, eb;
#pragma unroll
            for (int t = 0; t < 16; ++t)
                ea[t] = 0.f, eb[t] = 0.f;
#pragma unroll
            for (int np = 0; np < 2; ++np)
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) { // (static indices: both halves written, the foreign one with zeros)
                        const bool mine = h == (unsigned)hh;
                        const int t = np * 8 + u * 4 + 2 * hh;
                        ea[t] = mine ? v[np][u][0] : ea[t], eb[t] = mine ? v[np][u][1] : eb[t];
                        ea[t + 1] = mine ? v[np][u][2] : ea[t + 1], eb[t + 1] = mine ? v[np][u][3] : eb[t + 1];
                    }
            const u32x6 own = cvt_2xpk16_fp6_f32(ea, eb, scale); // (early-clobber form: device_common.hpp)
            unsigned full[6];
#pragma unroll
            for (int d = 0; d < 6; ++d) {
                const unsigned x = own[d];
                const auto sw = __builtin_amdgcn_permlane32_swap(x, x, false, false);
                const unsigned p0 = sw[0], p1 = sw[1];
                full[d] = p0 | p1;
            }
            if (m < m_total) {
                if (h == 0) {
                    *reinterpret_cast<u32x4 *>(out + ((size_t)kt * m_total + m) * 64 + 16 * wn) = u32x4{full[0], full[1], full[2], full[3]};
                } else {
                    uint2 tail;
                    tail.x = full[4], tail.y = full[5];
                    *reinterpret_cast<uint2 *>(out_hi + ((size_t)kt * m_total + m) * 32 + 8 * ((wn & 1u) * 2 + (wn >> 1))) = tail;
                }
            }
        } else if constexpr (OUTF == 4) {
            const float scale = __builtin_bit_cast(float, sbyte << 23);
            unsigned q[2];
#pragma unroll
            for (int np = 0; np < 2; ++np) {
                unsigned w = 0;
                w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(w, v[np][0][0], v[np][0][1], scale, 0);
                w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(w, v[np][0][2], v[np][0][3], scale, 1);
                w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(w, v[np][1][0], v[np][1][1], scale, 2);
                w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(w, v[np][1][2], v[np][1][3], scale, 3);
                q[np] = w;
            }
            // lane h now gets tile np = h whole: r0 = the {0-3, 8-11} bytes, r1 = the {4-7, 12-15} bytes
            const auto sw = __builtin_amdgcn_permlane32_swap(q[0], q[1], false, false);
            const unsigned r0 = sw[0], r1 = sw[1];
            uint2 o;
            o.x = __builtin_amdgcn_perm(r1, r0, 0x05040100u); // columns 0-7
            o.y = __builtin_amdgcn_perm(r1, r0, 0x07060302u); // columns 8-15
            if (m < m_total)
                *reinterpret_cast<uint2 *>(out + ((size_t)kt * m_total + m) * kRowB + 16 * wn + 8 * h) = o;
        } else {
            const float inv = __builtin_bit_cast(float, (254u - sbyte) << 23); // 2^-(sbyte - 127)
            unsigned d[2][2];
#pragma unroll
            for (int np = 0; np < 2; ++np)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    int w = 0;
                    w = __builtin_amdgcn_cvt_pk_fp8_f32(v[np][u][0] * inv, v[np][u][1] * inv, w, false);
                    w = __builtin_amdgcn_cvt_pk_fp8_f32(v[np][u][2] * inv, v[np][u][3] * inv, w, true);
                    d[np][u] = (unsigned)w;
                }
            const auto s0 = __builtin_amdgcn_permlane32_swap(d[0][0], d[1][0], false, false); // columns 0-3 | 4-7 of tile np = h
            const auto s1 = __builtin_amdgcn_permlane32_swap(d[0][1], d[1][1], false, false); // columns 8-11 | 12-15
            const unsigned a0 = s0[0], a1 = s0[1], b0 = s1[0], b1 = s1[1];
            // the 16-column unit u16 = 2 wn + h sits at position pos of the tile's 128 bytes (the operand order of the layout note)
            const unsigned u16 = 2 * wn + h, pos = (u16 & 4u) | ((u16 & 1u) << 1) | ((u16 >> 1) & 1u);
            if (m < m_total)
                *reinterpret_cast<u32x4 *>(out + ((size_t)kt * m_total + m) * kRowB + pos * 16) = u32x4{a0, a1, b0, b1};
        }
    }
    __syncthreads();
    if (tid < (unsigned)(32 * MB) && m0 + tid < m_total)
        *reinterpret_cast<unsigned *>(qs + ((size_t)kt * m_total + m0 + tid) * 4) = reinterpret_cast<const unsigned *>(lds_scales)[tid];
}

//   MB, NP, WAVES, D as in WideCfg; ACT = 8 (MXFP8 activations), 6 (MXFP6, e2m3) or 4 (MXFP4 activations).
//   KT   k-tiles per barrier ("stage"): the quantised activation tile is small (128 / 64 bytes per row and k-tile), and with
//        zero unpack work a k-tile is only 2*MB*NP MFMAs, so one barrier per tile leaves the wave waiting on it.
//   PF   stages requested ahead (NBUF = PF + 1 LDS stages): 1 = the next stage is requested at the top of a stage and waited
//        for at its end; 2 / 3 = that many ahead, in flight across the barrier (raw s_barrier + counted vmcnt).  Loads retire
//        in issue order, so the W refills can stay in flight for PF - 1 stages only: PF is also the weight ring's real depth.
//        Unlike the dequant kernels (power-limited) this kernel was latency-bound: a stage took the L2 round
//        trip (~1900 cycles) for 512 cycles of MFMA work.
//   WM   waves along M (1 or 2): the workgroup is WM x WAVES waves, every wave owns MB m32-blocks x NP n-pairs, the WM waves
//        of a column stream the SAME weight tiles (the second request hits the CU's L1).  WM = 2 puts two waves on every
//        SIMD for the same 128 x 128 workgroup tile: with one wave per SIMD the stage's DMA / LDS reads / W refills and its
//        MFMAs were issued by the same in-order instruction stream and did not overlap (0.25 MFMA utilisation measured).
//   KG   wave GROUPS along K inside the workgroup (1 or 2).  KG = 2: the workgroup is two complete copies of the KG = 1 wave set,
//        each with its own LDS stages and weight ring, walking one half of the workgroup's K range; their accumulators are summed
//        through LDS before the epilogue.  Two waves per SIMD that share NOTHING (WM = 2 shares the weight tiles and lost): what a
//        second workgroup on the CU gives the 128 x 256 kernel, for shapes whose grid is only one 128 x 128 tile per CU (N <= 10240
//        at M = 512: `o`, qkv).
//   LW   1: a LOADER wave: the activation tiles are staged by one extra wave whose queue holds nothing else; the compute waves issue weight loads
//        only and meet the loader at the stage barrier.  Built on the hypothesis that hits queue behind misses (loads retire in issue order); worth
//        4-13 % at M = 512, but the hypothesis itself did not survive the follow-up experiments (PETIT_N32_PW / PETIT_N32_LWPF below,
//        profiles/r03_native_ablation.md): a CU takes in ~60-65 GB/s of this kind of stream however it is requested.
//   WF   weight operand: 4 = MXFP4 weights RAW (the packed layout of layout.h, n-tile pairs merged in registers: everything above);
//        6 = the "petit-cdna4-nv6/1" image of NVFP4 weights (nvnative.hip: e2m3 elements + one E8M0 scale per 32 k, re-encoded ONCE at load
//        time, stored in the instruction's own 32-row operand geometry): three 16-byte loads per lane, n32-block and k-tile, no merge, no
//        unpack; the instruction runs at the rate of the ACTIVATION format (FP6 x FP6 / FP4 at the FP4 rate, x FP8 at the FP8 rate).
template <class AT_, int KS_, int MB_, int NP_, int WAVES_, int D_, int ACT_, int KT_ = 1, int PF_ = 1, int WM_ = 1, int KG_ = 1, int LW_ = 0, int WF_ = 4>
struct Native32Cfg {
    using AT = AT_;
    static constexpr int WF = WF_;
    static constexpr int kWLoads = WF_ == 6 ? 3 : 2;              // 16-byte loads per lane, n32-block and k-tile
    static_assert(WF_ == 4 || WF_ == 6, "weights: raw MXFP4 or the NV6 image");
#ifndef PETIT_N32_LWPF
#define PETIT_N32_LWPF 0
#endif
    static constexpr int kLwStageLoads = KT_ * (32 * MB_ * WM_ * ACT_ / 64 + (32 * MB_ * WM_ + 63) / 64); // wave-loads of the loader per stage (ACT = 4 / 8)
    static constexpr int kLwPfMax = 1 + 63 / kLwStageLoads;                                              // (vmcnt counts to 63)
    static constexpr int PF = (LW_ && PETIT_N32_LWPF) ? (PETIT_N32_LWPF < kLwPfMax ? PETIT_N32_LWPF : kLwPfMax) : PF_; // (experiment: deeper activation staging)
    static constexpr int KS = KS_, MB = MB_, NP = NP_, WAVES = WAVES_, D = D_, ACT = ACT_, KT = KT_, NBUF = PF + 1;
    static constexpr int WM = WM_, kWaves = WAVES * WM, KG = KG_, LW = LW_;
    static constexpr int kComputeThreads = 64 * kWaves * KG;
#ifndef PETIT_N32_PW
#define PETIT_N32_PW 0
#endif
#ifndef PETIT_N32_PWSHARE
#define PETIT_N32_PWSHARE 1
#endif
    // PW (experiment, with LW): a PREFETCH wave that touches the workgroup's weight lines PETIT_N32_PW k-tiles ahead (one dword per 128-byte line, into
    // an LDS dump slot), so that the compute waves' refills find them in L2
    static constexpr int PW = (LW_ && PETIT_N32_PW && WF_ == 4) ? 1 : 0;
    static constexpr int kThreads = kComputeThreads + 64 * (LW + PW);
    static_assert(KG == 1 || (KG == 2 && WM == 1), "two K groups only with one wave along M");
    static_assert(LW == 0 || (LW == 1 && KG == 1 && WM == 1 && PF_ >= 2), "the loader wave: one K group, stages in flight across the barrier");
    static constexpr int BM = 32 * MB * WM;
    static constexpr int kRowU4 = ACT == 6 ? 4 : ACT;         // 16-byte units per LDS row: 128 B (FP8) / 64 B (FP4; FP6: registers 0-3 of every block)
    static constexpr int kLoU4 = BM * kRowU4;                 // FP6: the image of registers 0-3 ...
    static constexpr int kHiU4 = ACT == 6 ? BM * 2 : 0;       // ... followed by the image of registers 4-5: 32 B per row
    static constexpr int kDataU4 = kLoU4 + kHiU4;             // one tile image
    static constexpr int kScaleU4 = (BM + 3) / 4 < kWaves * 16 ? kWaves * 16 : (BM + 3) / 4; // one dword per row (wave-loads of 64)
    static constexpr int kRowsPerLoad = 64 / kRowU4;          // rows one 1 KiB wave-load covers: 8 / 16
    // waves that stage the activation tile: all of them when they divide its wave-loads, else four (five-wave workgroups: 160- and 320-column
    // tiles put N = 10240 on the chip in ONE round of 256 workgroups where 128-column tiles need 1.25); the others issue the same number of
    // loads, out of range, into a dump slot behind the stage (loads retire in order: every wave counts the same)
    static constexpr int kDmaWaves = (BM / kRowsPerLoad) % kWaves == 0 ? kWaves : 4;
    static constexpr int kDumpU4 = kDmaWaves < kWaves ? 64 : 0;
    static constexpr int kDataLoads = BM / kRowsPerLoad / kDmaWaves;
    static constexpr int kHiLoads = ACT == 6 ? BM / 32 / kDmaWaves : 0; // FP6: wave-loads of the 32-byte rows (32 rows each)
    static_assert(ACT == 8 || ACT == 6 || ACT == 4, "activations are quantised to MXFP8, MXFP6 or MXFP4");
    static_assert(ACT != 6 || (BM % (32 * kDmaWaves) == 0 && LW_ == 0), "MXFP6: 128-row tiles, no loader wave");
    static_assert(KS % D == 0, "ring depth must divide the span");
    static_assert(WM == 1 || WM == 2, "one or two waves along M");
    static_assert(BM % (kRowsPerLoad * kDmaWaves) == 0 && BM <= 64 * kWaves, "A tile must split evenly over the staging waves");
    static_assert((kRowsPerLoad * kDmaWaves) % 16 == 0, "wave-loads must step by whole 16-row groups (swizzle term constant)");
    static constexpr int kStageU4 = KT * (kDataU4 + kScaleU4) + kDumpU4;   // one stage: KT tile images, then their KT scale arrays (+ dump slot)
    static constexpr int kStageLoads = KT * (kDataLoads + kHiLoads + 1); // VMEM ops one wave issues per stage
    static_assert(KS % KT == 0 && (KT == 1 || KT == 2) && PF >= 1 && (PF <= 3 || LW_), "stage = 1 or 2 k-tiles, 1 to 3 stages ahead");
    static_assert(D % KT == 0, "the W ring is refilled a stage at a time");
    static constexpr int BN = 32 * NP * WAVES;
    // 128 x 256 / 256 x 128 with a two-tile ring: asked to fit two workgroups per CU (256 registers, 128 of them accumulators):
    // 3/4 of the operand bytes per flop of 128 x 128 AND a second workgroup to overlap with (gate_up M = 512: 158 -> 144 us)
    // MXFP8 fragments are twice the size (the group-ahead double buffer of MB fragments is 64 registers + scales): that form reads its
    // fragments ONE m-block ahead instead (kLean: 18 registers; an FP8-rate MFMA pair covers the LDS latency), which is what lets the
    // 128 x 256 tile fit two workgroups per CU for MXFP8 activations too
    static constexpr bool kLean = (ACT == 8 || ACT == 6) && WM == 1 && LW_ == 0 &&
                                  ((KG == 1 && MB * NP == 8 && D == 2) || (ACT == 6 && KG == 2 && WAVES_ == 5)); // (ten waves: 168 registers each)
    static constexpr bool kWantTwoPerSimd = KG == 2 || kLean || (WM == 1 && (ACT == 4 || ACT == 6) && MB * NP == 8 && D == 2);
    static constexpr int kCTileU4 = CTile<BN>::u4(BM);             // the epilogue's image of the C tile (device_common.hpp)
    static constexpr int kRedU4 = KG == 2 ? BM * BN / 4 : 0;       // KG = 2: the second group's accumulators, f32
    static constexpr int kSmemU4a = KG * NBUF * kStageU4 > kCTileU4 ? KG * NBUF * kStageU4 : kCTileU4;
    static constexpr int kSmemU4 = (kSmemU4a > kRedU4 ? kSmemU4a : kRedU4) + 16 * PW;
    static_assert(kSmemU4 * 16 <= 160 * 1024, "LDS budget");
    // (a register budget for two workgroups per CU only where their LDS fits twice as well)
    static constexpr int kMinWavesPerSimd = (kWantTwoPerSimd && (KG == 2 || kSmemU4 * 16 <= 80 * 1024)) ? 2 : 1;
};

// W refills (wave-loads) the stage that starts at tile t_first of a span issues: the tiles whose slot is needed again
// (t_first < 0: a stage of the PREVIOUS span, which is never a last span: every tile is refilled)
constexpr int n32_stage_refills(bool last_span, int t_first, int kt, int d, int ks, int np, int wloads = 2) {
    if (t_first < 0)
        return wloads * np * kt;
    int n = 0;
    for (int t = t_first; t < t_first + kt; ++t)
        n += (!last_span || t + d < ks) ? wloads * np : 0;
    return n;
}

template <class Cfg>
__global__ __launch_bounds__(Cfg::kThreads, Cfg::kMinWavesPerSimd) void gemm_native32_kernel(const GemmArgs p, const unsigned char *ws) {
    using AT = typename Cfg::AT;
    constexpr int KS = Cfg::KS, MB = Cfg::MB, NP = Cfg::NP, WAVES = Cfg::WAVES, D = Cfg::D, ACT = Cfg::ACT;
    constexpr int KT = Cfg::KT, PF = Cfg::PF, NBUF = Cfg::NBUF, WM = Cfg::WM, kWaves = Cfg::kWaves;
    constexpr unsigned kRecBytes = ScaleRec<kFmtMx, KS>::kBytes;
    constexpr int kRecDw = ScaleRec<kFmtMx, KS>::kDwords;
    constexpr unsigned kOob = 0x80000000u;
    constexpr int kFragU4 = ACT == 4 ? 1 : 2; // 16-byte units of one activation operand (FP6: registers 0-3, and the unit that holds 4-5 of P1 and P2)
    constexpr int kBlgp = ACT == 8 ? 0 : ACT == 6 ? 2 : 4; // the activation operand's format: e4m3 / e2m3 / e2m1
    constexpr int WF = Cfg::WF, kWL = Cfg::kWLoads;
    constexpr int kCbsz = WF == 6 ? 2 : 4;                  // the weight operand's format: e2m3 (the NV6 image) / e2m1 (raw MXFP4)

    // NBUF stages of [KT tile images][KT scale arrays]
    __shared__ u32x4 smem[Cfg::kSmemU4];

    constexpr int KG = Cfg::KG;
    const unsigned tid = threadIdx.x;
    const unsigned lane = tid & 63u;
    const unsigned wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned kg = KG == 1 ? 0u : wave_all / kWaves;          // K group of this wave
    const unsigned wave = KG == 1 ? wave_all : wave_all % kWaves;  // index inside the group
    const unsigned wn = WM == 1 ? wave : wave % WAVES, wm = WM == 1 ? 0u : wave / WAVES; // (the WM waves of a column: wave, wave + WAVES)
    const unsigned m_l = lane & 31u, h = lane >> 5;
    u32x4 *const smem_g = smem + kg * (NBUF * Cfg::kStageU4);      // this group's LDS stages

    const unsigned ktiles = p.k / kTileK;
    const unsigned nspans = ktiles / KS;
    const unsigned ntiles = p.n / kTileN;
    unsigned bn, bm;
    tile_of_block(p.flags, bn, bm);
    const unsigned nt0 = (bn * WAVES + wn) * (2 * NP);
    const unsigned m0 = bm * Cfg::BM;
    // K slice of this wave group: part blockIdx.z * KG + kg of spans_per_wave spans each.  A part past the end (possible for the
    // LAST group of a KG = 2 workgroup only) walks the last span with every weight row masked off (valid_nt = 0: zeros in, zeros
    // accumulated) instead of branching around the MFMA stream, and then keeps the barrier count of its partner.
    const unsigned part_begin = (blockIdx.z * KG + kg) * p.spans_per_wave;
    const bool empty_part = KG == 2 && part_begin >= nspans;
    const unsigned sp_begin = min(part_begin, nspans - 1);
    const unsigned sp_end = min(sp_begin + p.spans_per_wave, nspans);
    const unsigned kt_begin = sp_begin * KS;

    f32x16 acc[MB][NP];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int np = 0; np < NP; ++np)
#pragma unroll
            for (int v = 0; v < 16; ++v)
                acc[mb][np][v] = 0.f;

    const unsigned valid_nt = (nt0 < ntiles && !empty_part) ? min((unsigned)(2 * NP), ntiles - nt0) : 0u;
    const unsigned w_row_bytes = ktiles * kTileBytes;
    const unsigned s_row_bytes = p.k / 2;
    const unsigned rows = min(p.m - m0, (unsigned)Cfg::BM);
    const unsigned pt0 = valid_nt ? physical_tile(nt0, ntiles, p.act) : 0u;
    const unsigned span_tiles = !valid_nt ? 0u : p.act ? (valid_nt >> 1) + (ntiles >> 1) : valid_nt;
    // WF = 6 (layout.h, "petit-cdna4-nv6/1"): the image is addressed from its start -- elements [N/32][K/128][3 planes][64 lanes] x 16 B, scales
    // [N/32][K/(128 KS)][64 lanes][2][KS] bytes -- and every LANE finds its own weight row: lane (row = l % 32, h = l / 32) of n32-block np is
    // row row % 16 of the logical n-tile nt0 + 2 np + row / 16, whose physical tile (SiLU-mul: a gate tile for rows 0-15, the matching up tile for
    // 16-31) is one HALF of an image block.  The operand a wave ends up with is what merge_tiles builds for the raw layout, so both epilogues
    // and the quantising SiLU-mul epilogue are shared.
    const __amdgpu_buffer_rsrc_t w_rsrc = WF == 6 ? make_rsrc(p.w, (unsigned)nv6_elem_bytes(p.n, p.k))
                                                  : make_rsrc((const char *)p.w + (size_t)pt0 * w_row_bytes, span_tiles * w_row_bytes);
    const __amdgpu_buffer_rsrc_t s_rsrc = WF == 6 ? make_rsrc(p.s, (unsigned)nv6_scale_bytes(p.n, p.k))
                                                  : make_rsrc((const char *)p.s + (size_t)pt0 * s_row_bytes, span_tiles * s_row_bytes);
    unsigned w_voff[2 * NP], s_voff[2 * NP]; // (WF = 6: one entry per n32-block, [np])
    if constexpr (WF == 6) {
#pragma unroll
        for (int np = 0; np < NP; ++np) {
            const unsigned nt = 2 * np + (m_l >> 4);
            const unsigned pt = physical_tile(nt0 + nt, ntiles, p.act);
            const unsigned lane_img = h * 32 + (pt & 1u) * 16 + (m_l & 15u);
            w_voff[np] = nt < valid_nt ? (pt >> 1) * (ktiles * kNv6TileBytes) + lane_img * 16 : kOob;
            s_voff[np] = nt < valid_nt ? (pt >> 1) * (ktiles * 128u) + lane_img * (2 * KS) : kOob;
        }
    } else {
#pragma unroll
        for (int nt = 0; nt < 2 * NP; ++nt) {
            const unsigned rel = physical_tile(nt0 + nt, ntiles, p.act) - pt0;
            w_voff[nt] = ((unsigned)nt < valid_nt) ? lane * 16 + rel * w_row_bytes : kOob;
            s_voff[nt] = ((unsigned)nt < valid_nt) ? lane * kRecBytes + rel * s_row_bytes : kOob;
        }
    }
    // the 16-byte weight loads of k-tile kt: raw layout -- the two n-tiles of every pair; NV6 image -- planes P1 / P2 / tails of every n32-block
    auto w_load = [&](int i, unsigned kt) -> u32x4 {
        if constexpr (WF == 6)
            return buf_load16(w_rsrc, w_voff[i / 3], kt * kNv6TileBytes + (i % 3) * 1024u, kAuxDefault);
        else
            return buf_load16(w_rsrc, w_voff[i], kt * kTileBytes, kAuxDefault);
    };
    // quantised activations and their scales, k-tile major (see the layout note above): tile kt of this m-block starts at
    // (kt M + m0) rows of 16 ACT (data) / 4 (scales) bytes; the descriptors end with the data / scale region
    constexpr unsigned kRowB = 16 * Cfg::kRowU4;
    const size_t qa_bytes = (size_t)p.m * (p.k / 8 * ACT), qs_bytes = (size_t)p.m * (p.k / 32);
    const size_t lo_bytes = (size_t)p.m * (p.k / 8 * Cfg::kRowU4); // (FP6: the first image; else all of the data)
    const __amdgpu_buffer_rsrc_t qa_rsrc = make_rsrc(ws + (size_t)m0 * kRowB, (unsigned)(lo_bytes - (size_t)m0 * kRowB));
    const __amdgpu_buffer_rsrc_t qh_rsrc = make_rsrc(ws + lo_bytes + (size_t)m0 * 32, ACT == 6 ? (unsigned)(qa_bytes - lo_bytes - (size_t)m0 * 32) : 0u);
    const __amdgpu_buffer_rsrc_t qs_rsrc = make_rsrc(ws + qa_bytes + (size_t)m0 * 4, (unsigned)(qs_bytes - (size_t)m0 * 4));
    const unsigned qa_tile = p.m * kRowB, qs_tile = p.m * 4; // bytes from one k-tile to the next

    // direct-to-LDS staging.  Data: wave-load i of this wave covers rows RPL*(i*WAVES + wave) .. +RPL-1 (RPL = 8 / 16 rows of
    // 128 / 64 bytes); lane l -> row + l / U, position l % U, which receives unit (l % U) ^ swz(row), U = 8 / 4 units per row,
    // swz(row) = (row / 2) % 8 resp. (row / 4) % 4: the 16 lanes of a ds_read_b128 group then hit 16 different 16-byte slots.
    constexpr unsigned U = Cfg::kRowU4, RPL = Cfg::kRowsPerLoad;
    constexpr int kDmaWaves = Cfg::kDmaWaves;
    const bool dma_wave = kDmaWaves == kWaves || wave < (unsigned)kDmaWaves;
    const unsigned dma_row0 = wave * RPL + lane / U;
    auto swz = [](unsigned row) -> unsigned { return ACT == 8 ? (row >> 1) & 7u : (row >> 2) & 3u; };
    // FP6 tail image: rows of two units; lane l of a wave-load -> row l / 2, position l % 2, which receives unit (l % 2) ^ (row / 8) % 2 (the 16 lanes of a
    // ds_read_b128 group -- 16 rows, 512 bytes -- then cover the 256-byte bank window twice without meeting)
    const unsigned hi_row0 = wave * 32 + (lane >> 1);
    const unsigned hi_voff = (ACT == 6 && dma_wave) ? hi_row0 * 32 + (((lane & 1u) ^ ((hi_row0 >> 3) & 1u)) * 16) : kOob;
    const unsigned dma_voff = dma_wave ? dma_row0 * kRowB + (((lane % U) ^ swz(dma_row0)) * 16) : kOob;
    const unsigned qs_voff = (wave * 64 < (unsigned)Cfg::BM) ? (wave * 64 + lane) * 4 : kOob;
    auto dma_stage = [&](unsigned kt, unsigned buf) { // k-tiles kt .. kt + KT - 1 -> stage `buf`
#if defined(__HIP_DEVICE_COMPILE__) && !(PETIT_ABLATE_N32 & 1)
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            u32x4 *const data = smem_g + buf * Cfg::kStageU4 + t * Cfg::kDataU4;
            // (a wave that does not stage writes its zeros -- out-of-range loads -- into the stage's dump slot)
            u32x4 *const dump = smem_g + buf * Cfg::kStageU4 + KT * (Cfg::kDataU4 + Cfg::kScaleU4);
#pragma unroll
            for (int i = 0; i < Cfg::kDataLoads; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(qa_rsrc, (__attribute__((address_space(3))) void *)(dma_wave ? data + (i * kDmaWaves + wave) * 64 : dump), 16,
                                                         dma_voff, i * (RPL * kDmaWaves) * kRowB + (kt + t) * qa_tile, 0, 0);
            if constexpr (ACT == 6) {
#pragma unroll
                for (int i = 0; i < Cfg::kHiLoads; ++i)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(qh_rsrc, (__attribute__((address_space(3))) void *)(dma_wave ? data + Cfg::kLoU4 + (i * kDmaWaves + wave) * 64 : dump), 16,
                                                             hi_voff, i * (32 * kDmaWaves) * 32 + (kt + t) * (p.m * 32), 0, 0);
            }
            u32x4 *const sc = smem_g + buf * Cfg::kStageU4 + KT * Cfg::kDataU4 + t * Cfg::kScaleU4;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(qs_rsrc, (__attribute__((address_space(3))) void *)(sc + wave * 16), 4, qs_voff, (kt + t) * qs_tile, 0, 0);
        }
#else
        (void)kt, (void)buf;
#endif
    };
    // fragment of (m32-block mb, operand q = P1 / P2): row 32 mb + m_l; FP8: units 4q + 2h, 4q + 2h + 1; FP4: unit 2q + h
    struct Frags {
        u32x4 d[MB][kFragU4];
        int s[MB];
    };
    struct Frag1 { // kLean: one m-block's fragment
        u32x4 d[kFragU4];
        int s;
    };
    const unsigned my_swz = swz(m_l); // (32 mb is a multiple of every swizzle period)
    const unsigned hi_swz = (m_l >> 3) & 1u;
    auto read_frag1 = [&](const u32x4 *a_cur, const unsigned char *sc_bytes, int q, int mb, Frag1 &f) {
        const unsigned row = (wm * MB + mb) * 32 + m_l;
        if constexpr (ACT == 6) {
            f.d[0] = a_cur[row * U + ((2 * q + h) ^ my_swz)];
            if (q == 0) // (the tail unit serves both operands of the k-tile: P2's half waits in tail_keep)
                f.d[1] = a_cur[Cfg::kLoU4 + row * 2 + (h ^ hi_swz)];
        } else {
#pragma unroll
            for (int e = 0; e < kFragU4; ++e) {
                const unsigned unit = ACT == 8 ? 4 * q + 2 * h + e : 2 * q + h;
                f.d[e] = a_cur[row * U + (unit ^ my_swz)];
            }
        }
        f.s = (int)sc_bytes[row * 4 + 2 * q + h];
    };
    auto read_frags = [&](const u32x4 *a_cur, const unsigned char *sc_bytes, int q, Frags &f) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const unsigned row = (wm * MB + mb) * 32 + m_l;
            if constexpr (ACT == 6) {
                f.d[mb][0] = a_cur[row * U + ((2 * q + h) ^ my_swz)];
                if (q == 0) // (the tail unit serves both operands of the k-tile: read with P1's, handed on to P2's below)
                    f.d[mb][1] = a_cur[Cfg::kLoU4 + row * 2 + (h ^ hi_swz)];
            } else {
#pragma unroll
                for (int e = 0; e < kFragU4; ++e) {
                    const unsigned unit = ACT == 8 ? 4 * q + 2 * h + e : 2 * q + h;
                    f.d[mb][e] = a_cur[row * U + (unit ^ my_swz)];
                }
            }
            f.s[mb] = (int)sc_bytes[row * 4 + 2 * q + h]; // the block this lane's k belongs to: 2 q + h
        }
    };

    // --- prologue: stage 0 (and 1 when two are kept ahead), scale records, the W ring
    const unsigned kt_end = sp_end * KS;
    if constexpr (Cfg::LW) {
        if (wave_all == (unsigned)(kWaves * KG)) {
            // The loader wave: stages every activation tile of the workgroup (all BM rows: BM / RPL wave-loads of 1 KiB + the scale dwords
            // per k-tile), PF stages ahead, and executes exactly the barriers of the compute waves' main loop; then it is done.
#if defined(__HIP_DEVICE_COMPILE__)
            constexpr int kLdData = Cfg::BM / (int)RPL, kLdScale = (Cfg::BM + 63) / 64, kLdStage = KT * (kLdData + kLdScale);
            unsigned lvoff[kLdData];
#pragma unroll
            for (int i = 0; i < kLdData; ++i) {
                const unsigned row = i * RPL + lane / U;
                lvoff[i] = row * kRowB + (((lane % U) ^ swz(row)) * 16);
            }
            auto load_stage = [&](unsigned kt, unsigned buf) {
                if constexpr (PETIT_ABLATE_N32 & 1)
                    return;
#pragma unroll
                for (int t = 0; t < KT; ++t) {
                    u32x4 *const data = smem + buf * Cfg::kStageU4 + t * Cfg::kDataU4;
#pragma unroll
                    for (int i = 0; i < kLdData; ++i)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(qa_rsrc, (__attribute__((address_space(3))) void *)(data + i * 64), 16, lvoff[i],
                                                                 (kt + t) * qa_tile, 0, 0);
                    u32x4 *const sc = smem + buf * Cfg::kStageU4 + KT * Cfg::kDataU4 + t * Cfg::kScaleU4;
#pragma unroll
                    for (int j = 0; j < kLdScale; ++j)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(qs_rsrc, (__attribute__((address_space(3))) void *)(sc + j * 16), 4,
                                                                 (j * 64 + lane < (unsigned)Cfg::BM) ? (j * 64 + lane) * 4 : kOob, (kt + t) * qs_tile, 0, 0);
                }
            };
#pragma unroll
            for (int i = 0; i < PF; ++i)
                load_stage(kt_begin + i * KT, i); // (past the K slice: out of the descriptor's range -- zeros into a stage nobody reads)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier(); // the compute waves' prologue barrier: stage 0 is complete and visible
            const unsigned nstages = (sp_end - sp_begin) * (KS / KT);
            unsigned buf = PF % NBUF;
            for (unsigned st = 0; st < nstages; ++st) {
                load_stage(kt_begin + (st + PF) * KT, buf);
                buf = buf == (unsigned)(NBUF - 1) ? 0u : buf + 1;
                if (st + 1 < nstages) {
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PF - 1) * kLdStage) : "memory"); // stage st + 1 has landed
                    __builtin_amdgcn_s_barrier();
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // nothing of this wave lands in LDS after it is gone (the epilogue reuses the stages)
#endif
            return;
        }
        if constexpr (Cfg::PW) {
            if (wave_all == (unsigned)(kWaves * KG + 1)) {
#if defined(__HIP_DEVICE_COMPILE__)
                constexpr int kPwTiles = 2 * NP * WAVES, kPwLoads = (kPwTiles * 8 + 63) / 64, kAhead = PETIT_N32_PW;
                const unsigned nt0_wg = bn * WAVES * (2 * NP);
                const unsigned valid_wg = nt0_wg < ntiles ? min((unsigned)kPwTiles, ntiles - nt0_wg) : 0u;
                const unsigned pt0_wg = valid_wg ? physical_tile(nt0_wg, ntiles, p.act) : 0u;
                const unsigned span_wg = !valid_wg ? 0u : p.act ? (valid_wg >> 1) + (ntiles >> 1) : valid_wg;
                const __amdgpu_buffer_rsrc_t pw_rsrc = make_rsrc((const char *)p.w + (size_t)pt0_wg * w_row_bytes, span_wg * w_row_bytes);
                u32x4 *const pw_dump = smem + (Cfg::kSmemU4 - 16);
                unsigned pvoff[kPwLoads];
#pragma unroll
                for (int j = 0; j < kPwLoads; ++j) {
                    const unsigned idx = j * 64 + lane, t = idx / 8, line = idx % 8;
                    const bool mine = PETIT_N32_PWSHARE == 1 || (idx % PETIT_N32_PWSHARE) == (bm % PETIT_N32_PWSHARE);
                    pvoff[j] = (t < valid_wg && mine) ? (physical_tile(nt0_wg + t, ntiles, p.act) - pt0_wg) * w_row_bytes + line * 128 : kOob;
                }
                auto prefetch = [&](unsigned kt) {
                    if (kt < kt_end) {
#pragma unroll
                        for (int j = 0; j < kPwLoads; ++j)
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(pw_rsrc, (__attribute__((address_space(3))) void *)pw_dump, 4, pvoff[j], kt * kTileBytes, 0, 0);
                    }
                };
                for (unsigned kt = kt_begin + D; kt < kt_begin + kAhead; ++kt)
                    prefetch(kt);
                __builtin_amdgcn_s_barrier();
                const unsigned nstages = (sp_end - sp_begin) * (KS / KT);
                for (unsigned st = 0; st < nstages; ++st) {
#pragma unroll
                    for (int t = 0; t < KT; ++t)
                        prefetch(kt_begin + st * KT + t + kAhead);
                    if (st + 1 < nstages)
                        __builtin_amdgcn_s_barrier();
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
                return;
            }
        }
    } else {
        dma_stage(kt_begin, 0);
#pragma unroll
        for (int i = 1; i < PF; ++i)
            if (kt_begin + i * KT < kt_end)
                dma_stage(kt_begin + i * KT, i);
    }
    ScaleRec<kFmtMx, KS> rec[NP][2], rec_next[NP][2];
    auto load_recs = [&](ScaleRec<kFmtMx, KS> (*dst)[2], unsigned sp) {
        if constexpr (WF == 6) { // the lane's record: [P1: KS bytes][P2: KS bytes]
#pragma unroll
            for (int np = 0; np < NP; ++np) {
                const ScaleRec<kFmtNv, KS> x = load_scale_rec<kFmtNv, KS>(s_rsrc, s_voff[np], sp * 64 * (2 * KS));
                if constexpr (KS == 2) {
                    dst[np][0].d[0] = x.d[0] & 0xffffu, dst[np][1].d[0] = x.d[0] >> 16;
                } else {
#pragma unroll
                    for (int d = 0; d < kRecDw; ++d)
                        dst[np][0].d[d] = x.d[d], dst[np][1].d[d] = x.d[kRecDw + d];
                }
            }
            return;
        }
#pragma unroll
        for (int np = 0; np < NP; ++np) {
            const ScaleRec<kFmtMx, KS> x = load_scale_rec<kFmtMx, KS>(s_rsrc, s_voff[2 * np], sp * 64 * kRecBytes);
            const ScaleRec<kFmtMx, KS> y = load_scale_rec<kFmtMx, KS>(s_rsrc, s_voff[2 * np + 1], sp * 64 * kRecBytes);
#pragma unroll
            for (int d = 0; d < kRecDw; ++d)
                merge_tiles(x.d[d], y.d[d], dst[np][0].d[d], dst[np][1].d[d]);
        }
    };
    load_recs(rec, sp_begin);
    u32x4 wring[D][kWL * NP];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int nt = 0; nt < kWL * NP; ++nt)
            wring[i][nt] = w_load(nt, kt_begin + i);
    __syncthreads(); // (drains everything once: stage 0 is complete and visible)

    unsigned cur_buf = 0; // stage index of the current k-tiles, rotates with period NBUF (wave-uniform)
    auto span_body = [&](const unsigned sp, auto last_c) {
        constexpr bool kLast = decltype(last_c)::value;
        const unsigned kt0 = sp * KS;
        if constexpr (!kLast)
            load_recs(rec_next, sp + 1);
        static_for<0, KS / KT>([&](auto s_c) {
            constexpr int S = decltype(s_c)::value, T0 = S * KT; // stage S of the span: k-tiles T0 .. T0 + KT - 1
            constexpr bool kNextStage = !kLast || (T0 + KT < KS);
            const unsigned kt = kt0 + T0;
            const u32x4 *const stage = smem_g + cur_buf * Cfg::kStageU4;
            const unsigned char *const stage_sc = reinterpret_cast<const unsigned char *>(stage + KT * Cfg::kDataU4);
            // request the stage PF ahead: everybody left that LDS stage at the barrier that ended the previous stage
            if constexpr (PF == 1) {
                if constexpr (kNextStage)
                    dma_stage(kt + KT, cur_buf ^ 1u);
            } else {
                // always issued (beyond the K slice the loads are out of the descriptor's range: zeros into a stage nobody
                // reads), so the wait below is one constant and there is no branch around the MFMA stream
                if constexpr (!Cfg::LW)
                    dma_stage(kt + PF * KT, cur_buf == 0 ? (unsigned)(NBUF - 1) : cur_buf - 1);
            }
            Frags fr[Cfg::kLean ? 1 : 2];
            Frag1 f1[2];
            unsigned tail_keep[MB][2]; // kLean, FP6: registers 4-5 of the P2 operands, read with P1's
            if constexpr (Cfg::kLean)
                read_frag1(stage, stage_sc, 0, 0, f1[0]);
            else
                read_frags(stage, stage_sc, 0, fr[0]);
            __builtin_amdgcn_sched_barrier(0);
            int refills = 0; // W loads issued in this stage (compile-time after unrolling)
            static_for<0, KT>([&](auto t_c) {
                constexpr int TI = decltype(t_c)::value, T = T0 + TI, SLOT = T % D;
                constexpr bool kRefill = !kLast || (T + D < KS);
                const u32x4 *const a_cur = stage + TI * Cfg::kDataU4;
                const unsigned char *const sc_cur = stage_sc + TI * Cfg::kScaleU4 * 16;
                // the pair's packed words, merged: the lane's 4 registers ARE the FP4 operand
                i32x8 wop[NP][2];
#pragma unroll
                for (int np = 0; np < NP; ++np) {
                    if constexpr (WF == 6) { // registers 0-3 of P1 / P2 as loaded; 4-5 from the shared tail plane
                        const u32x4 l1 = wring[SLOT][3 * np], l2 = wring[SLOT][3 * np + 1], tl = wring[SLOT][3 * np + 2];
                        wop[np][0] = i32x8{(int)l1[0], (int)l1[1], (int)l1[2], (int)l1[3], (int)tl[0], (int)tl[1], 0, 0};
                        wop[np][1] = i32x8{(int)l2[0], (int)l2[1], (int)l2[2], (int)l2[3], (int)tl[2], (int)tl[3], 0, 0};
                    } else {
                        unsigned pw[2][4];
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            merge_tiles(wring[SLOT][2 * np][j], wring[SLOT][2 * np + 1][j], pw[0][j], pw[1][j]);
#pragma unroll
                        for (int q = 0; q < 2; ++q)
                            wop[np][q] = i32x8{(int)pw[q][0], (int)pw[q][1], (int)pw[q][2], (int)pw[q][3], 0, 0, 0, 0};
                    }
                }
                // raw layout: the merged operands are copies: the ring slot is free as soon as they exist, and its refill goes out BEFORE the tile's MFMAs
                // (requests spread over the stage instead of a burst behind the barrier: -2...6 % at M = 512, profiles/r03_native_ablation.md).
                // NV6 image: the ring registers ARE the operands (a refill ahead of the MFMAs would need 12 more registers per n32-block: the
                // 128 x 256 forms then spill 100-450 registers at two waves per SIMD), so the slot is refilled right after its last MFMA is issued.
                auto refill_slot = [&]() {
                    if constexpr (kRefill && !(PETIT_ABLATE_N32 & 2)) {
#pragma unroll
                        for (int nt = 0; nt < kWL * NP; ++nt)
                            wring[SLOT][nt] = w_load(nt, kt0 + T + D);
                    }
                };
                if constexpr (WF != 6)
                    refill_slot();
                static_for<0, 2>([&](auto q_c) {
                    constexpr int q = decltype(q_c)::value, gi = 2 * TI + q; // group index inside the stage
                    if constexpr (Cfg::kLean) {
                        // one m-block ahead: fragment li + 1 is requested before the MFMAs of fragment li issue
                        static_for<0, MB>([&](auto mb_c) {
                            constexpr int mb = decltype(mb_c)::value, li = gi * MB + mb;
                            if constexpr (li + 1 < 2 * KT * MB) {
                                constexpr int ngi = (li + 1) / MB, nmb = (li + 1) % MB, nti = ngi / 2, nq = ngi % 2;
                                read_frag1(stage + nti * Cfg::kDataU4, stage_sc + nti * Cfg::kScaleU4 * 16, nq, nmb, f1[(li + 1) & 1]);
                            }
                            const u32x4 lo = f1[li & 1].d[0], hi = f1[li & 1].d[kFragU4 - 1];
                            i32x8 aop;
                            if constexpr (ACT == 6) {
                                if constexpr (q == 0) {
                                    tail_keep[mb][0] = hi[2], tail_keep[mb][1] = hi[3];
                                    aop = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], 0, 0};
                                } else {
                                    aop = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)tail_keep[mb][0], (int)tail_keep[mb][1], 0, 0};
                                }
                            } else {
                                aop = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
                            }
#pragma unroll
                            for (int np = 0; np < NP; ++np)
                                acc[mb][np] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wop[np][q], aop, acc[mb][np], kCbsz /* A = FP4 / FP6 */, kBlgp /* B = FP8 e4m3 / FP6 e2m3 */,
                                                                                              T % 4, (int)rec[np][q].d[T / 4], 0, f1[li & 1].s);
                        });
                    }
                    // fragments of the next group (next operand, or the next tile of the stage) while this group's MFMAs run
                    if constexpr (!Cfg::kLean && gi + 1 < 2 * KT && !(PETIT_ABLATE_N32 & 4)) {
                        constexpr int nti = (gi + 1) / 2, nq = (gi + 1) % 2;
                        read_frags(stage + nti * Cfg::kDataU4, stage_sc + nti * Cfg::kScaleU4 * 16, nq, fr[(gi + 1) & 1]);
                        if constexpr (ACT == 6 && nq == 1) { // P2's fragments: the tail unit P1's read brought (fr[0] is always a P1 set, fr[1] a P2 set)
#pragma unroll
                            for (int mb = 0; mb < MB; ++mb)
                                fr[1].d[mb][1] = fr[0].d[mb][1];
                        }
                    }
#pragma unroll
                    for (int mb = 0; mb < (Cfg::kLean ? 0 : MB); ++mb) {
                        i32x8 aop;
                        constexpr int fi = (PETIT_ABLATE_N32 & 4) ? 0 : (Cfg::kLean ? 0 : (gi & 1));
                        if constexpr (ACT == 8) {
                            const u32x4 lo = fr[fi].d[mb][0], hi = fr[fi].d[mb][kFragU4 - 1];
                            aop = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
                        } else if constexpr (ACT == 6) {
                            const u32x4 lo = fr[fi].d[mb][0], tail = fr[fi].d[mb][1];
                            aop = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)tail[2 * q], (int)tail[2 * q + 1], 0, 0};
                        } else {
                            const u32x4 lo = fr[fi].d[mb][0];
                            aop = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], 0, 0, 0, 0};
                        }
#pragma unroll
                        for (int np = 0; np < NP; ++np) {
                            if constexpr (PETIT_ABLATE_N32 & 8) {
#if defined(__HIP_DEVICE_COMPILE__)
                                asm volatile("" ::"v"(wop[np][q]), "v"(aop), "v"(fr[fi].s[mb]), "v"(rec[np][q].d[T / 4]));
#endif
                            } else {
                                acc[mb][np] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(
                                    wop[np][q], aop, acc[mb][np], kCbsz /* A = FP4 / FP6 */, kBlgp /* B = FP8 e4m3 / FP6 e2m3 / FP4 */, T % 4,
                                    (int)rec[np][q].d[T / 4], 0, fr[fi].s[mb]);
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
                if constexpr (WF == 6) {
                    refill_slot();
                    __builtin_amdgcn_sched_barrier(0);
                }
                (void)a_cur, (void)sc_cur, (void)refills;
            });
            if constexpr (PF == 1) {
                if constexpr (kNextStage)
                    __syncthreads();
                cur_buf ^= 1u;
            } else {
                // the next stage (requested PF - 1 stages ago) must have landed.  Loads retire in issue order, and after that
                // request came the W refills of the PF - 1 stages before this one and the PF - 1 requests since: all of them stay
                // in flight (a wait for "at most kStageLoads outstanding" would also drain the refills one stage after they were
                // issued, whatever the ring depth D -- the weight stream then pays an L2 round trip per stage).  The count is
                // exact per stage: a stage before the first of a span belongs to a non-last span (all refills issued) or to the
                // prologue (which drained everything: a larger count waits for nothing, and nothing is pending).
                if constexpr (kNextStage) {
                    constexpr int kPrevRefills = [] {
                        int n = 0;
                        for (int j = 1; j < PF; ++j) // (the refills of the PF - 1 stages before this one)
                            n += n32_stage_refills(kLast, T0 - j * KT, KT, D, KS, NP, Cfg::kWLoads);
                        n += n32_stage_refills(kLast, T0, KT, D, KS, NP, Cfg::kWLoads); // (this stage's refills are already in the queue)
                        return (PETIT_ABLATE_N32 & 2) ? 0 : n;
                    }();
#if defined(__HIP_DEVICE_COMPILE__)
                    if constexpr (!Cfg::LW) // (with a loader wave the compute waves wait for nothing here: the loader vouches for the stage)
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PF - 1) * Cfg::kStageLoads + kPrevRefills) : "memory");
                    __builtin_amdgcn_s_barrier();
#endif
                }
                __builtin_amdgcn_sched_barrier(0);
                cur_buf = cur_buf == (unsigned)(NBUF - 1) ? 0u : cur_buf + 1;
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (!kLast) {
#pragma unroll
            for (int np = 0; np < NP; ++np)
                rec[np][0] = rec_next[np][0], rec[np][1] = rec_next[np][1];
        }
    };
    for (unsigned sp = sp_begin; sp + 1 < sp_end; ++sp)
        span_body(sp, std::false_type{});
    span_body(sp_end - 1, std::true_type{});

    // every MFMA executes with all 64 lanes, before the lane-divergent stores (see pin_acc in gemm_native.hpp)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int np = 0; np < NP; ++np) {
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" : "+v"(acc[mb][np]));
#endif
        }

    if constexpr (KG == 2) {
        // a group with fewer spans than its partner (the last part of an odd split) keeps the partner's barrier count: a span is
        // KS / KT barriers (the prologue's stands in for the one the last stage does not execute)
        for (unsigned i = sp_end - sp_begin; i < p.spans_per_wave; ++i)
            for (int b = 0; b < KS / KT; ++b)
                __builtin_amdgcn_s_barrier();
        // sum the two groups: group 1 parks its accumulators in LDS (one f32 per lane and register: conflict-free), group 0 adds
        __syncthreads(); // every stage is dead
        float *const red = reinterpret_cast<float *>(smem);
        if (kg == 1) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int np = 0; np < NP; ++np)
#pragma unroll
                    for (int v = 0; v < 16; ++v)
                        red[(((wn * MB + mb) * NP + np) * 16 + v) * 64 + lane] = acc[mb][np][v];
        }
        __syncthreads();
        if (kg == 0) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int np = 0; np < NP; ++np)
#pragma unroll
                    for (int v = 0; v < 16; ++v)
                        acc[mb][np][v] += red[(((wn * MB + mb) * NP + np) * 16 + v) * 64 + lane];
        }
    }

    if constexpr (PETIT_ABLATE_N32 & 16)
        return;
    // --- epilogue: the 32x32 accumulator layout of gemm_wide.hpp
    const unsigned m_base = m0 + wm * (32 * MB) + m_l;
    if (gridDim.z > 1) {
        if (KG == 2 && kg != 0)
            return;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int np = 0; np < NP; ++np)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const unsigned m = m_base + mb * 32;
                    const unsigned nt = 2 * np + (u >> 1);
                    const unsigned n = (nt0 + nt) * 16 + (u & 1) * 8 + 4 * h;
                    if (m < p.m && nt < valid_nt)
                        *reinterpret_cast<f32x4 *>(p.workspace + ((size_t)blockIdx.z * p.m + m) * p.n + n) =
                            f32x4{acc[mb][np][4 * u], acc[mb][np][4 * u + 1], acc[mb][np][4 * u + 2], acc[mb][np][4 * u + 3]};
                }
        return;
    }
    const float gs = *p.gs;
    if (p.act) {
        const unsigned n_half = p.n >> 1;
        if (KG == 2 && kg != 0)
            return;
        if constexpr (NP == 2 && WM == 1 && WAVES == 4 && KG == 1 && Cfg::LW == 0) {
            if (p.out_format) { // (workgroup-uniform) the launcher admits it for full 256-column tiles only: n % 512 == 0
                __syncthreads(); // the stages are dead: their LDS holds the scale bytes of the tile
                unsigned char *const lds_sc = reinterpret_cast<unsigned char *>(smem);
                const unsigned col0 = (nt0 >> 1) * 16;
                if (p.out_format == 4)
                    n32_silu_quant_epilogue<AT, 4, MB>(acc, gs, p.bias, (unsigned char *)p.c, p.m, n_half, m_base, col0, bn, wn, m_l, h, lds_sc, m0, tid);
                else if (p.out_format == 6)
                    n32_silu_quant_epilogue<AT, 6, MB>(acc, gs, p.bias, (unsigned char *)p.c, p.m, n_half, m_base, col0, bn, wn, m_l, h, lds_sc, m0, tid);
                else
                    n32_silu_quant_epilogue<AT, 8, MB>(acc, gs, p.bias, (unsigned char *)p.c, p.m, n_half, m_base, col0, bn, wn, m_l, h, lds_sc, m0, tid);
                return;
            }
        }
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int np = 0; np < NP; ++np)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const unsigned m = m_base + mb * 32;
                    const unsigned n = ((nt0 + 2 * np) >> 1) * 16 + u * 8 + 4 * h;
                    const f32x16 &a = acc[mb][np];
                    if (m < p.m && (unsigned)(2 * np + 1) < valid_nt)
                        *reinterpret_cast<uint2 *>((char *)p.c + ((size_t)m * n_half + n) * 2) = finish4_silu_mul<AT>(
                            f32x4{a[4 * u], a[4 * u + 1], a[4 * u + 2], a[4 * u + 3]},
                            f32x4{a[8 + 4 * u], a[8 + 4 * u + 1], a[8 + 4 * u + 2], a[8 + 4 * u + 3]}, gs, p.bias, n, n_half);
                }
        return;
    }
    // plain / bias epilogue: through an LDS image of the C tile, whole rows out (c_tile_store, device_common.hpp).  The stages
    // are dead: every wave is past its last fragment read and (barrier) its last DMA has landed.
    __syncthreads();
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int np = 0; np < NP; ++np)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned nt = 2 * np + (u >> 1);
                const unsigned n = (nt0 + nt) * 16 + (u & 1) * 8 + 4 * h;
                const f32x4 v = f32x4{acc[mb][np][4 * u], acc[mb][np][4 * u + 1], acc[mb][np][4 * u + 2], acc[mb][np][4 * u + 3]};
                if (nt < valid_nt && (KG == 1 || kg == 0)) // (bias is read at n: only for columns that exist)
                    c_tile_put<Cfg::BN>(smem, (wm * MB + mb) * 32 + m_l, (wn * 2 * NP + nt) * 16 + (u & 1) * 8 + 4 * h,
                                        finish4<AT>(v, gs, p.bias, n));
            }
    __syncthreads();
    const unsigned n0 = bn * Cfg::BN;
    c_tile_store<Cfg::BM, Cfg::BN, Cfg::kComputeThreads>(smem, p.c, p.n, m0, n0, rows, n0 < p.n ? min((unsigned)Cfg::BN, p.n - n0) : 0u, tid);
}

} // namespace petit_amd
