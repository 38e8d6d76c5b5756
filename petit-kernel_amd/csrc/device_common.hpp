// device_common.hpp -- what the three GEMM kernels (gemm_stream / gemm_tiled / gemm_native) share on the device:
// vector types, the activation-type tags, raw buffer loads, the E2M1 unpack (hardware converts), the MFMA
// wrapper, the scale records of the packed layout, and the epilogue (global scale, optional bias / SiLU-mul, one
// rounding).  Reference counterparts: quantization/dequant.cuh (unpack), rocm/amd_intrinsics.cuh:92-130 (buffer
// resources), quantization/qgemm.cuh:95-192 (result write-out).
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

#include "layout.h"
#include "petit_internal.h"

namespace petit_amd {

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(2))) unsigned short u16x2;
typedef __attribute__((ext_vector_type(6))) unsigned u32x6;
typedef __attribute__((ext_vector_type(16))) unsigned u32x16;
typedef __attribute__((ext_vector_type(16))) float f32x16v;

enum : int { kFmtNv = 0, kFmtMx = 1 };

// gfx940-family cache-policy bits of the buffer intrinsics' aux operand.
enum : int { kAuxDefault = 0, kAuxNt = 2 };

struct Bf16 {
    using frag = bf16x8;
    static constexpr int kType = kDataTypeBf16;
    static constexpr bool kBfp = false, kAdaptive = false;
};
struct Fp16 {
    using frag = f16x8;
    static constexpr int kType = kDataTypeFp16;
    static constexpr bool kBfp = false, kAdaptive = false;
};
// bf16 activations on the fp16 pipeline (staged path only).  The fp16 unpack of an
// NVFP4 word is 8 VALU (convert + v_pk_mul_f16) against 12 for bf16 (convert to f32,
// v_pk_mul_f32, v_perm_b32), and the unpack is what the small-M kernel is exposed on.
// bf16 has 8 significant bits, fp16 has 11, so a bf16 value is exactly an fp16 value
// whenever its exponent fits.  Each wave therefore rescales ITS span of every
// activation row by a power of two (block floating point: 2^-sh with sh chosen from
// the span's largest exponent so the maximum lands in [2^14, 2^15)), converts to
// fp16, accumulates that span in f32 and multiplies the span's partial sum back by
// 2^sh before it joins the running total.  Exact except for elements more than 2^29
// below their span's maximum (fp16 subnormal range), whose contribution to the sum is
// below 2^-29 of the largest term -- far under the single bf16 rounding of the result.
struct Bf16Bfp {
    using frag = f16x8;
    static constexpr int kType = kDataTypeBf16;
    static constexpr bool kBfp = true, kAdaptive = false;
};
// fp16 activations against MXFP4 weights (a capability the reference does not have:
// fp4/warp_schedule_fp16.cuh:22-26 static_asserts it away).  e8m0 block scales span 2^-127..2^127, far outside fp16 -- but
// no real checkpoint's do: when a block scale byte lies in 114..140 (2^-13..2^13), e2m1 x scale is a NORMAL fp16 number
// (0.5 x 2^-13 = 2^-14 ... 6 x 2^13 = 49152) and v_cvt_scalef32_pk_f16_fp4 produces it exactly with the scale in the
// convert: 4 VALU per word, one f16 MFMA per fragment.  The kernels DECIDE THIS THEMSELVES, per wave and span, on the scale
// record they hold anyway (mx_rec_outside_f16: 3 VALU per dword + one ballot per span), so nobody has to promise anything:
// the first span with a byte outside 114..140 switches the wave, for the rest of its K range, to the fallback body, which
// is exact for ANY e8m0 scale: weights dequantised to bf16 (exact: bf16 has f32's exponent range) and every fp16 activation
// fragment split in registers into two bf16 fragments, a = hi + lo exactly (hi = the top 8 significant bits, lo = the
// remaining <= 3): two bf16 MFMAs per word, f32 accumulation, no precision lost anywhere.  One-way switch = two loops in
// sequence, never a diamond inside the unrolled span.
struct Fp16Mx {
    using frag = f16x8;
    static constexpr int kType = kDataTypeFp16;
    static constexpr bool kBfp = false, kAdaptive = true;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, unsigned bytes) {
    // raw buffer, no swizzle, bounds-checked: 0x00020000 = DATA_FORMAT_32
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}

__device__ __forceinline__ u32x4 buf_load16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff,
                                            int aux) {
    // aux must be a literal for the builtin
    if (aux == kAuxNt)
        return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, kAuxNt));
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, kAuxDefault));
}

// --- unpack: one 32-bit word = 8 consecutive-k E2M1 values -> one MFMA operand --

template <int SEL> __device__ __forceinline__ f32x2 cvt_fp4_f32(unsigned w, float scale) {
    return __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(w, scale, SEL);
}
template <int SEL> __device__ __forceinline__ bf16x2 cvt_fp4_bf16(unsigned w, float scale) {
    return __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, scale, SEL);
}
template <int SEL> __device__ __forceinline__ f16x2 cvt_fp4_f16(unsigned w, float scale) {
    return __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(w, scale, SEL);
}

// NVFP4 -> bf16: exact (fp4 x e4m3 needs <= 5 significant bits), so the f32
// products are turned into bf16 by TRUNCATION -- one v_perm_b32 per pair picking
// the two high halves -- instead of v_cvt_pk_bf16_f32, which measures ~10 cycles
// per wave-instruction on gfx950 against 4 for v_perm_b32 (tools/probes/valu_rate).
// NOTE (hipcc / ROCm 7.2): __builtin_bit_cast applied directly to a vector ELEMENT
// expression (p.y, h[d]) silently reads element 0.  Always copy the element into a
// scalar first -- every bit_cast in this file takes a named scalar or a whole vector.
__device__ __forceinline__ unsigned trunc_pack_bf16(f32x2 p) {
    const float x = p.x, y = p.y;
    return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, y), __builtin_bit_cast(unsigned, x), 0x07060302u);
}
__device__ __forceinline__ bf16x8 unpack_nv(Bf16, unsigned w, float s) {
    const f32x2 p0 = cvt_fp4_f32<0>(w, 1.0f) * s;
    const f32x2 p1 = cvt_fp4_f32<1>(w, 1.0f) * s;
    const f32x2 p2 = cvt_fp4_f32<2>(w, 1.0f) * s;
    const f32x2 p3 = cvt_fp4_f32<3>(w, 1.0f) * s;
    return __builtin_bit_cast(bf16x8, u32x4{trunc_pack_bf16(p0), trunc_pack_bf16(p1), trunc_pack_bf16(p2), trunc_pack_bf16(p3)});
}
// NVFP4 -> fp16: exact (|fp4 * s| <= 2688, >= 2^-10).
__device__ __forceinline__ f16x8 unpack_nv(Fp16, unsigned w, float s) {
    const f16x2 s2 = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(s, s));
    f16x2 q0 = cvt_fp4_f16<0>(w, 1.0f) * s2;
    f16x2 q1 = cvt_fp4_f16<1>(w, 1.0f) * s2;
    f16x2 q2 = cvt_fp4_f16<2>(w, 1.0f) * s2;
    f16x2 q3 = cvt_fp4_f16<3>(w, 1.0f) * s2;
    return f16x8{q0[0], q0[1], q1[0], q1[1], q2[0], q2[1], q3[0], q3[1]};
}
__device__ __forceinline__ f16x8 unpack_nv(Fp16, unsigned w, float s);
__device__ __forceinline__ f16x8 unpack_nv(Bf16Bfp, unsigned w, float s) { return unpack_nv(Fp16{}, w, s); }
// MXFP4 -> bf16: the e8m0 block scale (a power of two) rides in the convert.
__device__ __forceinline__ bf16x8 unpack_mx(Bf16, unsigned w, float s) {
    bf16x2 q0 = cvt_fp4_bf16<0>(w, s);
    bf16x2 q1 = cvt_fp4_bf16<1>(w, s);
    bf16x2 q2 = cvt_fp4_bf16<2>(w, s);
    bf16x2 q3 = cvt_fp4_bf16<3>(w, s);
    return bf16x8{q0[0], q0[1], q1[0], q1[1], q2[0], q2[1], q3[0], q3[1]};
}

// MXFP4 -> fp16 with the block scale in the convert: exact iff every product e2m1 x 2^(s - 127) is a NORMAL fp16 number, i.e. 114 <= s <= 140
// (0.5 x 2^-13 = 2^-14 ... 6 x 2^13 = 49152).  Only the fast body of the Fp16Mx kernels calls it, after mx_rec_outside_f16 has cleared the span.
__device__ __forceinline__ f16x8 unpack_mx(Fp16, unsigned w, float s) {
    f16x2 q0 = cvt_fp4_f16<0>(w, s);
    f16x2 q1 = cvt_fp4_f16<1>(w, s);
    f16x2 q2 = cvt_fp4_f16<2>(w, s);
    f16x2 q3 = cvt_fp4_f16<3>(w, s);
    return f16x8{q0[0], q0[1], q1[0], q1[1], q2[0], q2[1], q3[0], q3[1]};
}
__device__ __forceinline__ f16x8 unpack_mx(Fp16Mx, unsigned w, float s) { return unpack_mx(Fp16{}, w, s); }

// 8 fp16 values -> (hi, lo) bf16 fragments with hi + lo == the fp16 value exactly.
__device__ __forceinline__ void split_f16(const u32x4 &h, u32x4 &hi, u32x4 &lo) {
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const unsigned hd = h[d]; // scalar copy first: see the bit_cast note above
        const f16x2 p = __builtin_bit_cast(f16x2, hd);
        const float f0 = (float)p[0], f1 = (float)p[1];             // exact
        const unsigned u0 = __builtin_bit_cast(unsigned, f0) & 0xffff0000u;
        const unsigned u1 = __builtin_bit_cast(unsigned, f1) & 0xffff0000u;
        const float l0 = f0 - __builtin_bit_cast(float, u0);       // <= 3 significant bits: exact,
        const float l1 = f1 - __builtin_bit_cast(float, u1);       // and exactly a bf16
        hi[d] = (u0 >> 16) | u1;
        lo[d] = (__builtin_bit_cast(unsigned, l0) >> 16) | (__builtin_bit_cast(unsigned, l1) & 0xffff0000u);
    }
}

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(f16x8 a, f16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// hipcc (ROCm 7.2, gfx950) can miss the "MFMA writes VGPRs -> VALU reads them" hazard when the reader sits in a different basic block than the
// last MFMA: at the join after the Fp16Mx fast / fallback branch it issued v_pk_add_f32 two instructions after the v_mfma that writes its
// operand (found on hardware: the .zw half of every accumulator pair lost).  Wherever accumulators cross such a join:
// mfma_join_pin(acc) on every accumulator, mfma_join_settle() once, mfma_join_pin(acc) again -- the pins are empty asm statements that order
// the wait after the MFMAs and before the readers, the settle is 32 wait states (the longest MFMA here takes 16 passes).
template <class V> __device__ __forceinline__ void mfma_join_pin(V &acc) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(acc));
#endif
}
__device__ __forceinline__ void mfma_join_settle() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#endif
}

// Fp16Mx fallback: one weight word (bf16, any scale) against an fp16 activation fragment already split into hi + lo.
__device__ __forceinline__ f32x4 mfma16_hilo(bf16x8 w, const u32x4 &hi, const u32x4 &lo, f32x4 c) {
    c = mfma16(w, __builtin_bit_cast(bf16x8, hi), c);
    return mfma16(w, __builtin_bit_cast(bf16x8, lo), c);
}

// e4m3 byte SEL of a packed dword -> f32 (OCP e4m3 on gfx950).
template <int SEL> __device__ __forceinline__ float e4m3_byte(unsigned packed) {
    return __builtin_amdgcn_cvt_f32_fp8((int)packed, SEL);
}
// e8m0 byte SEL of a packed dword -> f32 2^(e-127)  (dequant.cuh:198-203).
template <int SEL> __device__ __forceinline__ float e8m0_byte(unsigned packed) {
    return __builtin_bit_cast(float, ((packed >> (8 * SEL)) & 0xffu) << 23);
}

__device__ __forceinline__ unsigned pack2(Bf16, float lo, float hi) {
    bf16x2 q = __builtin_convertvector(f32x2{lo, hi}, bf16x2); // RNE, qgemm.cuh:161-176
    return __builtin_bit_cast(unsigned, q);
}
__device__ __forceinline__ unsigned pack2(Bf16Bfp, float lo, float hi) { return pack2(Bf16{}, lo, hi); }
__device__ __forceinline__ unsigned pack2(Fp16, float lo, float hi);
__device__ __forceinline__ unsigned pack2(Fp16Mx, float lo, float hi) { return pack2(Fp16{}, lo, hi); }
__device__ __forceinline__ unsigned pack2(Fp16, float lo, float hi) {
    f16x2 q = __builtin_convertvector(f32x2{lo, hi}, f16x2); // RNE
    return __builtin_bit_cast(unsigned, q);
}

// The epilogue of every kernel: 4 consecutive n of one output row, x global scale (+ bias[n..n+3]
// when the caller fused one, petit_epilogue in include/petit_amd.h), ONE round-to-nearest-even to
// the 16-bit output type (qgemm.cuh:95-192 in the reference, which has no bias).
template <class AT> __device__ __forceinline__ uint2 finish4(const f32x4 v, const float gs, const void *bias, const unsigned n) {
    float b[4] = {0.f, 0.f, 0.f, 0.f};
    if (bias) {
        const uint2 raw = *reinterpret_cast<const uint2 *>((const char *)bias + (size_t)n * 2);
        const unsigned w0 = raw.x, w1 = raw.y; // named scalars: see the bit_cast note in DESIGN.md section 9
        if constexpr (AT::kType == kDataTypeBf16) {
            const unsigned b0 = w0 << 16, b1 = w0 & 0xffff0000u, b2 = w1 << 16, b3 = w1 & 0xffff0000u;
            b[0] = __builtin_bit_cast(float, b0), b[1] = __builtin_bit_cast(float, b1);
            b[2] = __builtin_bit_cast(float, b2), b[3] = __builtin_bit_cast(float, b3);
        } else {
            const f16x2 h0 = __builtin_bit_cast(f16x2, w0), h1 = __builtin_bit_cast(f16x2, w1);
            const _Float16 e0 = h0[0], e1 = h0[1], e2 = h1[0], e3 = h1[1];
            b[0] = (float)e0, b[1] = (float)e1, b[2] = (float)e2, b[3] = (float)e3;
        }
    }
    uint2 o;
    o.x = pack2(AT{}, __builtin_fmaf(v[0], gs, b[0]), __builtin_fmaf(v[1], gs, b[1]));
    o.y = pack2(AT{}, __builtin_fmaf(v[2], gs, b[2]), __builtin_fmaf(v[3], gs, b[3]));
    return o;
}

// Large-M epilogue through LDS.  C^T = W . A^T leaves a lane with 4 consecutive n of one row (8 bytes), so direct stores
// write C in 16- or 32-byte pieces of 16 / 32 different rows per instruction: every 128-byte line of C is touched by 4-8
// requests (measured on the native kernel, gate_up M = 512: 24-49 us of a 200 us launch went away with the stores,
// tools/ablate_native32.sh).  Instead every wave drops its finished 8-byte pieces into a [BM][BN] 16-bit image of the
// workgroup's C tile in LDS (rows padded by 16 B: a ds_write_b64 of 16 rows and a ds_read_b128 along a row are both
// conflict-free), and after one barrier the workgroup stores whole rows: 16 bytes per lane, BN / 8 consecutive lanes per
// row, i.e. full cache lines.
template <int BN> struct CTile {
    static constexpr int kStrideU4 = BN / 8 + 1; // 16-byte units per row, one of them padding
    static constexpr int u4(int bm) { return bm * kStrideU4; }
};
template <int BN> __device__ __forceinline__ void c_tile_put(u32x4 *img, unsigned row, unsigned col, uint2 v) {
    *reinterpret_cast<uint2 *>(reinterpret_cast<char *>(img) + row * (CTile<BN>::kStrideU4 * 16) + col * 2) = v;
}
// rows [0, rows_valid) x columns [0, cols_valid) of the image -> C[m0 + row][n0 + col]; cols_valid is a multiple of 8
template <int BM, int BN, int THREADS>
__device__ __forceinline__ void c_tile_store(const u32x4 *img, void *c, unsigned ldc, unsigned m0, unsigned n0, unsigned rows_valid,
                                             unsigned cols_valid, unsigned tid) {
    constexpr int kPerRow = BN / 8;
#pragma unroll
    for (int u = 0; u < BM * kPerRow; u += THREADS) {
        const unsigned unit = u + tid, row = unit / kPerRow, cu = unit % kPerRow;
        if ((BM * kPerRow) % THREADS != 0 && unit >= (unsigned)(BM * kPerRow))
            break;
        if (row < rows_valid && cu * 8 < cols_valid)
            *reinterpret_cast<u32x4 *>(reinterpret_cast<char *>(c) + ((size_t)(m0 + row) * ldc + n0 + cu * 8) * 2) =
                img[row * CTile<BN>::kStrideU4 + cu];
    }
}

// SiLU-mul epilogue (petit_epilogue.activation = 1): `gate` holds 4 consecutive columns j of the first half of
// the GEMM's N, `up` the same columns of the second half; out[j] = silu(y_gate[j]) * y_up[j], y = acc*gs + bias,
// rounded once.  bias (if any) spans the full N: gate part at n, up part at n + n_half.
template <class AT>
__device__ __forceinline__ f32x4 silu_mul4(const f32x4 gate, const f32x4 up, const float gs, const void *bias, const unsigned n,
                                           const unsigned n_half) {
    float bg[4] = {0.f, 0.f, 0.f, 0.f}, bu[4] = {0.f, 0.f, 0.f, 0.f};
    if (bias) {
        auto load4 = [&](unsigned col, float *b) {
            const uint2 raw = *reinterpret_cast<const uint2 *>((const char *)bias + (size_t)col * 2);
            const unsigned w0 = raw.x, w1 = raw.y;
            if constexpr (AT::kType == kDataTypeBf16) {
                const unsigned b0 = w0 << 16, b1 = w0 & 0xffff0000u, b2 = w1 << 16, b3 = w1 & 0xffff0000u;
                b[0] = __builtin_bit_cast(float, b0), b[1] = __builtin_bit_cast(float, b1);
                b[2] = __builtin_bit_cast(float, b2), b[3] = __builtin_bit_cast(float, b3);
            } else {
                const f16x2 h0 = __builtin_bit_cast(f16x2, w0), h1 = __builtin_bit_cast(f16x2, w1);
                const _Float16 e0 = h0[0], e1 = h0[1], e2 = h1[0], e3 = h1[1];
                b[0] = (float)e0, b[1] = (float)e1, b[2] = (float)e2, b[3] = (float)e3;
            }
        };
        load4(n, bg);
        load4(n + n_half, bu);
    }
    f32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float g = __builtin_fmaf(gate[i], gs, bg[i]), u = __builtin_fmaf(up[i], gs, bu[i]);
        o[i] = g / (1.0f + __expf(-g)) * u;
    }
    return o;
}
template <class AT>
__device__ __forceinline__ uint2 finish4_silu_mul(const f32x4 gate, const f32x4 up, const float gs, const void *bias,
                                                  const unsigned n, const unsigned n_half) {
    const f32x4 o = silu_mul4<AT>(gate, up, gs, bias, n, n_half);
    uint2 r;
    r.x = pack2(AT{}, o[0], o[1]);
    r.y = pack2(AT{}, o[2], o[3]);
    return r;
}

// Logical n-tile L of a kernel's grid -> physical n-tile of W.  Plain GEMM: identity.  SiLU-mul: consecutive
// logical tiles (2p, 2p+1) are the gate tile p and the up tile p + ntiles/2, so that one wave (even NT) holds both.
__device__ __forceinline__ unsigned physical_tile(unsigned l, unsigned ntiles, unsigned act) {
    return act ? (l >> 1) + (l & 1u) * (ntiles >> 1) : l;
}

// Large-M kernels: which (n-block, m-block) tile this workgroup computes.  Workgroups are dispatched round-robin over
// the 8 XCDs (observed: block b runs on XCD b % 8, each XCD with a private 4 MiB L2), so with the plain mapping the
// workgroups that share a weight panel sit on 8 different L2s and every one of them pulls the panel from HBM.  The
// XCD-aware raster gives each XCD a contiguous chunk of the tile list, m-blocks fastest: the tiles an XCD runs at the
// same time are a few weight panels x ALL their m-blocks, so a panel leaves HBM once per XCD-chunk instead of once per
// m-block (M = 512, gate_up: 1.06 GB -> ~0.3 GB per launch).  A pure speed choice: any placement computes the same.
// Where the large-M kernels put the workgroup with linear id b = blockIdx.y * gridDim.x + blockIdx.x of an nx x ny grid: pure arithmetic, host + device, so the
// bijection is tested on the CPU through petit_raster_tile (tests/test_layout_and_abi.py) and the kernels and the test run the SAME lines.
__host__ __device__ inline void tile_of_linear(unsigned b, unsigned nx, unsigned ny, unsigned flags, unsigned &bn, unsigned &bm) {
    if (!(flags & kFlagXcdRaster)) {
        bm = b / nx, bn = b - bm * nx;
        return;
    }
    const unsigned nwg = nx * ny;
    const unsigned q = nwg >> 3, r = nwg & 7u, xcd = b & 7u, idx = b >> 3;
    const unsigned t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx; // bijective for any nwg
    // An XCD's tiles are consecutive in t and its 32 CUs run 32 consecutive ones at a time.  Column-major t (m fastest)
    // makes those 32 tiles one W panel x 32 A panels once M has 32 m-tiles: at M = 16 k every n-tile column re-reads all of
    // A (gate_up: 224 x 268 MB per call).  Bands of `ph` m-tiles, m fastest inside a band: 32 consecutive tiles are
    // ph A panels x 32 / ph W panels (launch_flags() picks ph so that both cost the same bytes per k-step).
    const unsigned ph = (flags >> kFlagBandShift) & 0xffu;
    if (ph == 0 || ph >= ny) {
        bn = t / ny, bm = t - bn * ny;
        return;
    }
    const unsigned band = t / (ph * nx), tl = t - band * ph * nx;
    const unsigned left = ny - band * ph, rows = ph < left ? ph : left;
    bn = tl / rows, bm = band * ph + (tl - bn * rows);
}
__device__ __forceinline__ void tile_of_block(unsigned flags, unsigned &bn, unsigned &bm) {
    tile_of_linear(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x, gridDim.y, flags, bn, bm);
}

// Two waves share a SIMD in the 2-workgroups-per-CU kernels, run the same phases (unpack burst, MFMA burst, barrier) and,
// sharing the matrix pipe fairly, fall into lock step: both unpack at the same time (matrix pipe idle), both issue MFMAs
// at the same time (VALU idle).  A static priority by the parity of the wave's hardware slot breaks the convoy: the
// favoured wave runs its MFMA burst at full rate while the other fills every gap, so one wave's VALU work overlaps the
// other's MFMAs.  (s_setprio is a scalar instruction; the slot id is wave-uniform by construction.)
__device__ __forceinline__ void stagger_priority(unsigned flags) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (flags & kFlagPrio) {
        const unsigned slot = __builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | ((4 - 1) << 11)); // wave_id[3:0]
        if (slot & 1u)
            __builtin_amdgcn_s_setprio(1);
    }
#endif
}

// Scale record of one span for one n-tile: KS*2 bytes (NV) / KS bytes (MX).
template <int FMT, int KS> struct ScaleRec {
    static constexpr int kBytes = (FMT == kFmtNv ? 2 : 1) * KS;
    static constexpr int kDwords = (kBytes + 3) / 4;
    unsigned d[kDwords];
};

template <int FMT, int KS>
__device__ __forceinline__ ScaleRec<FMT, KS> load_scale_rec(__amdgpu_buffer_rsrc_t r, unsigned voff,
                                                            unsigned soff) {
    ScaleRec<FMT, KS> out;
    constexpr int B = ScaleRec<FMT, KS>::kBytes;
    if constexpr (B == 16) {
        u32x4 v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, kAuxNt));
        out.d[0] = v[0], out.d[1] = v[1], out.d[2] = v[2], out.d[3] = v[3];
    } else if constexpr (B == 8) {
        u32x2 v = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, kAuxNt));
        out.d[0] = v[0], out.d[1] = v[1];
    } else if constexpr (B == 4) {
        out.d[0] = __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, kAuxNt);
    } else {
        static_assert(B == 2, "span record is 2, 4, 8 or 16 bytes");
        out.d[0] = (unsigned)__builtin_amdgcn_raw_buffer_load_b16(r, voff, soff, kAuxNt);
    }
    return out;
}

// The two group scales (NV) or the block scale (MX) of tile T of the span.
template <int FMT, int KS, int T>
__device__ __forceinline__ void tile_scales(const ScaleRec<FMT, KS> &rec, float &s_lo, float &s_hi) {
    if constexpr (FMT == kFmtNv) {
        constexpr int byte = 2 * T;
        s_lo = e4m3_byte<byte % 4>(rec.d[byte / 4]);
        s_hi = e4m3_byte<(byte + 1) % 4>(rec.d[(byte + 1) / 4]);
    } else {
        s_lo = s_hi = e8m0_byte<T % 4>(rec.d[T / 4]);
    }
}

// Fp16Mx: does this lane's MX span record hold a byte outside 114..140?  Per dword x: 140 - b and b - 114 computed for all four
// bytes at once by two plain 32-bit subtractions (no byte of an in-range dword borrows; a byte that does borrow ends with bit 7
// set, and what the borrow does to its neighbour can only flag more), OR-ed over the record: every byte in range <=> the top
// three bits of every byte of both differences are clear (0 <= 140 - b <= 31 and 0 <= b - 114 <= 31  <=>  114 <= b <= 140).
// Returns non-zero for "outside"; wave-uniform use: __builtin_amdgcn_ballot_w64(bad != 0) != 0.
template <int KS> __device__ __forceinline__ unsigned mx_rec_outside_f16(const ScaleRec<kFmtMx, KS> &rec, unsigned acc = 0u) {
#pragma unroll
    for (int d = 0; d < ScaleRec<kFmtMx, KS>::kDwords; ++d) {
        const unsigned x = rec.d[d];
        acc |= (0x8C8C8C8Cu - x) | (x - 0x72727272u);
    }
    return acc & (KS >= 4 ? 0xE0E0E0E0u : 0x0000E0E0u); // (KS = 2: a two-byte record, zero-extended by its load)
}

// The 32-element FP6 converts, issued through inline asm with an EARLY-CLOBBER destination.  hipcc 7.2 treats the builtins
// (__builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32, ..._pk32_fp6_{bf16,f16}) as ordinary VOP3 instructions and is free to place the scale operand (or part of
// a source) inside the 6-register destination -- "v_cvt_scalef32_2xpk16_fp6_f32 v[34:39], v[2:17], v[18:33], v34" in the first build of nvnative.hip --
// and the hardware writes the destination while it still reads its sources: elements 0-1 came out right, the other 30 saturated or flushed to zero
// with the right sign (round 6; tools/probes/dbg_nv6.py).  "=&v" keeps the destination disjoint from every input.
//   element 2 t = a[t], 2 t + 1 = b[t]; element i at bits [6 i, 6 i + 6); dst = RNE_e2m3(src / scale), saturating at 7.5 (tools/probes/mfma32_fp6_probe.hip)
__device__ __forceinline__ u32x6 cvt_2xpk16_fp6_f32(const f32x16v a, const f32x16v b, const float scale) {
    u32x6 out;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_cvt_scalef32_2xpk16_fp6_f32 %0, %1, %2, %3" : "=&v"(out) : "v"(a), "v"(b), "v"(scale));
#else
    out = u32x6{0u, 0u, 0u, 0u, 0u, 0u};
#endif
    return out;
}
// 32 packed 16-bit values (16 dwords) -> 32 e2m3 codes; kBf16: the sources are bf16, else fp16
template <bool kBf16> __device__ __forceinline__ u32x6 cvt_pk32_fp6_16bit(const u32x16 packed, const float scale) {
    u32x6 out;
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (kBf16)
        asm("v_cvt_scalef32_pk32_fp6_bf16 %0, %1, %2" : "=&v"(out) : "v"(packed), "v"(scale));
    else
        asm("v_cvt_scalef32_pk32_fp6_f16 %0, %1, %2" : "=&v"(out) : "v"(packed), "v"(scale));
#else
    out = u32x6{0u, 0u, 0u, 0u, 0u, 0u};
#endif
    return out;
}

// Compile-time loop with a constant index (scale-record bytes, ring slots and
// convert byte-selects must all be literals).
template <int I, int N, class F> __device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

} // namespace petit_amd
