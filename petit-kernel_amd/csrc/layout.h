// layout.h -- the packed ("petit-cdna4") weight / scale layout of this library.
//
// The reference shuffles weights offline into a lane-major 64(K) x 32(N) tile
// and re-encodes every nibble so that its CDNA2/3 bit tricks can unpack it
// (lib/gemm/rocm/quantization/fp4/quantization_utils.cu:177-206, SURVEY.md
// Appendix A).  gfx950 converts raw E2M1 nibbles in hardware
// (v_cvt_scalef32_pk_{bf16,f16,f32}_fp4) and consumes raw nibbles in
// v_mfma_scale_f32_16x16x128_f8f6f4, so this layout keeps the nibbles raw and
// only shuffles whole 32-bit words.  The Python/C++ contracts are unchanged:
// same entry points, same output shapes and dtypes, opaque contents.
//
// Weights   in : u32 qw[N][K/8]          nibble i of word k8 = element 8*k8+i
//           out: uint4 pw[N/16][K/128][64]                (same byte count)
//                tile (nt, kt) is 1 KiB = one dwordx4 per lane of a wave;
//                lane l = 16*g + r  (r = n%16, g = (k%128)/32) holds
//                qw[16*nt + r][16*kt + 4*g + j],  j = 0..3
//                i.e. one weight row, 32 consecutive k.
//                Python view: int32 [N/16, 2K] (row = one n-tile), as
//                lib/pybind/fp4.cc:62-63.
//
// MFMA mapping (dequant path, v_mfma_f32_16x16x32_{bf16,f16}): word j of every
// lane forms the 16(n) x 32(k) operand of MFMA j of the tile, whose k-set is
// { 128*kt + 32*g + 8*j + i : g < 4, i < 8 }.  The reduction over k is a sum,
// so any k-permutation is legal as long as the activation fragment uses the
// same one: lane (m = l%16, g = l/16) reads A[m][128*kt + 32*g + 8*j .. +7].
// Native path (v_mfma_scale_f32_16x16x128_f8f6f4): the lane's uint4 IS the
// FP4 operand (32 consecutive k = one MX block) -- no unpack at all.
//
// NV scales in : u8 e4m3 s[N][K/16]
//           out: u8 ps[N/16][K/(128*KS)][64][KS][2]        (same byte count)
//                lane (r, g) of tile kt needs groups 2g, 2g+1 of its row:
//                bytes [t][0..1] = s[16*nt + r][8*(KS*sp + t) + 2*g + {0,1}]
//                for the KS tiles t of span sp.  One span record is KS*2
//                bytes per lane (16 B for KS = 8 -> one dwordx4 per 8 KiB of
//                weights).  Raw e4m3 bytes (decoded by v_cvt_f32_fp8).
//                Python view: float8_e4m3fn [N, K/16] (fp4.cc:101-107).
//
// MX scales in : u8 e8m0 s[N][K/32]
//           out: u8 ps[N/16][K/(128*KS)][64][KS]
//                byte [t] = s[16*nt + r][4*(KS*sp + t) + g].
//                Python view: uint8 [N/32, K] (fp4.cc:142-148).
//
// KS ("span tiles") is a pure function of K, so repack and GEMM agree without
// any side channel: 8 when K % 1024 == 0, else 4 when K % 512 == 0, else 2
// (K % 256 == 0 is the contract of process_*_scales, fp4.cc:82-84,126-128).
#pragma once

#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define PETIT_HD __host__ __device__ inline
#else
#define PETIT_HD inline
#endif

namespace petit_amd {

constexpr int kTileN = 16;       // weight rows per tile
constexpr int kTileK = 128;      // k per tile (32 per lane)
constexpr int kLaneK = 32;       // consecutive k owned by one lane
constexpr int kNvGroup = 16;     // NVFP4 scale group (fp4.cc:167-171)
constexpr int kMxGroup = 32;     // MXFP4 scale group (fp4.cc:132-135)
constexpr int kTileBytes = 1024; // 64 lanes x 16 B

PETIT_HD int span_tiles_for_k(unsigned k) {
    return (k % 1024u == 0) ? 8 : (k % 512u == 0) ? 4 : 2;
}

// u32 index of native word qw[n][k8] inside the packed weight buffer.
PETIT_HD size_t packed_weight_word_index(unsigned k_total, unsigned n, unsigned k8) {
    const unsigned nt = n / 16, r = n % 16;
    const unsigned kt = k8 / 16, g = (k8 % 16) / 4, j = k8 % 4;
    const size_t tile = (size_t)nt * (k_total / kTileK) + kt;
    return (tile * 64 + (g * 16 + r)) * 4 + j;
}

// byte index of NV scale s[n][grp] (grp = k/16) inside the packed buffer.
PETIT_HD size_t packed_nvscale_byte_index(unsigned k_total, unsigned n, unsigned grp) {
    const unsigned ks = (unsigned)span_tiles_for_k(k_total);
    const unsigned nt = n / 16, r = n % 16;
    const unsigned kt = grp / 8, g = (grp % 8) / 2, h = grp % 2;
    const unsigned sp = kt / ks, t = kt % ks;
    const size_t rec = ((size_t)nt * (k_total / (kTileK * ks)) + sp) * 64 + (g * 16 + r);
    return rec * (ks * 2) + t * 2 + h;
}

// byte index of MX scale s[n][blk] (blk = k/32) inside the packed buffer.
PETIT_HD size_t packed_mxscale_byte_index(unsigned k_total, unsigned n, unsigned blk) {
    const unsigned ks = (unsigned)span_tiles_for_k(k_total);
    const unsigned nt = n / 16, r = n % 16;
    const unsigned kt = blk / 4, g = blk % 4;
    const unsigned sp = kt / ks, t = kt % ks;
    const size_t rec = ((size_t)nt * (k_total / (kTileK * ks)) + sp) * 64 + (g * 16 + r);
    return rec * ks + t;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// "petit-cdna4-nv6/1": the MFMA-native image of NVFP4 weights (nvnative.hip builds it ONCE at load time; gemm_native32.hpp, WF = 6,
// consumes it).  NVFP4's e4m3 group-16 scales do not fit the block-scaled MFMA (one E8M0 scale per 32 k), so every 32-k block of a
// weight row is re-encoded:  E = floor(log2(max |fp4 x scale| of the block)) - 2,  element = RNE_e2m3(fp4 x scale / 2^E)  (FP6 e2m3:
// the block maximum lands in [4, 7.5]; never saturates, see nvnative.hip), scale byte = E + 127.  A different ACCURACY CLASS
// (petit_amd.h "Native-FP4 kernels"): 4-significant-bit elements where fp4 x e4m3 has up to 6.
//   elements : u8 img[N32][K/128][3][64][16]      N32 = ceil(N / 32) blocks of 32 weight rows (rows >= N: zeros)
//              the instruction's own operand geometry -- lane l = 32 h + r of block (b, kt) holds row 32 b + r;
//              plane 0: registers 0-3 of operand P1 (k = 128 kt + 32 h + e, e = 0 .. 31: element e at bits [6 e, 6 e + 6) of the lane's 192);
//              plane 1: registers 0-3 of operand P2 (k = 128 kt + 64 + 32 h + e);  plane 2: {P1 reg 4, P1 reg 5, P2 reg 4, P2 reg 5}.
//              6 bits per weight: 3 KiB per (block, k-tile), one contiguous stream per block along K.
//   scales   : u8 sc[N32][K/(128 KS)][64][2][KS]   lane (r, h): [q][t] = E8M0 byte of block (k / 32) = 4 (KS sp + t) + 2 q + h
constexpr unsigned kNv6TileBytes = 3072;
PETIT_HD size_t nv6_elem_bytes(unsigned n, unsigned k) { return (size_t)((n + 31) / 32) * (k / kTileK) * kNv6TileBytes; }
PETIT_HD size_t nv6_scale_bytes(unsigned n, unsigned k) { return (size_t)((n + 31) / 32) * (k / kTileK) * 128; }
PETIT_HD size_t nv6_image_bytes(unsigned n, unsigned k) { return nv6_elem_bytes(n, k) + nv6_scale_bytes(n, k); }
// byte offset (inside the element part) of the 16-byte unit `plane` of lane (row n % 32, h) of k-tile kt
PETIT_HD size_t nv6_unit_offset(unsigned k_total, unsigned n, unsigned kt, unsigned plane, unsigned h) {
    return ((size_t)(n / 32) * (k_total / kTileK) + kt) * kNv6TileBytes + plane * 1024 + (h * 32 + n % 32) * 16;
}
// byte offset (inside the scale part) of the E8M0 byte of 32-k block blk of row n
PETIT_HD size_t nv6_scale_offset(unsigned k_total, unsigned n, unsigned blk) {
    const unsigned ks = (unsigned)span_tiles_for_k(k_total);
    const unsigned kt = blk / 4, q = (blk % 4) / 2, h = blk % 2, sp = kt / ks, t = kt % ks;
    return (((size_t)(n / 32) * (k_total / (kTileK * ks)) + sp) * 64 + (h * 32 + n % 32)) * (2 * ks) + q * ks + t;
}

} // namespace petit_amd
