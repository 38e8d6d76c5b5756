// gemm_decode.hpp -- NVFP4 weight streaming for M <= 4 with the group scale applied AFTER the MFMA.
//
// Same job as gemm_stream.hpp (the reference's GemmFp4Fp16KernelGrid, fp4/gemm_fp4_fp16_grid.cuh:441-498, at decode
// batch sizes), same packed layout, different arithmetic.  At M = 1 the streaming kernel is exposed on its unpack:
// an NVFP4 word costs 12 VALU (convert to f32, multiply by the e4m3 group scale, repack to bf16) against 4 for MXFP4,
// whose power-of-two block scale rides in the hardware convert -- and bf16 x MXFP4 measures 0.9 us faster at 8192^2.
// The e4m3 scale cannot ride in the convert, but it does not have to be applied per weight either:
//
//   y[m][n] = sum_groups s[n][grp] * ( sum_{k in grp} a[m][k] * w4[n][k] )
//
// and the 16 x 16 MFMA has 15 idle output rows at M = 1.  So the MFMA runs as D[rho][n] = Amask[rho][k] . W4[k][n]
// with the weights converted WITHOUT their group scale (4 VALU per word) and the activations spread block-diagonally
// over the 16 output rows: row rho of Amask holds a[m][k] for the k of ONE scale group and zeros elsewhere.  D then
// holds per-group partial sums, one per register, and one v_pk_fma_f32 pair folds them into the running total with the
// four group scales of that lane.  The mapping rho -> (m, tile, group) is chosen so that accumulator lane (g', n) needs
// exactly the four scale bytes that lane (r = n, g = g') of the EXISTING scale record already holds (layout.h):
//
//   R = 1 (M = 1):  a D tile spans two k-tiles (8 MFMAs);  rho = 4g' + 2t + h  <->  tile 2G + t, group 2g' + h
//   R = 2 (M <= 2): one k-tile (4 MFMAs);                  rho = 4g' + 2m + h  <->  row m,       group 2g' + h
//   R = 4 (M <= 4): half a k-tile (MFMAs j = 2h, 2h + 1);  rho = 4g' + m       <->  row m,       group 2g' + h
//   R = 8 (M <= 8): 16 D rows hold 8 activation rows x TWO lane groups only, so a k-tile takes two masked passes ("sets"):
//                   pass s covers the weights' lane groups g = 2s, 2s + 1;  rho = 8c + m  <->  row m, lane group g = 2s + c,
//                   group 2g + h.  8 MFMAs per k-tile instead of 4 (the matrix pipe is > 80 % idle at this M), 4 VALU per weight
//                   word instead of 12.  The accumulator lane (g', n) then needs the scales of lane group 2s + (g' >> 1), not its
//                   own: the span's scale record is permuted once per span with two lane swaps per dword ([G0 G0 G1 G1] for s = 0,
//                   [G2 G2 G3 G3] for s = 1).  Pays for bf16 only (the fp16 unpack is 8 VALU per word: neutral).
//
// (MFMA j of a tile covers the k-set {128 kt + 32 g + 8 j + i}: lane group g sees group 2g + (j >> 1) of the tile.)
// The masked activation operand costs no VALU: the wave stages its span of A in a private LDS slice (as the streaming
// kernel does) next to a row of zeros, and every lane reads its fragment with ds_read_b128 from an address that points
// into the data for the (tile, j) it owns and into the zero row otherwise.
//
// Per 256 k of one n-tile at M = 1: 32 converts + 2 scale converts + 2 pk_fma = 36 VALU (96 + 4 in gemm_stream.hpp).
//
// Numerics: the products a * w4 are exact in f32 either way; the scale multiplies a 16-term f32 partial sum instead
// of each weight, i.e. the result differs from the pre-scaled form by f32 rounding of the partial sums only (the same
// class as the MFMA's own unspecified summation order; tests/test_gpu_parity.py holds it to the same tolerance).
// bf16: the weights are converted with a 2^-7 scale (free: the convert takes a scale operand) and the total is
// multiplied back by 2^7 in the epilogue, so that a partial sum of 16 terms |a| < 2^128, |w4| <= 6 cannot overflow f32
// where the pre-scaled form (|s| as small as 2^-9) would not.
#pragma once

#include "device_common.hpp"

namespace petit_amd {

//   AT    Bf16 / Fp16 activations (and output)
//   KS    tiles per span (layout.h)
//   NT    n-tiles per wave
//   WK    waves per workgroup, all along K
//   D     W ring depth in tiles (divides KS)
//   R     activation rows the kernel holds: 1, 2 or 4 (M <= R)
template <class AT_, int KS_, int NT_, int WK_, int D_, int R_> struct DecodeCfg {
    using AT = AT_;
    static constexpr int KS = KS_, NT = NT_, WK = WK_, D = D_, R = R_;
    static constexpr int kThreads = 64 * WK;
    static constexpr int TG = (R == 1) ? 2 : 1;           // k-tiles per D tile
    static constexpr int kARowU4 = KS * 16 + 1;           // one staged row (padded: rows land on different banks)
    static constexpr int kALdsU4 = R * kARowU4;           // R rows per wave; ONE zero row for the whole workgroup behind them (every wave
                                                          // writes the same zeros before it reads them: no barrier; at R = 4, WK = 8 the
                                                          // private zero rows were what kept a second workgroup off the CU: 85 -> 70 KiB)
    static constexpr int kRedFloats = NT * R * 16;        // per wave: y[nt][m][n % 16]
    static constexpr int kSmemU4 = WK * kALdsU4 + kARowU4 + WK * kRedFloats / 4;
    static constexpr int kAStageU4 = R * KS * 16;         // one span of A, in 16-byte units
    static constexpr int kAStageLoads = (kAStageU4 + 63) / 64;
    // resident waves per SIMD the register allocation must allow: four for one n-tile per wave (two 512-thread or four
    // 256-thread workgroups per CU), three otherwise
    static constexpr int kLdsWaves = ((160 * 1024) / (kSmemU4 * 16)) * WK / 4; // what the LDS footprint allows
    static constexpr int kRegWaves = R == 8 ? (NT <= 2 ? 2 : 1) : (NT == 1 && (R == 1 || (R == 2 && D < 8))) ? 4 : 3;
    static constexpr int kWavesPerSimd = kLdsWaves < 1 ? 1 : (kLdsWaves < kRegWaves ? kLdsWaves : kRegWaves);
    static constexpr int kSets = R == 8 ? 2 : 1;          // masked passes per k-tile
    static_assert(R == 1 || R == 2 || R == 4 || R == 8, "rows: 1, 2, 4 or 8");
    static_assert(KS % D == 0 && KS % TG == 0, "ring depth must divide the span");
    static_assert(!AT::kBfp && !AT::kAdaptive, "plain bf16 / fp16 activations");
    static_assert(kSmemU4 * 16 <= 160 * 1024, "LDS budget");
};

// FP4 word -> MFMA operand, no group scale (see the header): bf16 carries 2^-7, fp16 is plain.
__device__ __forceinline__ bf16x8 unpack_unscaled(Bf16, unsigned w) {
    constexpr float kDown = 0.0078125f; // 2^-7
    const bf16x2 q0 = cvt_fp4_bf16<0>(w, kDown), q1 = cvt_fp4_bf16<1>(w, kDown);
    const bf16x2 q2 = cvt_fp4_bf16<2>(w, kDown), q3 = cvt_fp4_bf16<3>(w, kDown);
    return bf16x8{q0[0], q0[1], q1[0], q1[1], q2[0], q2[1], q3[0], q3[1]};
}
__device__ __forceinline__ f16x8 unpack_unscaled(Fp16, unsigned w) {
    const f16x2 q0 = cvt_fp4_f16<0>(w, 1.0f), q1 = cvt_fp4_f16<1>(w, 1.0f);
    const f16x2 q2 = cvt_fp4_f16<2>(w, 1.0f), q3 = cvt_fp4_f16<3>(w, 1.0f);
    return f16x8{q0[0], q0[1], q1[0], q1[1], q2[0], q2[1], q3[0], q3[1]};
}
template <class AT> constexpr float decode_post_scale() { return AT::kType == kDataTypeBf16 ? 128.0f : 1.0f; }

// sum over the four 16-lane rows of a wave, result in every lane (two cross-row swaps)
__device__ __forceinline__ float sum_rows(float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false); // [r0 r0 r2 r2], [r1 r1 r3 r3]
    const unsigned a0 = a[0], a1 = a[1];
    const float s = __builtin_bit_cast(float, a0) + __builtin_bit_cast(float, a1);
    const unsigned su = __builtin_bit_cast(unsigned, s);
    const auto b = __builtin_amdgcn_permlane32_swap(su, su, false, false); // [lo lo], [hi hi]
    const unsigned b0 = b[0], b1 = b[1];
    return __builtin_bit_cast(float, b0) + __builtin_bit_cast(float, b1);
}

// (block_x: the workgroup's index along N -- blockIdx.x for a plain launch, the index inside its member for a grouped one)
template <class Cfg>
__device__ __forceinline__ void gemm_decode_body(const void *arg_w, const void *arg_s, const void *arg_a, unsigned arg_k, unsigned arg_n,
                                                 unsigned arg_m, unsigned arg_spw, unsigned arg_act, void *arg_c, const float *arg_gs,
                                                 const void *arg_bias, const unsigned block_x) {
    using AT = typename Cfg::AT;
    using Frag = typename AT::frag;
    constexpr int KS = Cfg::KS, NT = Cfg::NT, WK = Cfg::WK, D = Cfg::D, R = Cfg::R, TG = Cfg::TG;
    constexpr unsigned kRecBytes = ScaleRec<kFmtNv, KS>::kBytes;
    constexpr unsigned kOob = 0x80000000u;

    // ONE LDS object: [per-wave activation slices (R rows + zero row)][per-wave partial outputs]
    __shared__ u32x4 smem[Cfg::kSmemU4];

    const unsigned lane = threadIdx.x & 63u;
    const unsigned wk = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned r = lane & 15u, g = lane >> 4;

    const unsigned ktiles = arg_k / kTileK;
    const unsigned nspans = ktiles / KS;
    const unsigned ntiles = arg_n / kTileN;
    const unsigned nt0 = block_x * NT;
    const unsigned sp_begin = min(wk * arg_spw, nspans);
    const unsigned sp_end = min(sp_begin + arg_spw, nspans);

    f32x4 total[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
        total[nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    u32x4 *const a_lds = smem + wk * Cfg::kALdsU4;
    if (nt0 < ntiles && sp_begin < sp_end) {
        const unsigned valid_nt = min((unsigned)NT, ntiles - nt0);
        const unsigned w_row_bytes = ktiles * kTileBytes; // one n-tile of W
        const unsigned s_row_bytes = arg_k;               // one n-tile of NV scales
        const unsigned rows = min(arg_m, (unsigned)R);

        const unsigned pt0 = physical_tile(nt0, ntiles, arg_act);
        const unsigned span_tiles = arg_act ? (valid_nt >> 1) + (ntiles >> 1) : valid_nt;
        const __amdgpu_buffer_rsrc_t w_rsrc = make_rsrc((const char *)arg_w + (size_t)pt0 * w_row_bytes, span_tiles * w_row_bytes);
        const __amdgpu_buffer_rsrc_t s_rsrc = make_rsrc((const char *)arg_s + (size_t)pt0 * s_row_bytes, span_tiles * s_row_bytes);
        const __amdgpu_buffer_rsrc_t a_rsrc = make_rsrc(arg_a, rows * arg_k * 2);

        unsigned w_voff[NT], s_voff[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const unsigned rel = physical_tile(nt0 + nt, ntiles, arg_act) - pt0;
            w_voff[nt] = ((unsigned)nt < valid_nt) ? lane * 16 + rel * w_row_bytes : kOob;
            s_voff[nt] = ((unsigned)nt < valid_nt) ? lane * kRecBytes + rel * s_row_bytes : kOob;
        }

        // --- activations: a span of each row, coalesced, through the wave's LDS slice -------------------------------
        // KS >= 4: a wave-load lies inside one row (row and column base are compile-time, only lane * 16 is a VGPR);
        // KS == 2: a wave-load covers two rows of 512 B
        constexpr bool kRowLoads = (KS * 16) % 64 == 0;
        // (validity lives in the VGPR offset: it is bounds-checked against the descriptor on every generation)
        unsigned a_vrow[R];
#pragma unroll
        for (int row = 0; row < R; ++row)
            a_vrow[row] = lane * 16 + row * arg_k * 2;
        unsigned a_voff2 = 0;
        int a_dst2 = 0;
        if constexpr (!kRowLoads) {
            static_assert(kRowLoads || KS == 2, "span of 2, 4 or 8 tiles");
            a_voff2 = (lane >> 5) * arg_k * 2 + (lane & 31u) * 16;
            a_dst2 = (int)((lane >> 5) * Cfg::kARowU4 + (lane & 31u));
        }
        u32x4 astage[Cfg::kAStageLoads];
        auto issue_a_stage = [&](unsigned sp) {
#pragma unroll
            for (int i = 0; i < Cfg::kAStageLoads; ++i) {
                if constexpr (kRowLoads) { // rows >= M fall out of the descriptor: zeros
                    constexpr int kPerRow = KS * 16 / 64;
                    astage[i] = buf_load16(a_rsrc, a_vrow[i / kPerRow] + (i % kPerRow) * 1024, sp * (KS * 256), kAuxDefault);
                } else {
                    const unsigned vo = (2 * i + (lane >> 5) < (unsigned)R) ? a_voff2 : kOob;
                    astage[i] = buf_load16(a_rsrc, vo, sp * (KS * 256) + 2 * i * arg_k * 2, kAuxDefault);
                }
            }
        };
        auto write_a_stage = [&]() {
#pragma unroll
            for (int i = 0; i < Cfg::kAStageLoads; ++i) {
                if constexpr (kRowLoads) {
                    constexpr int kPerRow = KS * 16 / 64;
                    (a_lds + lane)[(i / kPerRow) * Cfg::kARowU4 + (i % kPerRow) * 64] = astage[i];
                } else if (2 * i + (lane >> 5) < (unsigned)R) {
                    (a_lds + a_dst2)[2 * i * Cfg::kARowU4] = astage[i];
                }
            }
        };
        issue_a_stage(sp_begin);
        ScaleRec<kFmtNv, KS> srec[NT], srec_next[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            srec[nt] = load_scale_rec<kFmtNv, KS>(s_rsrc, s_voff[nt], sp_begin * 64 * kRecBytes);
        const unsigned kt_begin = sp_begin * KS;
        u32x4 wring[D][NT];
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                wring[i][nt] = buf_load16(w_rsrc, w_voff[nt], (kt_begin + i) * kTileBytes, kAuxNt);

        // the workgroup's zero row, written (redundantly, by every wave) once: the 128 slots the masked reads can touch
        u32x4 *const zero_row = smem + WK * Cfg::kALdsU4;
#pragma unroll
        for (int i = 0; i < KS * 16; i += 64)
            if (KS * 16 - i >= 64 || lane < (unsigned)(KS * 16 - i))
                (zero_row + lane)[i] = u32x4{0, 0, 0, 0};

        // Fragment pointers: lane (g, rho = r) of the activation operand reads data for the MFMAs whose D row it owns
        // and the zero row for all others.  kBases distinct (tile parity, j >> 1) classes; everything else about a
        // read (k-tile, j & 1) is an immediate offset from the class pointer.
        constexpr int kBases = R == 8 ? 2 : 4 / R;
        const bool mine = (r >> 2) == g;
        const unsigned key = (R == 1) ? (r & 3u) : (R == 2) ? (r & 1u) : 0u;
        const unsigned row = (R == 1) ? 0u : (R == 2) ? ((r >> 1) & 1u) : (R == 4) ? (r & 3u) : (r & 7u);
        const u32x4 *fbase[kBases];
#pragma unroll
        for (int c = 0; c < kBases; ++c) {
            // R = 8: base c = pass (set) c; this lane's operand is live in the pass that covers its lane group, for the class its D row names
            const bool live = R == 8 ? ((g >> 1) == (unsigned)c && (g & 1u) == (r >> 3)) : (mine && key == (unsigned)c);
            fbase[c] = (live ? a_lds + (int)(row * Cfg::kARowU4) : zero_row) + (int)(g * 4);
        }

        write_a_stage();
        constexpr int kSets = Cfg::kSets, kFr = 4 * kSets;
        Frag afrag[kFr], anext[kFr];
        auto read_frags = [&](Frag(&dst)[kFr], auto t_c) { // k-tile T of the span now in LDS
            constexpr int T = decltype(t_c)::value;
#pragma unroll
            for (int st = 0; st < kSets; ++st)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = (R == 8) ? st : (R == 1) ? 2 * (T % 2) + (j >> 1) : (R == 2) ? (j >> 1) : 0;
                    dst[4 * st + j] = __builtin_bit_cast(Frag, fbase[c][T * 16 + j]);
                }
        };
        read_frags(afrag, std::integral_constant<int, 0>{});

        f32x4 dacc[NT]; // the open D tile (R = 1: lives across the two k-tiles of a pair)
        // R = 8: the span's scale records as the two passes need them (lane group g' reads the record of lane group 2 s + (g' >> 1))
        [[maybe_unused]] ScaleRec<kFmtNv, KS> srec8[NT][2];
        auto permute_recs = [&]() {
            if constexpr (R == 8) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int d = 0; d < ScaleRec<kFmtNv, KS>::kDwords; ++d) {
                        const unsigned x = srec[nt].d[d];
                        const auto a16 = __builtin_amdgcn_permlane16_swap(x, x, false, false); // [G0 G0 G2 G2], [G1 G1 G3 G3]
                        const unsigned a0 = a16[0], a1 = a16[1];
                        const auto a32 = __builtin_amdgcn_permlane32_swap(a0, a1, false, false); // [G0 G0 G1 G1], [G2 G2 G3 G3]
                        srec8[nt][0].d[d] = a32[0], srec8[nt][1].d[d] = a32[1];
                    }
            }
        };
        permute_recs();
        auto span_body = [&](const unsigned sp, auto last_c) {
            constexpr bool kLast = decltype(last_c)::value;
            const unsigned kt0 = sp * KS;
            if constexpr (!kLast) {
                issue_a_stage(sp + 1); // held in VGPRs until this span's fragments have been read
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    srec_next[nt] = load_scale_rec<kFmtNv, KS>(s_rsrc, s_voff[nt], (sp + 1) * 64 * kRecBytes);
            }
            static_for<0, KS>([&](auto t_c) {
                constexpr int T = decltype(t_c)::value;
                constexpr int SLOT = T % D;
                constexpr bool kNextInSpan = T + 1 < KS;
                constexpr bool kRefill = !kLast || (T + D < KS);
                if constexpr (kNextInSpan) { // next tile's fragments: the LDS latency hides behind this tile's converts
                    read_frags(anext, std::integral_constant<int, T + 1>{});
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    if constexpr (R == 8) {
                        // two passes per k-tile, each as the R = 4 form: D tiles (h = 0, 1) of MFMAs (0, 1) and (2, 3)
                        Frag wv[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            wv[j] = unpack_unscaled(AT{}, wring[SLOT][nt][j]);
#pragma unroll
                        for (int st = 0; st < 2; ++st) {
                            const f32x2 sv = __builtin_amdgcn_cvt_pk_f32_fp8((int)srec8[nt][st].d[T / 2], (T & 1) != 0);
                            f32x4 d0 = f32x4{0.f, 0.f, 0.f, 0.f}, d1 = f32x4{0.f, 0.f, 0.f, 0.f};
                            d0 = mfma16(afrag[4 * st + 0], wv[0], d0);
                            d1 = mfma16(afrag[4 * st + 2], wv[2], d1);
                            d0 = mfma16(afrag[4 * st + 1], wv[1], d0);
                            d1 = mfma16(afrag[4 * st + 3], wv[3], d1);
                            total[nt] += d0 * sv[0] + d1 * sv[1];
                        }
                    } else if constexpr (R == 4) {
                        // two D tiles per k-tile: MFMAs 0, 1 see group 2g, MFMAs 2, 3 group 2g + 1
                        const f32x2 sv = __builtin_amdgcn_cvt_pk_f32_fp8((int)srec[nt].d[T / 2], (T & 1) != 0);
                        f32x4 d0 = f32x4{0.f, 0.f, 0.f, 0.f}, d1 = f32x4{0.f, 0.f, 0.f, 0.f};
                        d0 = mfma16(afrag[0], unpack_unscaled(AT{}, wring[SLOT][nt][0]), d0);
                        d1 = mfma16(afrag[2], unpack_unscaled(AT{}, wring[SLOT][nt][2]), d1);
                        d0 = mfma16(afrag[1], unpack_unscaled(AT{}, wring[SLOT][nt][1]), d0);
                        d1 = mfma16(afrag[3], unpack_unscaled(AT{}, wring[SLOT][nt][3]), d1);
                        total[nt] += d0 * sv[0] + d1 * sv[1];
                    } else {
                        if constexpr (T % TG == 0)
                            dacc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            dacc[nt] = mfma16(afrag[j], unpack_unscaled(AT{}, wring[SLOT][nt][j]), dacc[nt]);
                        if constexpr (T % TG == TG - 1) { // D tile complete: x the four scales of this lane's four rows
                            f32x4 sv;
                            if constexpr (R == 1) {
                                const f32x2 lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)srec[nt].d[T / 2], false);
                                const f32x2 hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)srec[nt].d[T / 2], true);
                                sv = f32x4{lo[0], lo[1], hi[0], hi[1]};
                            } else {
                                const f32x2 p = __builtin_amdgcn_cvt_pk_f32_fp8((int)srec[nt].d[T / 2], (T & 1) != 0);
                                sv = f32x4{p[0], p[1], p[0], p[1]};
                            }
                            total[nt] += dacc[nt] * sv;
                        }
                    }
                }
                if constexpr (kRefill) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        wring[SLOT][nt] = buf_load16(w_rsrc, w_voff[nt], (kt0 + T + D) * kTileBytes, kAuxNt);
                }
                if constexpr (kNextInSpan) {
#pragma unroll
                    for (int j = 0; j < kFr; ++j)
                        afrag[j] = anext[j];
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            if constexpr (!kLast) {
                // every fragment of this span has been read (LDS is in order within a wave)
                write_a_stage();
                read_frags(afrag, std::integral_constant<int, 0>{});
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    srec[nt] = srec_next[nt];
                permute_recs();
            }
        };
        for (unsigned sp = sp_begin; sp + 1 < sp_end; ++sp)
            span_body(sp, std::false_type{});
        span_body(sp_end - 1, std::true_type{});
    }

    // --- y[m][n] of this wave: fold the D rows, sum the four lane rows, park in LDS ------------------------------
    float *const red = reinterpret_cast<float *>(smem + WK * Cfg::kALdsU4 + Cfg::kARowU4);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        if constexpr (R == 8) {
            // D row rho = 8c + m: accumulator lane (g', n) register i holds row m = 4 (g' & 1) + i; the two classes c = g' >> 1 are
            // partial sums of the same output: add lanes l and l ^ 32, then lane rows 0 and 1 park rows 0-3 and 4-7
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float v = total[nt][i];
                const unsigned u = __builtin_bit_cast(unsigned, v);
                const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
                const unsigned s0 = sw[0], s1 = sw[1];
                const float sum = __builtin_bit_cast(float, s0) + __builtin_bit_cast(float, s1);
                if (g < 2u)
                    red[((wk * NT + nt) * R + 4 * g + i) * 16 + r] = sum;
            }
            continue;
        }
        float y[R];
        if constexpr (R == 1)
            y[0] = (total[nt][0] + total[nt][1]) + (total[nt][2] + total[nt][3]);
        else if constexpr (R == 2)
            y[0] = total[nt][0] + total[nt][1], y[1] = total[nt][2] + total[nt][3];
        else
            y[0] = total[nt][0], y[1] = total[nt][1], y[2] = total[nt][2], y[3] = total[nt][3];
        float mine_v = 0.f;
#pragma unroll
        for (int m = 0; m < R; ++m) {
            const float s = sum_rows(y[m]);
            mine_v = (g == (unsigned)m) ? s : mine_v;
        }
        if (g < (unsigned)R) // lane row g parks output row m = g
            red[((wk * NT + nt) * R + g) * 16 + r] = mine_v;
    }
    __syncthreads();

    // --- K reduction across the waves, epilogue: 4 consecutive n of one m per thread ------------------------------
    const float gs = *arg_gs * decode_post_scale<AT>();
    constexpr unsigned kItems = NT * R * 4;
    static_assert(kItems <= (unsigned)Cfg::kThreads, "one pass");
    const unsigned item = threadIdx.x;
    if (item < kItems) {
        const unsigned q = item & 3u, m = (item >> 2) % R, nt = item / (4 * R);
        auto gather = [&](unsigned inn) {
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < WK; ++w)
                v += *reinterpret_cast<const f32x4 *>(&red[((w * NT + inn) * R + m) * 16 + q * 4]);
            return v;
        };
        const unsigned ntile = nt0 + nt;
        if (m < arg_m && ntile < ntiles) {
            if (arg_act) {
                if constexpr (NT % 2 == 0) {
                    if ((nt & 1u) == 0) { // logical tiles (nt, nt + 1) = gate / up halves of output tile ntile / 2
                        const unsigned n_half = arg_n >> 1, n = (ntile >> 1) * 16 + q * 4;
                        *reinterpret_cast<uint2 *>((char *)arg_c + ((size_t)m * n_half + n) * 2) =
                            finish4_silu_mul<AT>(gather(nt), gather(nt + 1), gs, arg_bias, n, n_half);
                    }
                }
            } else {
                const unsigned n = ntile * 16 + q * 4;
                *reinterpret_cast<uint2 *>((char *)arg_c + ((size_t)m * arg_n + n) * 2) = finish4<AT>(gather(nt), gs, arg_bias, n);
            }
        }
    }
}

template <class Cfg>
__global__ __launch_bounds__(Cfg::kThreads, Cfg::kWavesPerSimd) void gemm_decode_kernel(const void *arg_w, const void *arg_s, const void *arg_a,
                                                                    unsigned arg_k, unsigned arg_n, unsigned arg_m,
                                                                    unsigned arg_spw, unsigned arg_act, void *arg_c,
                                                                    const float *arg_gs, const void *arg_bias) {
    gemm_decode_body<Cfg>(arg_w, arg_s, arg_a, arg_k, arg_n, arg_m, arg_spw, arg_act, arg_c, arg_gs, arg_bias, blockIdx.x);
}

// Several weight matrices that share the activation rows (q / k / v shards, gate and up kept apart, the experts of a token) in ONE
// launch: the grid is the concatenation of the members' grids, a workgroup finds its member by a scalar search over <= 8 prefix
// sums.  The dependent-dispatch gap (~1.6 us on MI355X, a quarter to a half of a TP-8 shard's whole GEMM) is paid once per group.
template <class Cfg>
__global__ __launch_bounds__(Cfg::kThreads, Cfg::kWavesPerSimd) void gemm_decode_grouped_kernel(const GroupTable g, const void *arg_a, unsigned arg_k,
                                                                            unsigned arg_m, unsigned arg_spw) {
    unsigned idx = 0, base = 0;
    for (unsigned i = 0; i + 1 < g.count; ++i)
        if (blockIdx.x >= g.wg_end[i])
            idx = i + 1, base = g.wg_end[i];
    gemm_decode_body<Cfg>(g.w[idx], g.s[idx], arg_a, arg_k, g.n[idx], arg_m, arg_spw, 0u, g.c[idx], g.gs[idx], g.bias[idx], blockIdx.x - base);
}

} // namespace petit_amd
