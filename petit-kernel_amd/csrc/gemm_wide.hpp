// gemm_wide.hpp -- the FP4 GEMM for large M on the 32x32x16 MFMA (prefill regime, MFMA-bound).
//
// Same contract, packed layout and unpack as gemm_tiled.hpp; what changes is the matrix instruction and the
// way its operands are kept fed:
//  * v_mfma_f32_32x32x16_{bf16,f16}: one instruction does the work of two 16x16x32 ones, so a k-step issues half
//    as many MFMAs and reads half as many operand registers per flop (the chip is power-limited under a full MFMA
//    load: fewer instructions per flop is clock).
//  * The packed layout stores 16-row tiles (lane 16g + r = row r, k-chunk g), the 32x32 instruction wants 32 rows per
//    k-half.  Two neighbouring n-tiles X, Y are merged IN REGISTERS on the still-packed 4-bit words with two lane
//    swaps per register (v_permlane16_swap, v_permlane32_swap):
//        X = [X.g0 X.g1 X.g2 X.g3], Y = [Y.g0 ..]   ->   P1 = [X.g0 Y.g0 X.g1 Y.g1],  P2 = [X.g2 Y.g2 X.g3 Y.g3]
//    so lanes 0-31 of P1 are 32 weight rows with k-chunk 0 and lanes 32-63 the same rows with k-chunk 1: word j of
//    P1 is the 32(n) x 16(k) operand of one MFMA (k = {8j.. , 32 + 8j..}), word j of P2 the same for k-chunks 2, 3.
//    8 swaps per 2 KiB of weights, against 96 unpack VALU; the scale records are swapped the same way once per
//    span, so every lane keeps the scales of the words it now holds.
//  * The unpack is software-pipelined INTO the MFMA stream: while the MFMAs of k-step t run from one set of unpacked
//    fragments, the words of step t + 1 are merged and unpacked into a second set, one word per half-block, so the
//    12 VALU per word sit between MFMAs of the same wave instead of in a burst during which the matrix pipe idles
//    (measured before: MFMA busy 52 %, VALU busy 36 %, the two adding up instead of overlapping).
//  * A k-step is 8 groups (k-half set q, word j); a group issues MB*NP MFMAs on MB*NP DIFFERENT accumulators, the
//    fragments of group g + 1 are read from LDS while they run.  Independent back-to-back MFMAs matter: anything issued
//    between two MFMAs on the same accumulator costs ~43 cycles (MI355X_MICROARCH.md), which is what held the
//    16x16 kernel and the first 32x32 version at half the matrix rate.
//  * A tile through LDS by direct global -> LDS loads, un-padded XOR-swizzled rows, one barrier per k-step; W never
//    touches LDS; K split across workgroups (gridDim.z) as in gemm_tiled.hpp.
// Reference counterpart: fp4/gemm_fp4_fp16_grid.cuh:323-498 + warp_schedule_fp16.cuh (its 16x16x16 / 32x32x8 MFMA
// schedule for CDNA2/3).
#pragma once

#include "gemm_tiled.hpp"

namespace petit_amd {

typedef __attribute__((ext_vector_type(16))) float f32x16;

// Ablation builds (tools/ablate_wide.sh; 0 in the shipped library): 1 no A-tile DMA after the prologue and no barrier,
// 2 no W ring refill, 4 no unpack VALU, 8 fragments read once per step only, 16 no MFMA.  Results are garbage; only time counts.
#ifndef PETIT_ABLATE
#define PETIT_ABLATE 0
#endif

__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ void pin_here(unsigned &x) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(x));
#endif
}
__device__ __forceinline__ void pin_here(u32x4 &x) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(x));
#endif
}

// (x, y) -> (p1, p2) as described above, one register of each tile.
__device__ __forceinline__ void merge_tiles(unsigned x, unsigned y, unsigned &p1, unsigned &p2) {
    const auto a = __builtin_amdgcn_permlane16_swap(x, y, false, false); // [X0 Y0 X2 Y2], [X1 Y1 X3 Y3]
    const unsigned a0 = a[0], a1 = a[1];
    const auto b = __builtin_amdgcn_permlane32_swap(a0, a1, false, false); // [X0 Y0 X1 Y1], [X2 Y2 X3 Y3]
    p1 = b[0], p2 = b[1];
}

//   MB    m32-blocks per workgroup = per wave                  (BM = 32*MB)
//   NP    n32-blocks (pairs of n-tiles) per wave               (BN = 32*NP*WAVES)
//   WAVES waves per workgroup (along N)
//   D     W ring depth in k-tiles
//   PF    A-tile prefetch distance in k-steps: 1 = double buffer, the tile of step t + 1 is requested at the top of step t
//         and waited for at its end (one step of latency cover: with one workgroup per CU the step time IS the L2
//         round trip, ~1.1 us, whatever the compute does -- measured: every restructuring of the compute landed on
//         the same 70 us at M = 512, N = K = 8192); 2 = three LDS buffers, the tile of step t + 2 is requested at the
//         top of step t and stays in flight across the barrier (raw s_barrier + counted vmcnt: __syncthreads() would
//         drain it), two steps of cover.
//   KG    wave GROUPS along K inside the workgroup (1 or 2), as Native32Cfg (gemm_native32.hpp): two complete copies of the wave set, each with its
//         own LDS buffers and weight ring, half of the workgroup's K range each, accumulators summed through LDS before the epilogue -- two waves
//         per SIMD for grids of ONE 128 x 128 tile per CU (`o`, qkv at M = 512), where gate_up gets them from two resident workgroups.
//   GA    1 = "group ahead" (round 6, the 256 x 256 tile: MB = 8, NP = 2): the weight fragments are unpacked ONE GROUP ahead of their MFMAs instead
//         of one step ahead -- 2 NP fragments (16 registers) in flight instead of the double-buffered step (128 registers at NP = 2) -- so that the
//         256 architectural registers hold the activation fragments (64), the ring and the records while the 16 accumulator tiles (256 registers)
//         live in AGPRs: one wave per SIMD, 16 MFMAs per group on 16 different accumulators, the W unpack amortised over 8 m-blocks and one
//         activation-fragment read per two 32x32x16 MFMAs -- hipBLASLt's instruction mix at this M (MT256x256x64: 1.08 VALU and 0.25 LDS reads per
//         16x16x32 MFMA, profiles/r06_prefill_pmc.json; the 128 x 256 tiled kernel: 2.9 (NVFP4) / 1.8 (MXFP4) VALU per MFMA).
template <class AT_, int FMT_, int KS_, int MB_, int NP_, int WAVES_, int D_, int PF_ = 1, int KG_ = 1, int GA_ = 0> struct WideCfg {
    using AT = AT_;
    static constexpr int GA = GA_;
    static_assert(GA_ == 0 || (KG_ == 1 && PF_ == 1 && !AT_::kAdaptive), "group-ahead form: one K group, double-buffered A tile, plain 16-bit activations");
    static constexpr int FMT = FMT_, KS = KS_, MB = MB_, NP = NP_, WAVES = WAVES_, D = D_, PF = PF_, NBUF = PF_ + 1, KG = KG_;
    static constexpr int kThreads = 64 * WAVES * KG;
    static constexpr int BM = 32 * MB, BN = 32 * NP * WAVES;
    static constexpr int kBufU4 = BM * 16;                  // one A tile: BM rows x 16 units of 16 B, XOR-swizzled
    static constexpr int kDmaLoads = BM * 16 / 64 / WAVES;  // 1 KiB wave-loads per wave per tile
    // 64 accumulator registers per wave leave room for two waves per SIMD (two workgroups per CU)
    static constexpr int kMinWavesPerSimd = (KG == 2 || (NP == 1 && MB <= 4 && PF == 1)) ? 2 : 1; // (NP = 2 doubles the unpacked-fragment sets)
    static constexpr int kRedU4 = KG == 2 ? BM * BN / 4 : 0;  // KG = 2: the second group's accumulators, f32
    static constexpr int kSmemU4 = KG * NBUF * kBufU4 > kRedU4 ? KG * NBUF * kBufU4 : kRedU4;
    static_assert(KG == 1 || KG == 2, "one or two K groups");
    static_assert(PF == 1 || PF == 2, "prefetch distance 1 or 2");
    static_assert(!AT::kBfp, "plain bf16 / fp16 activations (or Fp16Mx: fast body + exact fallback, device_common.hpp)");
    static_assert(!AT::kAdaptive || FMT == kFmtMx, "Fp16Mx: fp16 activations x MXFP4 weights");
    static_assert(KS % D == 0, "ring depth must divide the span");
    static_assert((BM * 16) % (64 * WAVES) == 0 && (4 * WAVES) % 16 == 0, "A tile must split into whole 16-row groups per wave-load");
    static_assert(kSmemU4 * 16 <= 160 * 1024, "LDS budget");
};

template <class Cfg>
__global__ __launch_bounds__(Cfg::kThreads, Cfg::kMinWavesPerSimd) void gemm_wide_kernel(const GemmArgs p) {
    using AT = typename Cfg::AT;
    using Frag = typename AT::frag;
    constexpr int FMT = Cfg::FMT, KS = Cfg::KS, MB = Cfg::MB, NP = Cfg::NP, WAVES = Cfg::WAVES, D = Cfg::D;
    constexpr int PF = Cfg::PF, NBUF = Cfg::NBUF;
    constexpr unsigned kRecBytes = ScaleRec<FMT, KS>::kBytes;
    constexpr int kRecDw = ScaleRec<FMT, KS>::kDwords;
    constexpr unsigned kOob = 0x80000000u;

    constexpr int KG = Cfg::KG;
    __shared__ u32x4 smem_all[Cfg::kSmemU4];

    const unsigned tid = threadIdx.x;
    const unsigned lane = tid & 63u;
    const unsigned wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned kg = KG == 1 ? 0u : wave_all / WAVES;           // K group of this wave
    const unsigned wave = KG == 1 ? wave_all : wave_all % WAVES;   // index inside the group
    u32x4 *const smem = smem_all + kg * (NBUF * Cfg::kBufU4);      // this group's A-tile buffers
    const unsigned m_l = lane & 31u, h = lane >> 5; // fragment row (activation row / weight row of the pair), k-half

    const unsigned ktiles = p.k / kTileK;
    const unsigned nspans = ktiles / KS;
    const unsigned ntiles = p.n / kTileN;
    unsigned bn, bm;
    tile_of_block(p.flags, bn, bm);
    stagger_priority(p.flags);
    const unsigned nt0 = (bn * WAVES + wave) * (2 * NP); // first logical n-tile of this wave
    const unsigned m0 = bm * Cfg::BM;
    // K slice of this wave group: part blockIdx.z * KG + kg (gemm_native32.hpp: a part past the end walks the last span with its weight rows masked)
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
    const unsigned s_row_bytes = (FMT == kFmtNv) ? p.k : p.k / 2;
    const unsigned rows = min(p.m - m0, (unsigned)Cfg::BM);
    const unsigned pt0 = valid_nt ? physical_tile(nt0, ntiles, p.act) : 0u;
    const unsigned span_tiles = !valid_nt ? 0u : p.act ? (valid_nt >> 1) + (ntiles >> 1) : valid_nt;
    const __amdgpu_buffer_rsrc_t w_rsrc = make_rsrc((const char *)p.w + (size_t)pt0 * w_row_bytes, span_tiles * w_row_bytes);
    const __amdgpu_buffer_rsrc_t s_rsrc = make_rsrc((const char *)p.s + (size_t)pt0 * s_row_bytes, span_tiles * s_row_bytes);
    const __amdgpu_buffer_rsrc_t a_rsrc = make_rsrc((const char *)p.a + (size_t)m0 * p.k * 2, rows * p.k * 2);

    unsigned w_voff[2 * NP], s_voff[2 * NP];
#pragma unroll
    for (int nt = 0; nt < 2 * NP; ++nt) {
        const unsigned rel = physical_tile(nt0 + nt, ntiles, p.act) - pt0;
        w_voff[nt] = ((unsigned)nt < valid_nt) ? lane * 16 + rel * w_row_bytes : kOob;
        s_voff[nt] = ((unsigned)nt < valid_nt) ? lane * kRecBytes + rel * s_row_bytes : kOob;
    }
    // direct-to-LDS staging of the A tile, exactly as gemm_tiled.hpp: wave-load i of this wave covers rows
    // 4*(i*WAVES + wave) .. +3; lane l -> row + l/16, position l%16, which receives unit (l%16) ^ (row%16)
    const unsigned dma_row0 = wave * 4 + (lane >> 4);
    const unsigned dma_voff = dma_row0 * p.k * 2 + (((lane & 15u) ^ (dma_row0 & 15u)) * 16);
    auto dma_a_tile = [&](u32x4 *dst, unsigned kt) {
#pragma unroll
        for (int i = 0; i < Cfg::kDmaLoads; ++i) {
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (__attribute__((address_space(3))) void *)(dst + (i * WAVES + wave) * 64),
                                                     16, dma_voff, i * (4 * WAVES) * p.k * 2 + kt * 256, 0, 0);
#else
            (void)dst, (void)kt;
#endif
        }
    };
    // fragment of (m32-block mb, group g = 4 q + j: k-half set q, word j): row 32 mb + m_l, 16-byte unit 8 q + 4 h + j
    const unsigned frag_row = m_l * 16, frag_swz = m_l & 15u;
    auto read_frags = [&](const u32x4 *a_cur, int g, u32x4 *dst) {
        const int q = g >> 2, j = g & 3;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
            dst[mb] = a_cur[mb * 32 * 16 + frag_row + ((unsigned)(8 * q + 4 * h + j) ^ frag_swz)];
    };

    // Fp16Mx: this kernel unpacks one step AHEAD of its MFMAs, so a switch between the fast and the fallback body in mid-stream would need
    // a step that is half of each.  The wave decides ONCE, up front, for its whole K slice instead: the records of one n-tile over the spans
    // [sp_begin, sp_end) are one contiguous byte range, scanned 16 bytes per lane (a few independent loads, issued ahead of the prologue's
    // own and answered in the same round trip).
    bool mx_fb = false;
    if constexpr (AT::kAdaptive) {
        const unsigned range = (sp_end - sp_begin) * 64u * kRecBytes; // bytes of one n-tile's records in this slice
        unsigned bad = 0;
#pragma unroll
        for (int nt = 0; nt < 2 * NP; ++nt) {
            if ((unsigned)nt >= valid_nt)
                continue;
            const unsigned row0 = s_voff[nt] - lane * kRecBytes + sp_begin * 64u * kRecBytes;
#pragma unroll 4
            for (unsigned base = 0; base < range; base += 1024u) {
                const unsigned off = base + lane * 16u;
                const u32x4 v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(s_rsrc, off < range ? row0 + off : kOob, 0, kAuxDefault));
                unsigned t = 0;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const unsigned x = v[d];
                    t |= (0x8C8C8C8Cu - x) | (x - 0x72727272u); // mx_rec_outside_f16, device_common.hpp
                }
                bad |= off < range ? t : 0u;
            }
        }
        mx_fb = __builtin_amdgcn_ballot_w64((bad & 0xE0E0E0E0u) != 0) != 0;
    }

    // --- prologue.  A tiles live in buffer (k-tile index relative to the slice) % NBUF.
    const unsigned kt_end = sp_end * KS;
    dma_a_tile(smem, kt_begin);
    if constexpr (PF == 2) {
        if (kt_begin + 1 < kt_end)
            dma_a_tile(smem + Cfg::kBufU4, kt_begin + 1);
    }
    if (AT::kAdaptive && mx_fb) {
        // Fp16Mx fallback for this wave's whole K slice: exact for any e8m0 scale, written for size, not speed (device_common.hpp).  One k-tile
        // per trip of a rolled loop: the pair's two W tiles loaded on the spot and merged as in the fast body, the scale bytes loaded by
        // themselves (lane (n, h) of the merged operand P1 / P2 needs the byte of row n % 16 of tile n / 16, k-chunk h / 2 + h: the lane
        // 16 (2 q + h) + n % 16 of that tile's record), weights to bf16, every fp16 fragment split into hi + lo bf16 in registers, two
        // MFMAs per word.  It keeps the workgroup's protocol: its share of the activation-tile DMA PF steps ahead, one barrier per step.
        if constexpr (AT::kAdaptive) {
            __syncthreads(); // the prologue's barrier: tile kt_begin is in buffer 0
            unsigned rel = 0; // (k-tile index relative to the slice) % NBUF
#pragma unroll 1
            for (unsigned kt = kt_begin; kt < kt_end; ++kt) {
                const u32x4 *const a_cur = smem + rel * Cfg::kBufU4;
                if constexpr (PF == 1) {
                    if (kt + 1 < kt_end)
                        dma_a_tile(smem + (rel ^ 1u) * Cfg::kBufU4, kt + 1);
                } else {
                    dma_a_tile(smem + (rel == 0 ? 2u : rel - 1) * Cfg::kBufU4, kt + 2); // (rel + 2) % 3; past K: zeros nobody reads
                }
#pragma unroll
                for (int np = 0; np < NP; ++np) {
                    const u32x4 x = buf_load16(w_rsrc, w_voff[2 * np], kt * kTileBytes, kAuxDefault);
                    const u32x4 y = buf_load16(w_rsrc, w_voff[2 * np + 1], kt * kTileBytes, kAuxDefault);
                    float sc[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        // lanes 0-15 / 16-31 / 32-47 / 48-63 of P(q+1): [X.g(2q) Y.g(2q) X.g(2q+1) Y.g(2q+1)]
                        const unsigned tile = (lane >> 4) & 1u, src_lane = 16u * (2u * q + (lane >> 5)) + (lane & 15u);
                        const unsigned voff = s_voff[2 * np] == kOob ? kOob : s_voff[2 * np] - lane * kRecBytes + src_lane * kRecBytes + kt % KS;
                        const unsigned voff_y = s_voff[2 * np + 1] == kOob ? kOob : s_voff[2 * np + 1] - lane * kRecBytes + src_lane * kRecBytes + kt % KS;
                        const unsigned sb = __builtin_amdgcn_raw_buffer_load_b8(s_rsrc, tile ? voff_y : voff, (kt / KS) * 64 * kRecBytes, kAuxDefault);
                        sc[q] = __builtin_bit_cast(float, (sb & 0xffu) << 23);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        unsigned p[2];
                        merge_tiles(x[j], y[j], p[0], p[1]);
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const bf16x8 wb = unpack_mx(Bf16{}, p[q], sc[q]);
#pragma unroll
                            for (int mb = 0; mb < MB; ++mb) {
                                u32x4 hi, lo;
                                split_f16(a_cur[mb * 32 * 16 + frag_row + ((unsigned)(8 * q + 4 * h + j) ^ frag_swz)], hi, lo);
                                acc[mb][np] = mfma32(wb, __builtin_bit_cast(bf16x8, hi), acc[mb][np]);
                                acc[mb][np] = mfma32(wb, __builtin_bit_cast(bf16x8, lo), acc[mb][np]);
                            }
                        }
                    }
                }
                if (kt + 1 < kt_end)
                    __syncthreads();
                rel = rel + 1 == (unsigned)NBUF ? 0u : rel + 1;
            }
        }
    } else {
    // scale records, merged per pair like the weight words: rec[np][0] serves P1, rec[np][1] serves P2
    ScaleRec<FMT, KS> rec[NP][2], rec_next[NP][2];
    auto load_recs = [&](ScaleRec<FMT, KS> (*dst)[2], unsigned sp) {
#pragma unroll
        for (int np = 0; np < NP; ++np) {
            const ScaleRec<FMT, KS> x = load_scale_rec<FMT, KS>(s_rsrc, s_voff[2 * np], sp * 64 * kRecBytes);
            const ScaleRec<FMT, KS> y = load_scale_rec<FMT, KS>(s_rsrc, s_voff[2 * np + 1], sp * 64 * kRecBytes);
#pragma unroll
            for (int d = 0; d < kRecDw; ++d)
                merge_tiles(x.d[d], y.d[d], dst[np][0].d[d], dst[np][1].d[d]);
        }
    };
    load_recs(rec, sp_begin);
    u32x4 wring[D][2 * NP];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int nt = 0; nt < 2 * NP; ++nt)
            wring[i][nt] = buf_load16(w_rsrc, w_voff[nt], (kt_begin + i) * kTileBytes, kAuxDefault);

    if constexpr (Cfg::GA) {
    // ---- group-ahead form (WideCfg GA = 1): see the note at WideCfg
    Frag wfg[2][NP];
    unsigned pw2[NP][4]; // P2's words, between the merge (chunk j) and their unpack (chunk 4 + j)
    auto unpack_ga = [&](auto c_c, auto slot_c, auto t_c, const ScaleRec<FMT, KS> (*rc)[2]) {
        constexpr int C = decltype(c_c)::value, SLOT = decltype(slot_c)::value, TS = decltype(t_c)::value, j = C % 4, q = C / 4;
#pragma unroll
        for (int np = 0; np < NP; ++np) {
            unsigned w;
            if constexpr (q == 0) {
                unsigned x = wring[SLOT][2 * np][j], y = wring[SLOT][2 * np + 1][j];
                pin_here(x), pin_here(y);
                merge_tiles(x, y, w, pw2[np][j]);
            } else {
                w = pw2[np][j];
                pin_here(w);
            }
            float s_lo, s_hi;
            tile_scales<FMT, KS, TS>(rc[np][q], s_lo, s_hi);
            Frag f;
            if constexpr (PETIT_ABLATE & 4)
                f = __builtin_bit_cast(Frag, u32x4{w, w ^ __builtin_bit_cast(unsigned, s_lo), w, __builtin_bit_cast(unsigned, s_hi)});
            else if constexpr (FMT == kFmtNv)
                f = unpack_nv(AT{}, w, j < 2 ? s_lo : s_hi);
            else
                f = unpack_mx(AT{}, w, s_lo);
            u32x4 fb = __builtin_bit_cast(u32x4, f);
            pin_here(fb);
            wfg[C & 1][np] = __builtin_bit_cast(Frag, fb);
        }
    };
    unpack_ga(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, rec);
    __syncthreads();
    auto span_body = [&](const unsigned sp, auto last_c) {
        constexpr bool kLast = decltype(last_c)::value;
        const unsigned kt0 = sp * KS;
        if constexpr (!kLast)
            load_recs(rec_next, sp + 1);
        static_for<0, KS>([&](auto t_c) {
            constexpr int T = decltype(t_c)::value;
            constexpr int SLOT = T % D, NSLOT = (T + 1) % D;
            constexpr bool kNext = !kLast || (T + 1 < KS);
            constexpr bool kRefill = !kLast || (T + D < KS);
            constexpr int TN = (T + 1) % KS;
            const unsigned kt = kt0 + T;
            const u32x4 *const a_cur = smem + (T & 1) * Cfg::kBufU4;
            u32x4 *const a_pf = smem + ((T + 1) & 1) * Cfg::kBufU4;
            if constexpr (kNext && !(PETIT_ABLATE & 1))
                dma_a_tile(a_pf, kt + 1); // everybody left that buffer at the barrier that ended the previous step
            u32x4 fr[2][MB];
            read_frags(a_cur, 0, fr[0]);
            if constexpr (PETIT_ABLATE & 8)
                read_frags(a_cur, 1, fr[1]);
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, 8>([&](auto g_c) {
                constexpr int g = decltype(g_c)::value;
                if constexpr (g + 1 < 8 && !(PETIT_ABLATE & 8))
                    read_frags(a_cur, g + 1, fr[(g + 1) & 1]);
                // the group's MB * NP MFMAs on MB * NP different accumulators; chunk g + 1 (or chunk 0 of the next step) is unpacked in their shadow
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int np = 0; np < NP; ++np)
                        acc[mb][np] = mfma32(wfg[g & 1][np], __builtin_bit_cast(Frag, fr[g & 1][mb]), acc[mb][np]);
                if constexpr (g + 1 < 8) {
                    unpack_ga(std::integral_constant<int, g + 1>{}, std::integral_constant<int, SLOT>{}, t_c, rec);
                } else if constexpr (kNext) {
                    if constexpr (TN == 0)
                        unpack_ga(std::integral_constant<int, 0>{}, std::integral_constant<int, NSLOT>{}, std::integral_constant<int, 0>{}, rec_next);
                    else
                        unpack_ga(std::integral_constant<int, 0>{}, std::integral_constant<int, NSLOT>{}, std::integral_constant<int, TN>{}, rec);
                }
                if constexpr (g == 2 && kRefill && !(PETIT_ABLATE & 2)) { // chunk 3 has merged the slot's last word: tile t + D may land in it
#pragma unroll
                    for (int nt = 0; nt < 2 * NP; ++nt)
                        wring[SLOT][nt] = buf_load16(w_rsrc, w_voff[nt], (kt + D) * kTileBytes, kAuxDefault);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            if constexpr (kNext && !(PETIT_ABLATE & 32))
                __syncthreads();
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
    } else {
    // Unpacked weight fragments, double buffered: the MFMAs of step t read wf[t % 2] while the words of step t + 1 are
    // merged and unpacked into wf[(t + 1) % 2] BETWEEN them, one chunk (one word of P1 or P2 per pair: 12 VALU for NVFP4)
    // per half-block, so the unpack hides in the shadow of the matrix pipe even with a single wave on the SIMD.
    Frag wf[2][NP][2][4];
    unsigned pw2[NP][4]; // P2's words, between the merge (chunk j) and their unpack (chunk 4 + j)
    // chunk c of a step = the group it feeds a step later: q = c / 4, j = c % 4; the q = 0 chunk merges word j of the pair
    // (P2's half waits in pw2[j]) and unpacks P1's word, the q = 1 chunk unpacks P2's
    auto unpack_chunk = [&](auto c_c, auto slot_c, auto buf_c, auto t_c, const ScaleRec<FMT, KS> (*rc)[2]) {
        constexpr int C = decltype(c_c)::value, SLOT = decltype(slot_c)::value, BUF = decltype(buf_c)::value;
        constexpr int TS = decltype(t_c)::value, j = C % 4, q = C / 4;
#pragma unroll
        for (int np = 0; np < NP; ++np) {
            // pin_here: the chunk's inputs become opaque at this program point, so its (pure, otherwise freely movable)
            // VALU work cannot be hoisted into one burst at the top of the step -- hipcc did exactly that in one variant
            unsigned w;
            if constexpr (q == 0) {
                unsigned x = wring[SLOT][2 * np][j], y = wring[SLOT][2 * np + 1][j];
                pin_here(x), pin_here(y);
                merge_tiles(x, y, w, pw2[np][j]);
            } else {
                w = pw2[np][j];
                pin_here(w);
            }
            float s_lo, s_hi;
            tile_scales<FMT, KS, TS>(rc[np][q], s_lo, s_hi);
            Frag f;
            if constexpr (PETIT_ABLATE & 4)
                f = __builtin_bit_cast(Frag, u32x4{w, w ^ __builtin_bit_cast(unsigned, s_lo), w, __builtin_bit_cast(unsigned, s_hi)});
            else if constexpr (FMT == kFmtNv)
                f = unpack_nv(AT{}, w, j < 2 ? s_lo : s_hi);
            else
                f = unpack_mx(AT{}, w, s_lo);
            u32x4 fb = __builtin_bit_cast(u32x4, f);
            pin_here(fb); // ... nor sunk to its first use, the next step's MFMAs (hipcc did that too)
            wf[BUF][np][q][j] = __builtin_bit_cast(Frag, fb);
        }
    };
    // step 0 of the slice: unpacked up front (nothing to hide behind yet), its ring slot refilled
    static_for<0, 8>([&](auto c_c) {
        unpack_chunk(c_c, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, rec);
    });
    if (kt_begin + D < kt_end) {
#pragma unroll
        for (int nt = 0; nt < 2 * NP; ++nt)
            wring[0][nt] = buf_load16(w_rsrc, w_voff[nt], (kt_begin + D) * kTileBytes, kAuxDefault);
    }
    __syncthreads();

    unsigned rel3 = 0; // PF == 2: (k-tile index relative to the slice) % 3, wave-uniform
    auto span_body = [&](const unsigned sp, auto last_c) {
        constexpr bool kLast = decltype(last_c)::value;
        const unsigned kt0 = sp * KS;
        if constexpr (!kLast)
            load_recs(rec_next, sp + 1);
        static_for<0, KS>([&](auto t_c) {
            constexpr int T = decltype(t_c)::value;
            constexpr int CUR = T % 2, NXT = (T + 1) % 2;     // (KS is even: the parity carries across spans)
            constexpr int NSLOT = (T + 1) % D;                 // ring slot of step t + 1
            constexpr bool kNext = !kLast || (T + 1 < KS);     // is there a step t + 1 in this K slice
            constexpr bool kRefill = !kLast || (T + 1 + D < KS);
            constexpr int TN = (T + 1) % KS;                   // its tile index inside ITS span (0: the next span's record)
            const unsigned kt = kt0 + T;
            // relative tile index: buffers rotate with period NBUF (2: parity of T, KS being even; 3: a running counter)
            unsigned cur_buf, pf_buf;
            if constexpr (NBUF == 2) {
                cur_buf = T & 1, pf_buf = (T + 1) & 1;
            } else {
                cur_buf = rel3, pf_buf = rel3 == 0 ? 2u : rel3 - 1; // (rel + 2) % 3
            }
            const u32x4 *const a_cur = smem + cur_buf * Cfg::kBufU4;
            u32x4 *const a_pf = smem + pf_buf * Cfg::kBufU4;
            // the tile PF steps ahead: everybody left that buffer at the barrier that ended the previous step
            if constexpr (PETIT_ABLATE & 1) {
            } else if constexpr (PF == 1) {
                if constexpr (kNext)
                    dma_a_tile(a_pf, kt + 1);
            } else {
                // always issued (beyond K the loads are out of the descriptor's range: zeros into a buffer nobody reads), so the
                // wait at the end of the step is one constant and no branch splits the MFMA stream
                dma_a_tile(a_pf, kt + 2);
            }
            u32x4 fr[2][MB];
            read_frags(a_cur, 0, fr[0]);
            __builtin_amdgcn_sched_barrier(0);
            // 8 groups per step, group g = (k-half set q, word j): MB * NP MFMAs on MB * NP DIFFERENT accumulators, so no MFMA
            // waits for the one before it (an instruction placed between two MFMAs on the SAME accumulator costs ~43 cycles:
            // with the m-block outermost, 4 dependent MFMAs and the interleaved unpack ran at half the matrix rate).  While
            // they run: the fragments of group g + 1 are read and chunk g of step t + 1 is unpacked.
            static_for<0, 8>([&](auto g_c) {
                constexpr int g = decltype(g_c)::value;
                if constexpr (g + 1 < 8 && !(PETIT_ABLATE & 8))
                    read_frags(a_cur, g + 1, fr[(g + 1) & 1]);
                constexpr int q = g >> 2, j = g & 3;
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int np = 0; np < NP; ++np)
                        if constexpr (PETIT_ABLATE & 16) {
                            const u32x4 wb = __builtin_bit_cast(u32x4, wf[CUR][np][q][j]), ab = fr[(PETIT_ABLATE & 8) ? 0 : (g & 1)][mb];
                            acc[mb][np][g] += __builtin_bit_cast(float, wb[0] ^ ab[1]);
                        } else {
                            acc[mb][np] = mfma32(wf[CUR][np][q][j], __builtin_bit_cast(Frag, fr[(PETIT_ABLATE & 8) ? 0 : (g & 1)][mb]), acc[mb][np]);
                        }
                if constexpr (kNext) {
                    if constexpr (TN == 0)
                        unpack_chunk(g_c, std::integral_constant<int, NSLOT>{}, std::integral_constant<int, NXT>{},
                                     std::integral_constant<int, 0>{}, rec_next);
                    else
                        unpack_chunk(g_c, std::integral_constant<int, NSLOT>{}, std::integral_constant<int, NXT>{},
                                     std::integral_constant<int, TN>{}, rec);
                }
                __builtin_amdgcn_sched_barrier(0); // keep prefetch and unpack one group ahead, no further
            });
            if constexpr (PF == 1) {
                if constexpr (kRefill && !(PETIT_ABLATE & 2)) { // the slot of step t + 1 is free again: tile t + 1 + D
#pragma unroll
                    for (int nt = 0; nt < 2 * NP; ++nt)
                        wring[NSLOT][nt] = buf_load16(w_rsrc, w_voff[nt], (kt + 1 + D) * kTileBytes, kAuxDefault);
                }
                if constexpr (kNext && !(PETIT_ABLATE & 1))
                    __syncthreads();
            } else {
                // Tile t + 1 (requested a step ago) must have landed; the tile requested at the top of THIS step stays in
                // flight: vmcnt counts in issue order, so "at most kDmaLoads outstanding" retires everything older than it.
                // Raw barrier: __syncthreads() would make hipcc drain the LDS-DMA queue (vmcnt(0)).
                if constexpr (kNext) {
#if defined(__HIP_DEVICE_COMPILE__)
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Cfg::kDmaLoads) : "memory");
                    __builtin_amdgcn_s_barrier();
#endif
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (kRefill) { // after the wait, so the count above stays exact
#pragma unroll
                    for (int nt = 0; nt < 2 * NP; ++nt)
                        wring[NSLOT][nt] = buf_load16(w_rsrc, w_voff[nt], (kt + 1 + D) * kTileBytes, kAuxDefault);
                }
                rel3 = rel3 == 2 ? 0u : rel3 + 1;
            }
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
    } // (step-ahead form)
    } // (fast body)
    if constexpr (AT::kAdaptive) { // the join of the two bodies: see mfma_join_settle (device_common.hpp)
        static_for<0, 2>([&](auto pass) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int np = 0; np < NP; ++np)
                    mfma_join_pin(acc[mb][np]);
            if constexpr (decltype(pass)::value == 0)
                mfma_join_settle();
        });
    }

    if constexpr (KG == 2) {
        // a group with fewer spans than its partner keeps the partner's barrier count (a span is KS barriers: the prologue's stands in for
        // the one the last step does not execute), then group 1 parks its accumulators in LDS and group 0 adds them (gemm_native32.hpp)
        for (unsigned i = sp_end - sp_begin; i < p.spans_per_wave; ++i)
            for (int b = 0; b < KS; ++b)
                __builtin_amdgcn_s_barrier();
        __syncthreads();
        float *const red = reinterpret_cast<float *>(smem_all);
        if (kg == 1) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int np = 0; np < NP; ++np)
#pragma unroll
                    for (int v = 0; v < 16; ++v)
                        red[(((wave * MB + mb) * NP + np) * 16 + v) * 64 + lane] = acc[mb][np][v];
        }
        __syncthreads();
        if (kg == 1)
            return;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int np = 0; np < NP; ++np)
#pragma unroll
                for (int v = 0; v < 16; ++v)
                    acc[mb][np][v] += red[(((wave * MB + mb) * NP + np) * 16 + v) * 64 + lane];
    }

    // --- epilogue.  32x32 accumulator: lane (m = l%32, h = l/32) holds, for v = 4u + e, row n = 8u + 4h + e of the
    // pair's 32 weight rows: four groups of 4 consecutive n (rows 0-15 = first tile of the pair, 16-31 = second).
    const unsigned m_base = m0 + m_l;
    if (gridDim.z > 1) { // K split across workgroups: fp32 partial tile -> this slice's slab
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
    if (p.act) { // SiLU-mul: the pair (2 np, 2 np + 1) is the gate / up halves of output tile (nt0 + 2 np) / 2
        const unsigned n_half = p.n >> 1;
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
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int np = 0; np < NP; ++np)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned m = m_base + mb * 32;
                const unsigned nt = 2 * np + (u >> 1);
                const unsigned n = (nt0 + nt) * 16 + (u & 1) * 8 + 4 * h;
                if (m < p.m && nt < valid_nt) {
                    const f32x4 v = f32x4{acc[mb][np][4 * u], acc[mb][np][4 * u + 1], acc[mb][np][4 * u + 2], acc[mb][np][4 * u + 3]};
                    *reinterpret_cast<uint2 *>((char *)p.c + ((size_t)m * p.n + n) * 2) = finish4<AT>(v, gs, p.bias, n);
                }
            }
}

} // namespace petit_amd
