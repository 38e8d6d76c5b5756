// gemm_tiled.hpp -- the FP4 GEMM for large M (prefill regime, MFMA-bound).
//
// Same contract, layout and unpack as gemm_stream.hpp; what changes is the reuse
// structure.  At M >= ~64 every dequantised weight fragment must feed many MFMAs, and
// every activation fragment must be shared by many weight tiles:
//  * a workgroup owns a BM x BN tile of C (BM = 16*MT rows, BN = 16*NTW*WAVES columns)
//    and walks K in steps of one 128-k weight tile;
//  * the A tile of a step (BM x 128) is staged in LDS once per workgroup, double
//    buffered, one barrier per step (the reference's 2-stage LDS pipeline,
//    gemm_fp4_fp16_grid.cuh:323-431, needs 2-4 barriers per step because W goes
//    through LDS as well);
//  * W still never touches LDS: each wave streams the tiles of its own NTW n-tiles
//    straight into VGPRs (ring of D), unpacks each word ONCE (hardware convert) and
//    reuses the fragment across all MT m-tiles: 4*MT*NTW MFMAs per step against
//    NTW*4 unpacks, so the VALU work hides in the MFMA shadow;
//  * C^T = W . A^T as in the streaming kernel: a lane ends with 4 consecutive n of one
//    m, 8-byte stores.
// One workgroup per CU at 1 wave per SIMD is the intended residency (the kernel uses
// ~200 VGPRs on purpose: accumulators for a 128 x 32 slab per wave).
#pragma once

#include "gemm_stream.hpp"

namespace petit_amd {

//   MT   m-tiles (of 16) per workgroup = per wave          (BM = 16*MT)
//   NTW  n-tiles (of 16) per wave                          (BN = 16*NTW*WAVES)
//   WAVES waves per workgroup (along N)
//   D    W ring depth in k-tiles
// gridDim.z > 1 splits K across workgroups (contiguous spans per slice, spans_per_wave of them): every slice writes its
// fp32 partial tile into its slab of p.workspace and splitk_reduce_kernel sums the slabs in a fixed order (deterministic;
// the reference has no K split at all, gemm_fp4_fp16_grid.cuh:554-555, and leaves half the chip idle at M = 512, N = 8192).
template <class AT_, int FMT_, int KS_, int MT_, int NTW_, int WAVES_, int D_> struct TiledCfg {
    using AT = AT_;
    static constexpr int FMT = FMT_, KS = KS_, MT = MT_, NTW = NTW_, WAVES = WAVES_, D = D_;
    static constexpr int kThreads = 64 * WAVES;
    static constexpr int BM = 16 * MT;
    // A tile in LDS.  Plain 16-bit activations go global -> LDS directly (buffer_load ... lds: no VGPR staging, no
    // ds_write; probed in tools/probes/lds_dma_probe.hip: lane i of a wave-load lands at base + 16 i, out-of-range
    // lanes write zeros).  That forces lane-linear placement, so rows are 16 units with an XOR swizzle instead of
    // a pad: unit u of row r sits at position u ^ (r % 16), and the 16 rows a fragment read touches hit 16
    // different bank groups.
    static constexpr int kRowU4 = 16;                     // 16 units of 16 B
    static constexpr int kBufU4 = BM * kRowU4;            // one A tile
    static constexpr int kUnitsPerThread = BM * 16 / kThreads;
    // 64 accumulator registers + direct-to-LDS staging fit two waves per SIMD; hipcc lands on 260 VGPRs unless told
    static constexpr int kMinWavesPerSimd = (MT == 8 && NTW == 2) ? 2 : 1;
    static_assert(KS % D == 0, "ring depth must divide the span");
    static_assert((BM * 16) % kThreads == 0, "A tile must split evenly over the workgroup");
    static_assert(2 * kBufU4 * 16 <= 160 * 1024, "LDS budget");
    static_assert(!AT::kBfp, "plain bf16 / fp16 activations (or Fp16Mx: the fast / fallback pair of device_common.hpp)");
    static_assert(!AT::kAdaptive || FMT == kFmtMx, "Fp16Mx: fp16 activations x MXFP4 weights");
};

template <class Cfg>
__global__ __launch_bounds__(Cfg::kThreads, Cfg::kMinWavesPerSimd) void gemm_tiled_kernel(const GemmArgs p) {
    using AT = typename Cfg::AT;
    using Frag = typename AT::frag;
    constexpr int FMT = Cfg::FMT, KS = Cfg::KS, MT = Cfg::MT, NTW = Cfg::NTW, WAVES = Cfg::WAVES, D = Cfg::D;
    constexpr unsigned kRecBytes = ScaleRec<FMT, KS>::kBytes;
    constexpr unsigned kOob = 0x80000000u;

    __shared__ u32x4 smem[2 * Cfg::kBufU4];

    const unsigned tid = threadIdx.x;
    const unsigned lane = tid & 63u;
    const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned r = lane & 15u, g = lane >> 4;

    const unsigned ktiles = p.k / kTileK;
    const unsigned nspans = ktiles / KS;
    const unsigned ntiles = p.n / kTileN;
    unsigned bn, bm;
    tile_of_block(p.flags, bn, bm);
    stagger_priority(p.flags);
    const unsigned nt0 = (bn * WAVES + wave) * NTW;
    const unsigned m0 = bm * Cfg::BM;
    // K slice of this workgroup (whole spans; the host guarantees every slice is non-empty)
    const unsigned sp_begin = min(blockIdx.z * p.spans_per_wave, nspans - 1);
    const unsigned sp_end = min(sp_begin + p.spans_per_wave, nspans);
    const unsigned kt_begin = sp_begin * KS;

    f32x4 acc[MT][NTW];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
            acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const unsigned valid_nt = nt0 < ntiles ? min((unsigned)NTW, ntiles - nt0) : 0u;
    const unsigned w_row_bytes = ktiles * kTileBytes;
    const unsigned s_row_bytes = (FMT == kFmtNv) ? p.k : p.k / 2;
    const unsigned rows = min(p.m - m0, (unsigned)Cfg::BM);

    // a wave whose n-tiles all fall beyond N still helps staging A and hits the barriers
    // logical -> physical n-tiles (identity, or gate/up pairs for the SiLU-mul epilogue; device_common.hpp)
    const unsigned pt0 = valid_nt ? physical_tile(nt0, ntiles, p.act) : 0u;
    const unsigned span_tiles = !valid_nt ? 0u : p.act ? (valid_nt >> 1) + (ntiles >> 1) : valid_nt;
    const __amdgpu_buffer_rsrc_t w_rsrc =
        make_rsrc((const char *)p.w + (size_t)pt0 * w_row_bytes, span_tiles * w_row_bytes);
    const __amdgpu_buffer_rsrc_t s_rsrc =
        make_rsrc((const char *)p.s + (size_t)pt0 * s_row_bytes, span_tiles * s_row_bytes);
    const __amdgpu_buffer_rsrc_t a_rsrc = make_rsrc((const char *)p.a + (size_t)m0 * p.k * 2, rows * p.k * 2);

    unsigned w_voff[NTW], s_voff[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const unsigned rel = physical_tile(nt0 + nt, ntiles, p.act) - pt0;
        w_voff[nt] = ((unsigned)nt < valid_nt) ? lane * 16 + rel * w_row_bytes : kOob;
        s_voff[nt] = ((unsigned)nt < valid_nt) ? lane * kRecBytes + rel * s_row_bytes : kOob;
    }
    // direct-to-LDS staging: wave-load i of this wave covers rows 4*(i*WAVES + wave) .. +3; lane l -> row + l/16,
    // position l%16, which receives unit (l%16) ^ (row%16) of that row
    constexpr int kDmaLoads = Cfg::BM * 16 / 64 / WAVES; // wave-loads per wave per tile
    // one VGPR: wave-load i differs from wave-load 0 only by 4*WAVES whole rows, a multiple of 16, so the swizzle
    // term is the same and the row step rides in the SGPR offset
    static_assert((4 * WAVES) % 16 == 0, "direct-to-LDS staging: wave-loads must step by whole 16-row groups");
    const unsigned dma_row0 = wave * 4 + (lane >> 4);
    const unsigned dma_voff = dma_row0 * p.k * 2 + (((lane & 15u) ^ (dma_row0 & 15u)) * 16); // rows >= M: out of range -> zeros
    auto dma_a_tile = [&](u32x4 *dst, unsigned kt) {
#pragma unroll
        for (int i = 0; i < kDmaLoads; ++i) {
#if defined(__HIP_DEVICE_COMPILE__) // (the host pass knows neither the builtin nor the LDS address space)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (__attribute__((address_space(3))) void *)(dst + (i * WAVES + wave) * 64),
                                                     16, dma_voff, i * (4 * WAVES) * p.k * 2 + kt * 256, 0, 0);
#else
            (void)dst, (void)kt;
#endif
        }
    };

    // --- prologue
    dma_a_tile(smem, kt_begin); // (kt_begin is even: KS is, so the first tile lands in buffer 0)
    ScaleRec<FMT, KS> srec[NTW], srec_next[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
        srec[nt] = load_scale_rec<FMT, KS>(s_rsrc, s_voff[nt], sp_begin * 64 * kRecBytes);
    u32x4 wring[D][NTW];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
            wring[i][nt] = buf_load16(w_rsrc, w_voff[nt], (kt_begin + i) * kTileBytes, kAuxDefault);
    __syncthreads();

    // Fp16Mx: is any scale byte of the span now in srec outside 114..140?  Wave-uniform; the first such span switches THIS WAVE to the
    // fallback loop below for the rest of its K slice
    auto mx_span_needs_fallback = [&]() -> bool {
        if constexpr (AT::kAdaptive) {
            unsigned bad = 0;
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
                if ((unsigned)nt < valid_nt)
                    bad = mx_rec_outside_f16<KS>(srec[nt], bad);
            return __builtin_amdgcn_ballot_w64(bad != 0) != 0;
        } else {
            return false;
        }
    };
    bool mx_fb = mx_span_needs_fallback();
    auto span_body = [&](const unsigned sp, auto last_c) {
        constexpr bool kLast = decltype(last_c)::value;
        const unsigned kt0 = sp * KS;
        if constexpr (!kLast) {
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
                srec_next[nt] = load_scale_rec<FMT, KS>(s_rsrc, s_voff[nt], (sp + 1) * 64 * kRecBytes);
        }
        static_for<0, KS>([&](auto t_c) {
            constexpr int T = decltype(t_c)::value;
            constexpr int SLOT = T % D;
            constexpr bool kRefill = !kLast || (T + D < KS);
            constexpr bool kNextA = !kLast || (T + 1 < KS);
            const unsigned kt = kt0 + T;
            const u32x4 *const a_cur = smem + ((kt & 1u) ? Cfg::kBufU4 : 0);
            u32x4 *const a_nxt = smem + ((kt & 1u) ? 0 : Cfg::kBufU4);
            // next step's A tile: global -> registers now, LDS after this step's reads
            if constexpr (kNextA)
                dma_a_tile(a_nxt, kt + 1); // everybody left a_nxt at the barrier that ended the previous step
            // unpack this step's weight words once
            Frag wf[NTW][4];
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                float s_lo, s_hi;
                tile_scales<FMT, KS, T>(srec[nt], s_lo, s_hi);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned w = wring[SLOT][nt][j];
                    if constexpr (FMT == kFmtNv)
                        wf[nt][j] = unpack_nv(AT{}, w, j < 2 ? s_lo : s_hi);
                    else
                        wf[nt][j] = unpack_mx(AT{}, w, s_lo);
                }
            }
            if constexpr (kRefill) {
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
                    wring[SLOT][nt] = buf_load16(w_rsrc, w_voff[nt], (kt + D) * kTileBytes, kAuxDefault);
            }
            // the step's global loads (next A tile, W refill) are requested before its MFMAs: hipcc otherwise
            // sinks the A loads next to their LDS stores at the end of the step and every step exposes the
            // whole L2 / HBM latency.  Only for the small tiles: with >= 64 accumulator registers the pinned
            // loads cost the second wave per SIMD, which hides that latency better (measured both ways:
            // 64x128 tiles +7 % at M = 256, 128x128 tiles -12 % at M = 512 / 2048).
            __builtin_amdgcn_sched_barrier(0);
            // every m-tile: 4 fragments from LDS, 4*NTW MFMAs
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                u32x4 af[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    af[j] = a_cur[(mt * 16 + r) * 16 + ((g * 4 + j) ^ r)];
#pragma unroll
                for (int j = 0; j < 4; ++j) // j outer: consecutive MFMAs hit different accumulators
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt)
                        acc[mt][nt] = mfma16(wf[nt][j], __builtin_bit_cast(Frag, af[j]), acc[mt][nt]);
            }
            if constexpr (kNextA)
                __syncthreads();
        });
        if constexpr (!kLast) {
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
                srec[nt] = srec_next[nt];
            if constexpr (AT::kAdaptive)
                mx_fb = mx_span_needs_fallback();
        }
    };
    if constexpr (AT::kAdaptive) {
        unsigned sp = sp_begin;
        for (; sp + 1 < sp_end && !mx_fb; ++sp)
            span_body(sp, std::false_type{});
        if (!mx_fb) {
            span_body(sp_end - 1, std::true_type{});
        } else {
            // Fp16Mx fallback, spans [sp, sp_end): exact for any e8m0 scale, written for size, not speed (device_common.hpp): one k-tile per
            // trip of a rolled loop, the W tiles and their scale bytes loaded on the spot, weights to bf16, every fp16 fragment split into
            // hi + lo bf16 in registers, two MFMAs per word.  It keeps the workgroup's protocol -- its share of the next activation tile's
            // DMA at the top of the step, one barrier per step -- so the waves still in the fast body never notice.
            const unsigned kt_end = sp_end * KS;
#pragma unroll 1
            for (unsigned kt = sp * KS; kt < kt_end; ++kt) {
                const u32x4 *const a_cur = smem + ((kt & 1u) ? Cfg::kBufU4 : 0);
                if (kt + 1 < kt_end)
                    dma_a_tile(smem + ((kt & 1u) ? 0 : Cfg::kBufU4), kt + 1);
                bf16x8 wb[NTW][4];
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) {
                    const u32x4 wt = buf_load16(w_rsrc, w_voff[nt], kt * kTileBytes, kAuxDefault);
                    const unsigned sb = __builtin_amdgcn_raw_buffer_load_b8(s_rsrc, s_voff[nt] + kt % KS, (kt / KS) * 64 * kRecBytes, kAuxDefault);
                    const float sc = __builtin_bit_cast(float, (sb & 0xffu) << 23);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        wb[nt][j] = unpack_mx(Bf16{}, wt[j], sc);
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        u32x4 hi, lo;
                        split_f16(a_cur[(mt * 16 + r) * 16 + ((g * 4 + j) ^ r)], hi, lo);
#pragma unroll
                        for (int nt = 0; nt < NTW; ++nt)
                            acc[mt][nt] = mfma16_hilo(wb[nt][j], hi, lo, acc[mt][nt]);
                    }
                if (kt + 1 < kt_end)
                    __syncthreads();
            }
        }
        // the join of the two bodies: see mfma_join_settle (device_common.hpp)
        static_for<0, 2>([&](auto pass) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
                    mfma_join_pin(acc[mt][nt]);
            if constexpr (decltype(pass)::value == 0)
                mfma_join_settle();
        });
    } else {
        for (unsigned sp = sp_begin; sp + 1 < sp_end; ++sp)
            span_body(sp, std::false_type{});
        span_body(sp_end - 1, std::true_type{});
    }

    if (gridDim.z > 1) { // K split across workgroups: fp32 partial tile -> this slice's slab
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const unsigned m = m0 + mt * 16 + r;
                const unsigned n = (nt0 + nt) * 16 + g * 4;
                if (m < p.m && (unsigned)nt < valid_nt)
                    *reinterpret_cast<f32x4 *>(p.workspace + ((size_t)blockIdx.z * p.m + m) * p.n + n) = acc[mt][nt];
            }
        return;
    }
    // --- epilogue: x global scale, one RNE rounding, 8-byte stores
    const float gs = *p.gs;
    if (p.act) { // SiLU-mul: tiles (nt, nt + 1) are the gate / up halves of output tile (nt0 + nt) / 2
        if constexpr (NTW % 2 == 0) {
            const unsigned n_half = p.n >> 1;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTW; nt += 2) {
                    const unsigned m = m0 + mt * 16 + r;
                    const unsigned n = ((nt0 + nt) >> 1) * 16 + g * 4;
                    if (m < p.m && (unsigned)nt < valid_nt)
                        *reinterpret_cast<uint2 *>((char *)p.c + ((size_t)m * n_half + n) * 2) =
                            finish4_silu_mul<AT>(acc[mt][nt], acc[mt][nt + 1], gs, p.bias, n, n_half);
                }
        }
        return;
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const unsigned m = m0 + mt * 16 + r;
            const unsigned n = (nt0 + nt) * 16 + g * 4;
            if (m < p.m && (unsigned)nt < valid_nt) {
                const f32x4 v = acc[mt][nt];
                *reinterpret_cast<uint2 *>((char *)p.c + ((size_t)m * p.n + n) * 2) = finish4<AT>(v, gs, p.bias, n);
            }
        }
}

} // namespace petit_amd
