// gemm_batch.hpp -- weight streaming for 17 <= M <= 128 ("batched decode": the everyday batch sizes of a serving engine).
//
// The regime between the decode kernels (M <= 16: one m-tile, HBM-bound) and the large-M tiles (M >= 128: MFMA-bound).  The weights
// are still the dominant HBM stream -- at M = 64 on 8192 x 8192 the arithmetic intensity is 228 FLOP/B against a ridge of 315 -- but
// two more budgets bind before HBM does (profiles/r05_midm_pmc.md):
//   * what a CU must pull in: its share of W plus EVERY activation row of its K range, once per workgroup: a workgroup that walks
//     the whole K at M = 64 ingests 1 MiB of activations for 128 KiB of weights through a ~64 B/clk vector-memory path;
//   * the second launch of a cross-workgroup K split: splitk * M * N * 4 bytes of fp32 slabs written and read back -- at M = 64,
//     split 8, half as many bytes as the weights themselves (the tiled kernels' answer to the first budget: 13 us GEMM + 5 us reduce).
// This kernel is gemm_mid.hpp's organisation (5 <= M <= 16) carried to MT m-tiles: a workgroup is WN x WK waves; K is split over the
// WK parts INSIDE the workgroup (partial sums meet once, in LDS, after the loop: no slabs), the WN waves of a part share one
// activation tile per k-tile in LDS (BM = 16 MT rows x 256 B, global -> LDS by buffer_load ... lds, D tiles ahead in a ring of D + 1
// slots, counted vmcnt + bare s_barrier per step), W never touches LDS (ring of D tiles in VGPRs per n-tile), every unpacked weight
// word feeds MT MFMAs.  8-16 waves per CU walk disjoint K ranges and column blocks, so unpack VALU, MFMA and the LDS fragment reads of
// different waves overlap without any intra-wave software pipelining.  An optional split across workgroups (gridDim.z, fp32 slabs +
// the fixed-order reduce pass) remains for shapes whose N alone cannot fill the chip.
// Reference counterpart: fp4/algo_chooser.cc:89-104 (its `m <= 64` branch: the same 16x32 / 32x32 tiles as decode, K walked by four
// warps, W through LDS) is the reference's whole answer to this regime.
#pragma once

#include "device_common.hpp"

// Ablation builds (tools/ablate_batch.sh; 0 in the shipped library): 1 no activation-tile DMA after the prologue, 2 no unpack VALU (the packed words
// stand in for the fragments), 4 no MFMA, 8 no LDS fragment reads (one fragment per step stands in), 16 no weight refills.  Results are garbage; only
// time counts.
#ifndef PETIT_ABLATE_BATCH
#define PETIT_ABLATE_BATCH 0
#endif

namespace petit_amd {

//   AT    Bf16 / Fp16 activations (and output)
//   FMT   kFmtNv / kFmtMx
//   KS    tiles per span
//   MT    m-tiles (of 16 rows) per workgroup: BM = 16 MT
//   NT    n-tiles per wave
//   WN,WK waves along N / K in the workgroup (the WN waves of a K part share its activation tiles)
//   D     ring depth in k-tiles (W in VGPRs, A in LDS slots), divides KS
//   DA    0: every wave loads its slice of the activation tiles together with its weights (one load queue per wave: tile t + D is requested with
//            the weights of t + D, ring of D + 1 LDS slots -- gemm_mid.hpp's protocol).  A wave's loads retire in order, so the wait for the NEXT
//            activation tile also waits for every weight tile requested before it: the weight stream is never more than D steps deep, and D is
//            bounded by the LDS the activation ring takes.
//         > 0: a LOADER wave per K part requests the activation tiles DA steps ahead (ring of DA + 1 slots) and does nothing else; the compute
//            waves request weights only, D tiles deep per n-tile (hipcc tracks their vmcnt: no manual waits), and meet the loader at the step's
//            barrier.  Weight bytes in flight per CU: WN WK NT D KiB, independent of the activation ring.
template <class AT_, int FMT_, int KS_, int MT_, int NT_, int WN_, int WK_, int D_, int DA_ = 0> struct BatchCfg {
    using AT = AT_;
    static constexpr int FMT = FMT_, KS = KS_, MT = MT_, NT = NT_, WN = WN_, WK = WK_, D = D_, DA = DA_;
    static constexpr int kComputeWaves = WN * WK;
    static constexpr int kThreads = 64 * (WN + (DA > 0 ? 1 : 0)) * WK;  // loader wave of part wk: wave kComputeWaves + wk
    static constexpr int BM = 16 * MT;
    static constexpr int kTileU4 = BM * 16;               // one activation tile: BM rows x 16 units of 16 B
    static constexpr int kSlots = (DA > 0 ? DA : D) + 1;
    static constexpr int kPartU4 = kSlots * kTileU4;      // per K part
    static constexpr int kDma = DA > 0 ? BM * 16 / 64 : BM * 16 / 64 / WN;   // KiB wave-loads per (loading) wave per tile
    static constexpr int kRedU4 = WN * NT * MT * 64;      // float4 partial outputs per K part
    // the reduction scratch reuses the activation ring (every wave has left the loop before the first partial sum is parked)
    static constexpr int kSmemU4 = WK * kPartU4 > WK * kRedU4 ? WK * kPartU4 : WK * kRedU4;
    static_assert(MT >= 1 && MT <= 8 && (MT >= 2 || DA > 0), "1..8 m-tiles (MT = 1 without a loader wave is gemm_mid.hpp)");
    static_assert(DA > 0 || (kDma >= 1 && kDma * WN * 4 == BM), "the waves of a K part split a tile into whole KiB loads");
    static_assert(DA == 0 || (DA <= 4 && (DA > 1 ? (DA - 1) * kDma : 0) <= 63), "the loader's counted vmcnt is a 6-bit field");
    static_assert(DA <= KS, "the loader's prologue requests DA tiles of the part's K range: a part is at least one span of KS tiles (ADVICE r05)");
    static_assert(KS % D == 0, "ring depth must divide the span");
    static_assert(!AT::kBfp, "plain bf16 / fp16 activations, or Fp16Mx (fast body + exact fallback, device_common.hpp) in the loader-wave form");
    static_assert(!AT::kAdaptive || (FMT == kFmtMx && DA > 0), "Fp16Mx: fp16 activations x MXFP4 weights, loader-wave form only");
    static_assert(kThreads <= 1024 && kSmemU4 * 16 <= 160 * 1024, "workgroup / LDS budget");
};

template <class Cfg>
__global__ __launch_bounds__(Cfg::kThreads) void gemm_batch_kernel(const void *arg_w, const void *arg_s, const void *arg_a, unsigned arg_k,
                                                                   unsigned arg_n, unsigned arg_m, unsigned arg_spw, unsigned arg_act,
                                                                   void *arg_c, const float *arg_gs, const void *arg_bias,
                                                                   float *arg_workspace) {
    using AT = typename Cfg::AT;
    using Frag = typename AT::frag;
    constexpr int FMT = Cfg::FMT, KS = Cfg::KS, MT = Cfg::MT, NT = Cfg::NT, WN = Cfg::WN, WK = Cfg::WK, D = Cfg::D, DA = Cfg::DA;
    constexpr unsigned kRecBytes = ScaleRec<FMT, KS>::kBytes;
    constexpr unsigned kOob = 0x80000000u;
    // loads a wave issues per step: its slice of the A tile + one W tile per n-tile
    constexpr int kLoadsPerStep = Cfg::kDma + NT;

    __shared__ u32x4 smem[Cfg::kSmemU4];

    const unsigned lane = threadIdx.x & 63u;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool loader = DA > 0 && wave >= (unsigned)Cfg::kComputeWaves;                       // (wave-uniform)
    const unsigned wn = loader ? 0u : wave % WN, wk = loader ? wave - Cfg::kComputeWaves : wave / WN;
    const unsigned r = lane & 15u, g = lane >> 4;

    const unsigned ktiles = arg_k / kTileK;
    const unsigned nspans = ktiles / KS;
    const unsigned ntiles = arg_n / kTileN;
    const unsigned nt0 = (blockIdx.x * WN + wn) * NT;
    const unsigned m0 = blockIdx.y * Cfg::BM;
    // K range of this part: gridDim.z splits K across workgroups first, WK across the parts of a workgroup second.  Every wave walks arg_spw
    // spans' worth of barriers (the count must agree across the workgroup); a part whose range ends early idles through the rest.
    const unsigned part = blockIdx.z * WK + wk;
    const unsigned sp_begin = min(part * arg_spw, nspans);
    const unsigned sp_end = min(sp_begin + arg_spw, nspans);

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    u32x4 *const a_part = smem + wk * Cfg::kPartU4;
    const unsigned valid_nt = (nt0 < ntiles && !loader) ? min((unsigned)NT, ntiles - nt0) : 0u;
    const unsigned w_row_bytes = ktiles * kTileBytes;
    const unsigned s_row_bytes = (FMT == kFmtNv) ? arg_k : arg_k / 2;
    const unsigned rows = min(arg_m - m0, (unsigned)Cfg::BM);
    const unsigned pt0 = valid_nt ? physical_tile(nt0, ntiles, arg_act) : 0u;
    const unsigned span_tiles = !valid_nt ? 0u : arg_act ? (valid_nt >> 1) + (ntiles >> 1) : valid_nt;
    const __amdgpu_buffer_rsrc_t w_rsrc = make_rsrc((const char *)arg_w + (size_t)pt0 * w_row_bytes, span_tiles * w_row_bytes);
    const __amdgpu_buffer_rsrc_t s_rsrc = make_rsrc((const char *)arg_s + (size_t)pt0 * s_row_bytes, span_tiles * s_row_bytes);
    const __amdgpu_buffer_rsrc_t a_rsrc = make_rsrc((const char *)arg_a + (size_t)m0 * arg_k * 2, rows * arg_k * 2);

    unsigned w_voff[NT], s_voff[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const unsigned rel = valid_nt ? physical_tile(nt0 + nt, ntiles, arg_act) - pt0 : 0u;
        w_voff[nt] = ((unsigned)nt < valid_nt) ? lane * 16 + rel * w_row_bytes : kOob;
        s_voff[nt] = ((unsigned)nt < valid_nt) ? lane * kRecBytes + rel * s_row_bytes : kOob;
    }
    // A tile slice of this wave: wave-load i covers rows 4 (i WN + wn) .. + 3; lane l -> row + l / 16, position l % 16, which receives
    // unit (l % 16) ^ (row % 16) of that row.  Rows >= M fall out of the descriptor: zeros.
    // (DA > 0: the part's loader wave takes the whole tile: wave-load i covers rows 4 i .. 4 i + 3)
    constexpr unsigned kDmaStride = DA > 0 ? 1u : (unsigned)WN;
    unsigned dma_voff[Cfg::kDma];
#pragma unroll
    for (int i = 0; i < Cfg::kDma; ++i) {
        const unsigned row = 4 * (i * kDmaStride + wn) + (lane >> 4);
        dma_voff[i] = row * arg_k * 2 + (((lane & 15u) ^ (row & 15u)) * 16);
    }
    auto dma_a_tile = [&](unsigned slot, unsigned kt) {
#pragma unroll
        for (int i = 0; i < Cfg::kDma; ++i) {
#if defined(__HIP_DEVICE_COMPILE__) // (the host pass knows neither the builtin nor the LDS address space)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                a_rsrc, (__attribute__((address_space(3))) void *)(a_part + slot * Cfg::kTileU4 + (i * kDmaStride + wn) * 64), 16, dma_voff[i],
                kt * 256, 0, 0);
#else
            (void)slot, (void)kt;
#endif
        }
    };
    // fragment of (m-tile mt, MFMA j): row 16 mt + r, unit (4 g + j) ^ r
    const u32x4 *fptr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        fptr[j] = a_part + (int)(r * 16 + ((g * 4 + j) ^ r));

    const bool part_on = sp_begin < sp_end;
    if constexpr (DA > 0) {
        if (loader) {
            // ---- the part's loader wave: activation tiles only, DA steps ahead.  Rolled loops: the trip count is the part's, the wait counts are
            // constants (a tile is kDma loads; tile t + 1 is complete when at most the DA - 1 tiles requested after it are outstanding).
            if (part_on) {
                const unsigned kt_begin = sp_begin * KS, kt_end = sp_end * KS; // (>= KS >= DA tiles)
                unsigned slot = kt_begin % Cfg::kSlots;
                for (unsigned i = 0; i < (unsigned)DA; ++i)
                    dma_a_tile((slot + i) % Cfg::kSlots, kt_begin + i);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DA - 1) * Cfg::kDma) : "memory");
                __builtin_amdgcn_s_barrier();
                unsigned kt = kt_begin;
                for (; kt + DA < kt_end; ++kt) { // step kt: request tile kt + DA into the slot step kt - 1 read; tile kt + 1 must land
                    if constexpr (!(PETIT_ABLATE_BATCH & 1))
                        dma_a_tile(slot == 0 ? (unsigned)DA : slot - 1, kt + DA);
                    slot = slot + 1 == (unsigned)Cfg::kSlots ? 0u : slot + 1;
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DA - 1) * Cfg::kDma) : "memory");
                    __builtin_amdgcn_s_barrier();
                }
                for (; kt < kt_end; ++kt) { // the last DA steps request nothing: drain (nothing may be in flight when the ring's memory is reused)
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                }
            }
        } else if (part_on) {
            // ---- a compute wave: weights only in its load queue, D tiles per n-tile in flight
            const unsigned kt_begin = sp_begin * KS;
            ScaleRec<FMT, KS> srec[NT], srec_next[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                srec[nt] = load_scale_rec<FMT, KS>(s_rsrc, s_voff[nt], sp_begin * 64 * kRecBytes);
            u32x4 wring[D][NT];
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    wring[i][nt] = buf_load16(w_rsrc, w_voff[nt], (kt_begin + i) * kTileBytes, kAuxNt);
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier(); // the loader's first tile is in its slot
            asm volatile("" ::: "memory");
            unsigned aslot = kt_begin % Cfg::kSlots;
            auto span_body = [&](const unsigned sp, auto last_c) {
                constexpr bool kLast = decltype(last_c)::value;
                const unsigned kt0 = sp * KS;
                if constexpr (!kLast) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        srec_next[nt] = load_scale_rec<FMT, KS>(s_rsrc, s_voff[nt], (sp + 1) * 64 * kRecBytes);
                }
                static_for<0, KS>([&](auto t_c) {
                    constexpr int T = decltype(t_c)::value;
                    constexpr int SLOT = T % D;
                    constexpr bool kAhead = !kLast || (T + D < KS);
                    const unsigned kt = kt0 + T;
                    const unsigned cur = aslot * Cfg::kTileU4;
                    aslot = aslot + 1 == (unsigned)Cfg::kSlots ? 0u : aslot + 1;
                    __builtin_amdgcn_sched_barrier(0);
                    Frag wf[NT][4];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        float s_lo, s_hi;
                        tile_scales<FMT, KS, T>(srec[nt], s_lo, s_hi);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const unsigned w = wring[SLOT][nt][j];
                            if constexpr (PETIT_ABLATE_BATCH & 2)
                                wf[nt][j] = __builtin_bit_cast(Frag, u32x4{w, w ^ __builtin_bit_cast(unsigned, s_lo), w, __builtin_bit_cast(unsigned, s_hi)});
                            else if constexpr (FMT == kFmtNv)
                                wf[nt][j] = unpack_nv(AT{}, w, j < 2 ? s_lo : s_hi);
                            else
                                wf[nt][j] = unpack_mx(AT{}, w, s_lo);
                        }
                    }
                    if constexpr (kAhead && !(PETIT_ABLATE_BATCH & 16)) {
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            wring[SLOT][nt] = buf_load16(w_rsrc, w_voff[nt], (kt + D) * kTileBytes, kAuxNt);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        Frag af[MT];
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            af[mt] = __builtin_bit_cast(Frag, fptr[(PETIT_ABLATE_BATCH & 8) ? 0 : j][cur + ((PETIT_ABLATE_BATCH & 8) ? 0 : mt * 256)]);
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) {
                                if constexpr (PETIT_ABLATE_BATCH & 4) {
                                    const u32x4 wb = __builtin_bit_cast(u32x4, wf[nt][j]), ab = __builtin_bit_cast(u32x4, af[mt]);
                                    acc[mt][nt][j] += __builtin_bit_cast(float, wb[0] ^ ab[1]);
                                } else {
                                    acc[mt][nt] = mfma16(wf[nt][j], af[mt], acc[mt][nt]);
                                }
                            }
                    }
                    // the step's fragment reads must have RETURNED before the loader may overwrite that slot (their MFMAs usually forced that already;
                    // nothing keeps hipcc from sinking an MFMA below the barrier, so say it); the weight loads stay in flight across the barrier
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                });
                if constexpr (!kLast) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        srec[nt] = srec_next[nt];
                }
            };
            if constexpr (AT::kAdaptive) {
                // Fp16Mx: the fast body while this lane's scale bytes of the span lie in 114..140; the first span that does not switches THIS WAVE, for the
                // rest of its range, to the fallback -- exact for any e8m0 scale, written for size, not speed (device_common.hpp): one k-tile per trip of a
                // rolled loop, the W tile and its scale byte loaded on the spot (the ring's tiles in flight are simply dropped), weights to bf16, every fp16
                // fragment split into hi + lo bf16 in registers, two MFMAs per word.  It keeps the workgroup's protocol: one barrier per step, the loader
                // wave never notices.
                auto needs_fallback = [&]() -> bool {
                    unsigned bad = 0;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        if ((unsigned)nt < valid_nt)
                            bad = mx_rec_outside_f16<KS>(srec[nt], bad);
                    return __builtin_amdgcn_ballot_w64(bad != 0) != 0;
                };
                unsigned sp = sp_begin;
                bool fb = needs_fallback();
                for (; sp + 1 < sp_end && !fb; ++sp) {
                    span_body(sp, std::false_type{});
                    fb = needs_fallback();
                }
                if (!fb) {
                    span_body(sp_end - 1, std::true_type{});
                } else {
#pragma unroll 1
                    for (unsigned kt = sp * KS; kt < sp_end * KS; ++kt) {
                        const unsigned cur = aslot * Cfg::kTileU4;
                        aslot = aslot + 1 == (unsigned)Cfg::kSlots ? 0u : aslot + 1;
                        u32x4 wt[NT];
                        float sc[NT];
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            wt[nt] = buf_load16(w_rsrc, w_voff[nt], kt * kTileBytes, kAuxNt);
                            const unsigned sb = __builtin_amdgcn_raw_buffer_load_b8(s_rsrc, s_voff[nt] + kt % KS, (kt / KS) * 64 * kRecBytes, kAuxDefault);
                            sc[nt] = __builtin_bit_cast(float, (sb & 0xffu) << 23);
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt) {
                                u32x4 hi, lo;
                                split_f16(fptr[j][cur + mt * 256], hi, lo);
#pragma unroll
                                for (int nt = 0; nt < NT; ++nt)
                                    acc[mt][nt] = mfma16_hilo(unpack_mx(Bf16{}, wt[nt][j], sc[nt]), hi, lo, acc[mt][nt]);
                            }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                        asm volatile("" ::: "memory");
                    }
                }
                // the join of the two bodies: see mfma_join_settle (device_common.hpp)
                static_for<0, 2>([&](auto pass) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            mfma_join_pin(acc[mt][nt]);
                    if constexpr (decltype(pass)::value == 0)
                        mfma_join_settle();
                });
            } else {
                for (unsigned sp = sp_begin; sp + 1 < sp_end; ++sp)
                    span_body(sp, std::false_type{});
                span_body(sp_end - 1, std::true_type{});
            }
        }
    } else if (part_on) {
        const unsigned kt_begin = sp_begin * KS;
        // --- prologue: D steps of loads, oldest first (A slice, then W, per step)
        ScaleRec<FMT, KS> srec[NT], srec_next[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            srec[nt] = load_scale_rec<FMT, KS>(s_rsrc, s_voff[nt], sp_begin * 64 * kRecBytes);
        u32x4 wring[D][NT];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            const unsigned kt = kt_begin + i; // (a part owns at least one span and D <= KS)
            dma_a_tile(kt % Cfg::kSlots, kt);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                wring[i][nt] = buf_load16(w_rsrc, w_voff[nt], kt * kTileBytes, kAuxNt);
        }
        // tile kt_begin visible to the whole part: D - 1 steps of loads may stay in flight
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * kLoadsPerStep) : "memory");
        __builtin_amdgcn_s_barrier();

        unsigned aslot = kt_begin % Cfg::kSlots; // LDS slot of the step's tile, advanced once per step
        auto span_body = [&](const unsigned sp, auto last_c) {
            constexpr bool kLast = decltype(last_c)::value;
            const unsigned kt0 = sp * KS;
            if constexpr (!kLast) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    srec_next[nt] = load_scale_rec<FMT, KS>(s_rsrc, s_voff[nt], (sp + 1) * 64 * kRecBytes);
            }
            static_for<0, KS>([&](auto t_c) {
                constexpr int T = decltype(t_c)::value;
                constexpr int SLOT = T % D;
                const unsigned kt = kt0 + T;
                const unsigned cur = aslot * Cfg::kTileU4; // (tile kt became visible at the barrier that ended the previous step)
                // loads of step kt + D (it exists unless this is the wave's last span and T + D runs past it)
                constexpr bool kAhead = !kLast || (T + D < KS);
                const unsigned ktn = kt + D;
                if constexpr (kAhead)
                    dma_a_tile(aslot == 0 ? (unsigned)D : aslot - 1, ktn); // (aslot + D) % (D + 1)
                aslot = aslot + 1 == (unsigned)Cfg::kSlots ? 0u : aslot + 1;
                // (hipcc otherwise hoists the pure unpack VALU of several steps of the unrolled span to the top: 256 VGPRs and spills)
                __builtin_amdgcn_sched_barrier(0);
                // this step's weight words, unpacked once, used by all MT m-tiles
                Frag wf[NT][4];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
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
                if constexpr (kAhead) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        wring[SLOT][nt] = buf_load16(w_rsrc, w_voff[nt], ktn * kTileBytes, kAuxNt);
                }
                // word j of every m-tile before word j + 1: MT * NT independent accumulators between two MFMAs on the same one
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    Frag af[MT];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        af[mt] = __builtin_bit_cast(Frag, fptr[j][cur + mt * 256]);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[mt][nt] = mfma16(wf[nt][j], af[mt], acc[mt][nt]);
                }
                // tile kt + 1 complete (this wave's slice), then visible (everybody's): the steps after it that have been
                // requested (D - 1 of them, fewer at the end of the wave's range) stay in flight
                constexpr int kYounger = !kLast ? D - 1 : (KS - 2 - T < 0 ? 0 : (KS - 2 - T < D - 1 ? KS - 2 - T : D - 1));
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kYounger * kLoadsPerStep) : "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            });
            if constexpr (!kLast) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    srec[nt] = srec_next[nt];
            }
        };
        for (unsigned sp = sp_begin; sp + 1 < sp_end; ++sp)
            span_body(sp, std::false_type{});
        span_body(sp_end - 1, std::true_type{});
    }
    // barrier count of the spans this part does not have (ragged K split): 1 for the prologue + KS per span
    {
        const unsigned mine = part_on ? 1u + (sp_end - sp_begin) * KS : 0u;
        const unsigned want = 1u + arg_spw * KS;
        for (unsigned i = mine; i < want; ++i)
            __builtin_amdgcn_s_barrier();
    }

    // --- cross-wave K reduction through LDS (the activation ring's memory: every wave is past its last fragment read), then the epilogue
    f32x4 *const red = reinterpret_cast<f32x4 *>(smem);
    constexpr int kItems = Cfg::kRedU4; // per K part: [(wn NT + nt) MT + mt][lane]
    if constexpr (WK > 1) {
        if (!loader) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    red[wk * kItems + ((wn * NT + nt) * MT + mt) * 64 + lane] = acc[mt][nt];
        }
        __syncthreads();
    }
    if (loader)
        return;
    const float gs = *arg_gs;
    // every thread finishes a share of the workgroup's (tile, lane) items: with WK parts each wave takes 1 / WK of ITS OWN tiles' items
    // (the other parts' copies of the same item come from LDS), so the stores stay 8 bytes per lane, 16 rows x 32 B per instruction
    constexpr int kPerWave = NT * MT; // accumulator tiles per wave
#pragma unroll
    for (int t = 0; t < kPerWave; ++t) {
        if (WK > 1 && (unsigned)(t % WK) != wk)
            continue;
        const int nt = t / MT, mt = t % MT; // (compile-time after unrolling)
        if (arg_act && nt % 2 != 0)
            continue; // the up half is consumed together with its gate tile
        const unsigned item = ((wn * NT + nt) * MT + mt) * 64 + lane;
        f32x4 v;
        if constexpr (WK > 1) {
            v = red[item];
#pragma unroll
            for (int q = 1; q < WK; ++q)
                v += red[q * kItems + item];
        } else {
            v = acc[mt][nt];
        }
        const unsigned m = m0 + mt * 16 + r;
        const unsigned ntile = nt0 + nt;
        if (m >= arg_m || (unsigned)nt >= valid_nt)
            continue;
        if (gridDim.z > 1) { // K split across workgroups: fp32 partial tile -> this slice's slab (plain product: the reduce pass finishes)
            *reinterpret_cast<f32x4 *>(arg_workspace + ((size_t)blockIdx.z * arg_m + m) * arg_n + ntile * 16 + g * 4) = v;
        } else if (arg_act) {
            if constexpr (NT % 2 == 0) {
                f32x4 u;
                if constexpr (WK > 1) {
                    u = red[item + MT * 64];
#pragma unroll
                    for (int q = 1; q < WK; ++q)
                        u += red[q * kItems + item + MT * 64];
                } else {
                    u = acc[mt][nt + 1 < NT ? nt + 1 : nt];
                }
                const unsigned n_half = arg_n >> 1, n = (ntile >> 1) * 16 + g * 4;
                *reinterpret_cast<uint2 *>((char *)arg_c + ((size_t)m * n_half + n) * 2) = finish4_silu_mul<AT>(v, u, gs, arg_bias, n, n_half);
            }
        } else {
            const unsigned n = ntile * 16 + g * 4;
            *reinterpret_cast<uint2 *>((char *)arg_c + ((size_t)m * arg_n + n) * 2) = finish4<AT>(v, gs, arg_bias, n);
        }
    }
}

} // namespace petit_amd
