// gemm_mid.hpp -- weight streaming for 5 <= M <= 16 with the activation tile SHARED by the waves of a workgroup.
//
// Same job and same arithmetic as gemm_stream.hpp's staged path (AM = 8 / 16), different organisation.  There every
// wave stages its own copy of the activation block, so a workgroup that wants more waves pays for them in activation
// traffic and staging work, and the fastest shape at M = 16 is ONE wave per SIMD that issues ~2300 instructions in
// order: measured issue-bound (52 % issuing, 30 % issue stalls, 18 % waiting on memory).
// Here a workgroup is WN x WK waves: K is split over WK parts as before, and the WN waves of a part (one or two
// n-tiles each) share ONE activation tile per k-tile in LDS.  Twice the waves per SIMD at the same activation traffic
// per column, half the instruction stream per wave.
//
//  * A goes global -> LDS directly (buffer_load ... lds, 16 B per lane, no VGPR staging, no ds_write), a tile =
//    AM rows x 256 B, the WN waves of the part load AM / (4 WN) KiB-slices each; rows are XOR-swizzled through the
//    SOURCE address so that the 16 rows a fragment read touches land on 16 different bank groups (as gemm_tiled.hpp).
//  * The loads of a wave retire in order (one vmcnt counter), so an activation tile requested LATER than the weights
//    still in flight could only be waited for by draining the weight ring.  The A tile of k-tile t + D is therefore
//    requested together with the W tile of t + D, into a ring of D + 1 LDS slots per K part, and the wait at the end
//    of a step is a COUNTED vmcnt that leaves the later steps' loads in flight (D - 1 steps, fewer in the last span of
//    the wave's range, where nothing is requested past its end: the count is a compile-time constant per step),
//    followed by a bare s_barrier.
//  * One barrier per k-tile.  Slot (t + D) % (D + 1) was last read in step t - 1, which every wave left at the
//    barrier that ended it.
//  * W never touches LDS; unpack and MFMA exactly as gemm_stream.hpp (C^T = W . A^T, 16x16x32).
//  * Partial sums of the WK parts meet once, in LDS, after the loop (as gemm_stream.hpp).
#pragma once

#include "device_common.hpp"

namespace petit_amd {

//   AT    Bf16 / Fp16 activations (and output)
//   FMT   kFmtNv / kFmtMx
//   KS    tiles per span
//   NT    n-tiles per wave
//   WN,WK waves along N / K in the workgroup (the WN waves of a K part share its activation tiles)
//   D     ring depth in k-tiles (W in VGPRs, A in LDS slots), divides KS
//   AM    activation rows staged: 8 or 16 (M <= AM)
template <class AT_, int FMT_, int KS_, int NT_, int WN_, int WK_, int D_, int AM_> struct MidCfg {
    using AT = AT_;
    static constexpr int FMT = FMT_, KS = KS_, NT = NT_, WN = WN_, WK = WK_, D = D_, AM = AM_;
    static constexpr int kThreads = 64 * WN * WK;
    static constexpr int kTileU4 = AM * 16;               // one activation tile: AM rows x 16 units of 16 B
    static constexpr int kSlots = D + 1;
    static constexpr int kPartU4 = kSlots * kTileU4;      // per K part
    static constexpr int kDma = AM * 16 / 64 / WN;        // KiB wave-loads per wave per tile
    static constexpr int kRedU4 = WN * NT * 64;           // float4 partial outputs per K part
    static constexpr int kSmemU4 = WK * kPartU4 + WK * kRedU4;
    static_assert(AM == 8 || AM == 16, "rows: 8 or 16");
    static_assert(kDma >= 1 && kDma * WN * 4 == AM, "the waves of a K part split a tile into whole KiB loads");
    static_assert(KS % D == 0, "ring depth must divide the span");
    static_assert(!AT::kBfp, "plain bf16 / fp16 activations (or Fp16Mx: fast body + exact fallback, device_common.hpp)");
    static_assert(!AT::kAdaptive || FMT == kFmtMx, "Fp16Mx: fp16 activations x MXFP4 weights");
    static_assert(kThreads <= 1024 && kSmemU4 * 16 <= 160 * 1024, "workgroup / LDS budget");
};

template <class Cfg>
__global__ __launch_bounds__(Cfg::kThreads) void gemm_mid_kernel(const void *arg_w, const void *arg_s, const void *arg_a,
                                                                 unsigned arg_k, unsigned arg_n, unsigned arg_m, unsigned arg_spw,
                                                                 unsigned arg_act, void *arg_c, const float *arg_gs,
                                                                 const void *arg_bias, unsigned arg_dbg) {
    using AT = typename Cfg::AT;
    using Frag = typename AT::frag;
    constexpr int FMT = Cfg::FMT, KS = Cfg::KS, NT = Cfg::NT, WN = Cfg::WN, WK = Cfg::WK, D = Cfg::D, AM = Cfg::AM;
    constexpr unsigned kRecBytes = ScaleRec<FMT, KS>::kBytes;
    constexpr unsigned kOob = 0x80000000u;
    // loads a wave issues per step: its slice of the A tile + one W tile per n-tile
    constexpr int kLoadsPerStep = Cfg::kDma + NT;

    __shared__ u32x4 smem[Cfg::kSmemU4];

    const unsigned lane = threadIdx.x & 63u;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned wn = wave % WN, wk = wave / WN;
    const unsigned r = lane & 15u, g = lane >> 4;

    const unsigned ktiles = arg_k / kTileK;
    const unsigned nspans = ktiles / KS;
    const unsigned ntiles = arg_n / kTileN;
    const unsigned nt0 = (blockIdx.x * WN + wn) * NT;
    // every wave walks arg_spw spans (the barrier count must agree across the workgroup); a part whose range ends early
    // idles through the rest
    const unsigned sp_begin = min(wk * arg_spw, nspans);
    const unsigned sp_end = min(sp_begin + arg_spw, nspans);

    constexpr int NACC = (NT <= 2) ? 2 : 1;
    f32x4 acc[NT][NACC];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int q = 0; q < NACC; ++q)
            acc[nt][q] = f32x4{0.f, 0.f, 0.f, 0.f};

    u32x4 *const a_part = smem + wk * Cfg::kPartU4;
    const unsigned valid_nt = nt0 < ntiles ? min((unsigned)NT, ntiles - nt0) : 0u;
    const unsigned w_row_bytes = ktiles * kTileBytes;
    const unsigned s_row_bytes = (FMT == kFmtNv) ? arg_k : arg_k / 2;
    const unsigned rows = min(arg_m, (unsigned)AM);
    const unsigned pt0 = valid_nt ? physical_tile(nt0, ntiles, arg_act) : 0u;
    const unsigned span_tiles = !valid_nt ? 0u : arg_act ? (valid_nt >> 1) + (ntiles >> 1) : valid_nt;
    const __amdgpu_buffer_rsrc_t w_rsrc = make_rsrc((const char *)arg_w + (size_t)pt0 * w_row_bytes, span_tiles * w_row_bytes);
    const __amdgpu_buffer_rsrc_t s_rsrc = make_rsrc((const char *)arg_s + (size_t)pt0 * s_row_bytes, span_tiles * s_row_bytes);
    const __amdgpu_buffer_rsrc_t a_rsrc = make_rsrc(arg_a, rows * arg_k * 2);

    unsigned w_voff[NT], s_voff[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const unsigned rel = valid_nt ? physical_tile(nt0 + nt, ntiles, arg_act) - pt0 : 0u;
        w_voff[nt] = ((unsigned)nt < valid_nt) ? lane * 16 + rel * w_row_bytes : kOob;
        s_voff[nt] = ((unsigned)nt < valid_nt) ? lane * kRecBytes + rel * s_row_bytes : kOob;
    }
    // A tile slice of this wave: wave-load i covers rows 4 (i WN + wn) .. + 3; lane l -> row + l / 16, position l % 16,
    // which receives unit (l % 16) ^ row of that row.  Rows >= M fall out of the descriptor: zeros.
    unsigned dma_voff[Cfg::kDma];
#pragma unroll
    for (int i = 0; i < Cfg::kDma; ++i) {
        const unsigned row = 4 * (i * WN + wn) + (lane >> 4);
        dma_voff[i] = row * arg_k * 2 + (((lane & 15u) ^ row) * 16);
    }
    auto dma_a_tile = [&](unsigned slot, unsigned kt) {
#pragma unroll
        for (int i = 0; i < Cfg::kDma; ++i) {
#if defined(__HIP_DEVICE_COMPILE__) // (the host pass knows neither the builtin nor the LDS address space)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                a_rsrc, (__attribute__((address_space(3))) void *)(a_part + slot * Cfg::kTileU4 + (i * WN + wn) * 64), 16,
                dma_voff[i], kt * 256, 0, 0);
#else
            (void)slot, (void)kt;
#endif
        }
    };
    // fragment of MFMA j: row m = r (rows >= AM read row r % AM: those output columns are never stored)
    const unsigned frow = r % (unsigned)AM;
    const u32x4 *fptr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        fptr[j] = a_part + (int)(frow * 16 + ((g * 4 + j) ^ frow));

    const bool part_on = sp_begin < sp_end;
    if (part_on) {
        const unsigned kt_begin = sp_begin * KS;
        // --- prologue: D steps of loads, oldest first (A slice, then W, per step) ------------------------------------
        ScaleRec<FMT, KS> srec[NT], srec_next[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            srec[nt] = load_scale_rec<FMT, KS>(s_rsrc, s_voff[nt], sp_begin * 64 * kRecBytes);
        u32x4 wring[D][NT];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            const unsigned kt = kt_begin + i; // (a part owns at least one span and D <= KS)
            dma_a_tile((kt_begin + i) % Cfg::kSlots, kt);
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
                // this step's fragments (tile kt became visible at the barrier that ended the previous step)
                Frag af[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    af[j] = __builtin_bit_cast(Frag, fptr[j][aslot * Cfg::kTileU4]);
                // loads of step kt + D (it exists unless this is the wave's last span and T + D runs past it)
                constexpr bool kAhead = !kLast || (T + D < KS);
                const unsigned ktn = kt + D;
                if constexpr (kAhead)
                    dma_a_tile(aslot == 0 ? (unsigned)D : aslot - 1, ktn); // (aslot + D) % (D + 1)
                aslot = aslot + 1 == (unsigned)Cfg::kSlots ? 0u : aslot + 1;
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
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[nt][j % NACC] = mfma16(wf[nt][j], af[j], acc[nt][j % NACC]);
                // tile kt + 1 complete (this wave's slice), then visible (everybody's): the steps after it that have been
                // requested (D - 1 of them, fewer at the end of the wave's range) stay in flight
                constexpr int kYounger = !kLast ? D - 1 : (KS - 2 - T < 0 ? 0 : (KS - 2 - T < D - 1 ? KS - 2 - T : D - 1));
#ifdef PETIT_AMD_DEBUG_WAITS // debug builds only: $PETIT_AMD_MID_DEBUG=1 drains every step (isolates a wrong wait count)
                if (arg_dbg & 1u)
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else
#endif
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kYounger * kLoadsPerStep) : "memory");
                __builtin_amdgcn_s_barrier();
            });
            if constexpr (!kLast) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    srec[nt] = srec_next[nt];
            }
        };
        if constexpr (AT::kAdaptive) {
            // Fp16Mx: the fast body while the span's scale record (this lane's bytes of this wave's n-tiles) lies in 114..140; the first span
            // that does not switches THIS WAVE to the fallback for the rest of its range -- exact for any e8m0 scale, written for size, not
            // speed: one k-tile per trip of a rolled loop, the W tile and its scale byte loaded on the spot, weights to bf16, the fp16
            // fragments split into hi + lo bf16 in registers, two MFMAs per word.  It keeps the workgroup's protocol (its share of the
            // activation-tile DMA D steps ahead, one barrier per step; every wait is a full drain), so the other waves never notice.
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
                    u32x4 af[4], wt[NT];
                    float sc[NT];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        af[j] = fptr[j][aslot * Cfg::kTileU4];
                    if (kt + D < sp_end * KS)
                        dma_a_tile(aslot == 0 ? (unsigned)D : aslot - 1, kt + D);
                    aslot = aslot + 1 == (unsigned)Cfg::kSlots ? 0u : aslot + 1;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        wt[nt] = buf_load16(w_rsrc, w_voff[nt], kt * kTileBytes, kAuxNt);
                        const unsigned sb = __builtin_amdgcn_raw_buffer_load_b8(s_rsrc, s_voff[nt] + kt % KS, (kt / KS) * 64 * kRecBytes, kAuxDefault);
                        sc[nt] = __builtin_bit_cast(float, (sb & 0xffu) << 23);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        u32x4 hi, lo;
                        split_f16(af[j], hi, lo);
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[nt][0] = mfma16_hilo(unpack_mx(Bf16{}, wt[nt][j], sc[nt]), hi, lo, acc[nt][0]);
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                }
            }
            // the join of the two bodies: see mfma_join_settle (device_common.hpp)
            static_for<0, 2>([&](auto pass) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int q = 0; q < NACC; ++q)
                        mfma_join_pin(acc[nt][q]);
                if constexpr (decltype(pass)::value == 0)
                    mfma_join_settle();
            });
        } else {
            for (unsigned sp = sp_begin; sp + 1 < sp_end; ++sp)
                span_body(sp, std::false_type{});
            span_body(sp_end - 1, std::true_type{});
        }
    }
    // barrier count of the spans this part does not have (ragged K split): 1 for the prologue + KS per span
    {
        const unsigned mine = part_on ? 1u + (sp_end - sp_begin) * KS : 0u;
        const unsigned want = 1u + arg_spw * KS;
        for (unsigned i = mine; i < want; ++i)
            __builtin_amdgcn_s_barrier();
    }

    // --- cross-wave K reduction through LDS, then the epilogue (as gemm_stream.hpp) ----------------------------------
    const float gs = *arg_gs;
    f32x4 *const red = reinterpret_cast<f32x4 *>(smem + WK * Cfg::kPartU4);
    constexpr int kItems = Cfg::kRedU4;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        f32x4 v = acc[nt][0];
        if constexpr (NACC == 2)
            v += acc[nt][1];
        red[wk * kItems + (wn * NT + nt) * 64 + lane] = v;
    }
    __syncthreads();
    for (unsigned item = threadIdx.x; item < (unsigned)kItems; item += Cfg::kThreads) {
        const unsigned tile = item >> 6, il = item & 63u;
        if (arg_act && (tile % NT) % 2 != 0)
            continue; // the up half is consumed together with its gate tile
        f32x4 v = red[item];
#pragma unroll
        for (int q = 1; q < WK; ++q)
            v += red[q * kItems + item];
        const unsigned m = il & 15u;
        const unsigned ntile = (blockIdx.x * WN + tile / NT) * NT + tile % NT;
        if (m >= arg_m || ntile >= ntiles)
            continue;
        if (arg_act) {
            if constexpr (NT % 2 == 0) {
                f32x4 u = red[item + 64];
#pragma unroll
                for (int q = 1; q < WK; ++q)
                    u += red[q * kItems + item + 64];
                const unsigned n_half = arg_n >> 1, n = (ntile >> 1) * 16 + (il >> 4) * 4;
                *reinterpret_cast<uint2 *>((char *)arg_c + ((size_t)m * n_half + n) * 2) =
                    finish4_silu_mul<AT>(v, u, gs, arg_bias, n, n_half);
            }
        } else {
            const unsigned n = ntile * 16 + (il >> 4) * 4;
            *reinterpret_cast<uint2 *>((char *)arg_c + ((size_t)m * arg_n + n) * 2) = finish4<AT>(v, gs, arg_bias, n);
        }
    }
}

} // namespace petit_amd
