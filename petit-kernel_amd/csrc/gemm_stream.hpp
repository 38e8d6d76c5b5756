// gemm_stream.hpp -- the weight-streaming FP4 GEMM for small M (decode regime).
//
// Replaces GemmFp4Fp16KernelGrid and everything under it
// (fp4/gemm_fp4_fp16_grid.cuh:441-498, fp4/warp_schedule_fp16.cuh:73-193,
// quantization/dequant.cuh, quantization/memory_ops.cuh, qgemm.cuh,
// gpu/quantization/reduce.cuh) for M small enough that the op is a scan of W.
//
// Design (gfx950):
//  * W is used exactly once, so it never touches LDS: every wave streams whole
//    1 KiB tiles (layout.h) straight into VGPRs with non-temporal
//    buffer_load_dwordx4, a ring of D tiles per n-tile deep, and there is no
//    barrier anywhere in the main loop (the reference round-trips every W byte
//    global -> VGPR -> LDS -> VGPR with 2-4 barriers per K step,
//    gemm_fp4_fp16_grid.cuh:353-394).
//  * Unpack is the hardware E2M1 convert (v_cvt_scalef32_pk_*_fp4): 4 VALU per
//    8 weights for MX (block scale folded into the convert), +4 v_pk_mul for the
//    non-power-of-two NV group scale.  The reference needs bfrev/and/cvt_bf8/
//    perm sequences and a nibble re-encode (dequant.cuh:326-363).
//  * The contraction is v_mfma_f32_16x16x32_{bf16,f16} computed as
//    C^T[n][m] = W[n][k] . A[m][k]: W is the 16-row operand, the activations are
//    the 16-column operand, so a lane ends with 4 consecutive n of one m and the
//    epilogue store is one 8-byte write.
//  * K is split across the WK waves of a workgroup (and optionally across
//    gridDim.z workgroups); partial sums meet once, in LDS, after the loop.
//  * A rows beyond M and n-tiles beyond N/16 are masked by buffer-descriptor
//    bounds (out-of-range loads return 0), so M and N/16 need not divide the
//    tile.
#pragma once

#include "device_common.hpp"

namespace petit_amd {

// Compile-time configuration of one kernel instance.
//   AT    Bf16 / Fp16 activations (and output)
//   FMT   kFmtNv / kFmtMx
//   KS    tiles per span (scale-record granularity; layout.h)
//   MT    m-tiles (of 16) per workgroup
//   NT    n-tiles (of 16) per wave
//   WN,WK waves along N / K in the workgroup
//   D     ring depth (tiles in flight per n-tile), D divides KS
//   AM    activation path: 0 = fragment-shaped loads straight from L2 (any M);
//         1, 2, 4, 8, 16 = the wave stages AM rows x SL tiles of A in its private LDS
//         slice with fully coalesced loads and reads MFMA fragments back with
//         ds_read_b128 (M <= AM; needs MT == 1).  SL = one span for AM <= 4, and 4 / 2
//         tiles for AM = 8 / 16 (8 KiB per wave either way).  Measured on MI355X at
//         M = 1, 8192^2: the 36 four-lane buffer_loads per span of the direct path cost
//         3 us of a 12 us launch (TA issue-bound), see DESIGN.md.
//   ABL   ablation bits for tools/ablate (0 in every shipped kernel; 16 = per-wave s_memrealtime stamps into p.workspace):
//         1 no activation loads, 2 no unpack, 4 no MFMA, 8 empty kernel
//   PA    direct path (AM == 0) only: how many tiles ahead the activation fragments are
//         requested from L2 (ring of PA fragment sets; PA divides KS)
template <class AT_, int FMT_, int KS_, int MT_, int NT_, int WN_, int WK_, int D_, int AM_ = 0, int ABL_ = 0, int PA_ = 1>
struct StreamCfg {
    using AT = AT_;
    static constexpr int FMT = FMT_, KS = KS_, MT = MT_, NT = NT_, WN = WN_, WK = WK_, D = D_, AM = AM_, ABL = ABL_;
    static constexpr int PA = PA_;
    static_assert(KS % PA_ == 0 && PA_ >= 1, "activation prefetch distance must divide the span");
    static constexpr int kThreads = 64 * WN * WK;
    // stage length in tiles: a whole span while that is <= 8 KiB per wave, else 32 / AM tiles
    static constexpr int kStageRows = 32; // rows x tiles per stage (8 KiB per wave; 16 KiB measured slower)
    static constexpr int SL = (AM <= 4) ? KS : ((kStageRows / (AM > 0 ? AM : 1)) < KS ? kStageRows / (AM > 0 ? AM : 1) : KS);
    // one staged activation row: SL tiles x 256 B, padded by one 16-byte slot so
    // that the rows a ds_read_b128 lane group touches fall on different banks
    static constexpr int kARowU4 = SL * 16 + 1;
    static constexpr int kALdsU4 = AM * kARowU4;                       // per wave
    static constexpr int kRedItems = WN * MT * NT * 64;                // float4 outputs per workgroup
    static constexpr int kSmemU4 = WN * WK * kALdsU4 + (WK > 1 ? WK * kRedItems : 0);
    static_assert(KS % D == 0, "ring depth must divide the span");
    static_assert(kThreads <= 1024, "workgroup too large");
    static_assert(AM == 0 || (MT == 1 && (AM == 1 || AM == 2 || AM == 4 || AM == 8 || AM == 16)),
                  "staged path: MT == 1, AM in {1,2,4,8,16}");
    static_assert(AM <= 4 || !AT::kBfp, "AM = 8 / 16: not with block-floating-point activations");
    static_assert(AM == 0 || KS % SL == 0, "stage length must divide the span");
    static_assert(kSmemU4 * 16 <= 160 * 1024, "LDS budget");
    static_assert(!AT::kBfp || (AM > 0 && KS >= 4 && FMT == 0), "block-floating-point A: staged NVFP4 path only");
    static_assert(!AT::kAdaptive || FMT == kFmtMx, "Fp16Mx: fp16 activations x MXFP4 weights");
};

// Scalar kernel arguments, most urgent first (not one GemmArgs struct): with -mllvm -amdgpu-kernarg-preload-count the
// command processor hands the leading ones to every wave in SGPRs at launch, so a wave computes its first W / scale / A
// addresses without waiting for the s_load round trip of the kernarg segment -- the decode kernel is all prologue
// (every wave issues its whole share of loads at once), and that round trip sits in front of all of them.
// (block_x: the workgroup's index along N -- blockIdx.x for a plain launch, the index inside its member for a grouped one)
// kCombine (gemm_stream_combine_kernel below): a K split across workgroups (gridDim.z) whose partial sums meet INSIDE the launch -- every
// slice publishes its fp32 partial tile in its slab, takes a ticket for the output tile, and the last arriver sums the slabs in slice
// order (deterministic) and finishes the epilogue -- instead of in a second launch (splitk_reduce_kernel).
template <class Cfg, bool kCombine = false>
__device__ __forceinline__ void gemm_stream_body(const void *arg_w, const void *arg_s, const void *arg_a, unsigned arg_k, unsigned arg_n,
                                                 unsigned arg_m, unsigned arg_spw, unsigned arg_act, void *arg_c, const float *arg_gs,
                                                 const void *arg_bias, float *arg_workspace, const unsigned block_x,
                                                 unsigned *arg_tickets = nullptr) {
    GemmArgs p;
    p.c = arg_c, p.a = arg_a, p.w = arg_w, p.s = arg_s, p.gs = arg_gs, p.bias = arg_bias, p.act = arg_act;
    p.workspace = arg_workspace, p.m = arg_m, p.n = arg_n, p.k = arg_k, p.spans_per_wave = arg_spw, p.flags = 0;
    using AT = typename Cfg::AT;
    constexpr int FMT = Cfg::FMT, KS = Cfg::KS, MT = Cfg::MT, NT = Cfg::NT;
    constexpr int WN = Cfg::WN, WK = Cfg::WK, D = Cfg::D, AM = Cfg::AM, ABL = Cfg::ABL, PA = Cfg::PA;
    constexpr unsigned kRecBytes = ScaleRec<FMT, KS>::kBytes;
    // A voffset this large is out of range for every descriptor built below
    // whatever the generation's rule for soffset is (masked loads use soffset 0).
    constexpr unsigned kOob = 0x80000000u;

    // ONE LDS object: [per-wave activation slices][cross-wave reduction scratch]
    __shared__ u32x4 smem[Cfg::kSmemU4 > 0 ? Cfg::kSmemU4 : 1];

    const unsigned lane = threadIdx.x & 63u;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned wn = wave % WN, wk = wave / WN;
    const unsigned r = lane & 15u, g = lane >> 4;
    [[maybe_unused]] unsigned long long ts[4] = {0, 0, 0, 0};
    if constexpr (ABL & 16)
        ts[0] = __builtin_amdgcn_s_memrealtime();

    const unsigned ktiles = p.k / kTileK;
    const unsigned nspans = ktiles / KS;
    const unsigned ntiles = p.n / kTileN;
    const unsigned nt0 = (block_x * WN + wn) * NT;
    const unsigned m0 = blockIdx.y * (16 * MT);

    // K range of this wave: contiguous spans.  gridDim.z splits K across
    // workgroups first, WK across the waves of a workgroup second.
    const unsigned part = blockIdx.z * WK + wk;
    const unsigned chunk = p.spans_per_wave; // ceil(nspans / (gridDim.z * WK)), computed on the host
    const unsigned sp_begin = min(part * chunk, nspans);
    const unsigned sp_end = min(sp_begin + chunk, nspans);

    // two accumulators per output tile (even / odd MFMA of a tile) when registers
    // allow, so consecutive MFMAs do not wait on each other's result
    constexpr int NACC = (MT * NT <= 2) ? 2 : 1;
    f32x4 acc[MT][NT][NACC];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int q = 0; q < NACC; ++q)
                acc[mt][nt][q] = f32x4{0.f, 0.f, 0.f, 0.f};

    // block-floating-point path: acc holds ONE span's partial sums, total the running sum
    f32x4 total[AT::kBfp ? NT : 1];
#pragma unroll
    for (int nt = 0; nt < (AT::kBfp ? NT : 1); ++nt)
        total[nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    if ((ABL & 8) == 0 && nt0 < ntiles && sp_begin < sp_end) {
        const unsigned valid_nt = min((unsigned)NT, ntiles - nt0);
        const unsigned w_row_bytes = ktiles * kTileBytes;             // one n-tile of W
        const unsigned s_row_bytes = (FMT == kFmtNv) ? p.k : p.k / 2; // one n-tile of scales
        const unsigned rows = min(p.m - m0, (unsigned)(16 * MT));

        // tile offsets relative to the wave's first physical tile (identity mapping: nt; SiLU-mul: gate/up pairs)
        const unsigned pt0 = physical_tile(nt0, ntiles, p.act);
        const unsigned span_tiles = p.act ? (valid_nt >> 1) + (ntiles >> 1) : valid_nt; // tiles the descriptor must cover
        const __amdgpu_buffer_rsrc_t w_rsrc =
            make_rsrc((const char *)p.w + (size_t)pt0 * w_row_bytes, span_tiles * w_row_bytes);
        const __amdgpu_buffer_rsrc_t s_rsrc =
            make_rsrc((const char *)p.s + (size_t)pt0 * s_row_bytes, span_tiles * s_row_bytes);
        const char *a_base = (const char *)p.a + (size_t)m0 * p.k * 2;
        if constexpr (ABL & 32) // tools/ablate: every workgroup reads its own copy of A (is the shared A a hot spot in L2?)
            a_base += (size_t)(block_x & 63u) * p.m * p.k * 2;
        const __amdgpu_buffer_rsrc_t a_rsrc = make_rsrc(a_base, rows * p.k * 2);

        // Everything that decides validity lives in the VGPR offset (bounds
        // checked on every generation); the SGPR offset only walks along K.
        unsigned w_voff[NT], s_voff[NT], a_voff[MT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const unsigned rel = physical_tile(nt0 + nt, ntiles, p.act) - pt0;
            w_voff[nt] = ((unsigned)nt < valid_nt) ? lane * 16 + rel * w_row_bytes : kOob;
            s_voff[nt] = ((unsigned)nt < valid_nt) ? lane * kRecBytes + rel * s_row_bytes : kOob;
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            a_voff[mt] = ((mt * 16 + r) * p.k + g * kLaneK) * 2; // rows >= M fall out of range

        const unsigned kt_begin = sp_begin * KS;

        // --- prologue: activations first (they are needed first), then scale
        // records, then the W ring
        u32x4 *const a_lds = smem + wave * Cfg::kALdsU4;               // staged path only
        const unsigned a_frag_base = ((r < (unsigned)AM) ? r : 0u) * Cfg::kARowU4 + g * 4;
        constexpr int SL = Cfg::SL;                                    // tiles per stage
        constexpr int kAStageLoads = AM * SL / 4;                      // 1 KiB per wave-load
        u32x4 astage[kAStageLoads > 0 ? kAStageLoads : 1];             // next stage, in flight
        // stage `st` = tiles [st*SL, st*SL + SL) of K (absolute index: span * (KS/SL) + stage in span)
        auto issue_a_stage = [&](unsigned st, bool ok) {
            (void)ok;
            if constexpr (AM > 0 && (ABL & 1) == 0) {
#pragma unroll
                for (int i = 0; i < kAStageLoads; ++i) {
                    unsigned vo;
                    if constexpr (SL >= 4) { // a row of the stage is SL/4 whole wave-loads
                        constexpr int kPerRow = SL / 4;
                        vo = (i / kPerRow) * p.k * 2 + (i % kPerRow) * 1024 + lane * 16;
                    } else { // SL == 2: one wave-load covers two rows of 512 B
                        vo = (2 * i + (lane >> 5)) * p.k * 2 + (lane & 31u) * 16;
                    }
                    astage[i] = buf_load16(a_rsrc, vo, st * (SL * 256), kAuxDefault);
                }
            }
        };
        float bfp_up = 1.0f; // 2^sh of this lane's activation row for the span now in LDS
        // Fp16Mx: wave-uniform "the span whose record is in srec (and every later one of this wave) runs the exact fallback body";
        // set from the record itself (mx_rec_outside_f16), first after the prologue's loads, then whenever srec advances
        bool mx_fb = false;
        // Block-floating-point spans are EXACT or not taken at all: a span whose rows all satisfy
        // (smallest non-zero exponent) >= (largest exponent) - 31 converts to fp16 without losing a bit (a bf16 value
        // has 8 significant bits; after the shift its last bit sits at >= 2^-24, fp16's subnormal step, and
        // cvt_pkrtz is exact on representable values).  Any other span -- outliers, bf16 subnormals next to normal
        // values, inf / NaN among finite data -- stays bf16 in LDS and runs through the bf16 pipeline
        // (span_body<..., fallback>), so the kernel's result never depends on the activations' dynamic range.
        // The switch is one-way per wave (first inexact span onwards), which keeps the control flow a chain of two loops.
        bool bfp_fb = false; // wave-uniform: the span now in LDS (and every later one of this wave) is raw bf16
        auto write_a_stage = [&](auto raw_c) {
            constexpr bool kRaw = decltype(raw_c)::value; // already switched: no check, no conversion
            if constexpr (AM > 0 && (ABL & 1) == 0) {
                if constexpr (AT::kBfp && !kRaw) {
                    constexpr int kPerRow = KS / 4;
                    float up_row[AM], dn_row[AM];
                    int inexact = 0;
#pragma unroll
                    for (int rr = 0; rr < AM; ++rr) {
                        // largest and smallest non-zero |bf16| bit pattern of the row's span, two halves at a time:
                        // per lane, then across the wave.  (pattern - 1 wraps zero to 0xffff, so zeros never win the min.)
                        u16x2 mx2 = u16x2{0, 0}, mn2 = u16x2{0xffff, 0xffff};
#pragma unroll
                        for (int q = 0; q < kPerRow; ++q)
#pragma unroll
                            for (int d = 0; d < 4; ++d) {
                                const unsigned v = astage[rr * kPerRow + q][d] & 0x7fff7fffu;
                                const u16x2 pk = __builtin_bit_cast(u16x2, v);
                                mx2 = __builtin_elementwise_max(mx2, pk);
                                mn2 = __builtin_elementwise_min(mn2, (u16x2)(pk - u16x2{1, 1}));
                            }
                        const unsigned short mx_l = mx2.x > mx2.y ? mx2.x : mx2.y, mn_l = mn2.x < mn2.y ? mn2.x : mn2.y;
                        u16x2 c = u16x2{mx_l, (unsigned short)(0xffffu - mn_l)}; // one max-reduction serves both
#pragma unroll
                        for (int off = 32; off >= 1; off >>= 1) {
                            const unsigned cu = __builtin_bit_cast(unsigned, c);
                            const unsigned co = (unsigned)__shfl_xor((int)cu, off);
                            c = __builtin_elementwise_max(c, __builtin_bit_cast(u16x2, co));
                        }
                        const unsigned mx = c.x, mn_m1 = 0xffffu - c.y; // mn_m1 = (smallest non-zero pattern) - 1
                        int sh = (int)(mx >> 7) - 141; // biased exponent - 127 - 14
                        sh = mx == 0 ? 0 : min(max(sh, -126), 126);
                        // exact iff every non-zero element's last bit survives: E - 134 - sh >= -24
                        inexact |= (mn_m1 != 0xffffu && (int)((mn_m1 + 1u) >> 7) < sh + 110) ? 1 : 0;
                        dn_row[rr] = __builtin_bit_cast(float, (unsigned)(127 - sh) << 23);
                        up_row[rr] = __builtin_bit_cast(float, (unsigned)(127 + sh) << 23);
                    }
                    bfp_fb = __builtin_amdgcn_readfirstlane(inexact) != 0;
                    if (!bfp_fb) {
#pragma unroll
                        for (int rr = 0; rr < AM; ++rr) {
                            const float dn = dn_row[rr];
#pragma unroll
                            for (int q = 0; q < kPerRow; ++q)
#pragma unroll
                                for (int d = 0; d < 4; ++d) {
                                    const unsigned v = astage[rr * kPerRow + q][d];
                                    const unsigned lo_bits = v << 16, hi_bits = v & 0xffff0000u;
                                    const float lo = __builtin_bit_cast(float, lo_bits) * dn;
                                    const float hi = __builtin_bit_cast(float, hi_bits) * dn;
                                    const auto h2 = __builtin_amdgcn_cvt_pkrtz(lo, hi); // exact: checked above
                                    astage[rr * kPerRow + q][d] = __builtin_bit_cast(unsigned, h2);
                                }
                        }
                        bfp_up = up_row[0];
#pragma unroll
                        for (int rr = 1; rr < AM; ++rr)
                            bfp_up = (r == (unsigned)rr) ? up_row[rr] : bfp_up;
                    } else {
                        bfp_up = 1.0f;
                    }
                }
#pragma unroll
                for (int i = 0; i < kAStageLoads; ++i) {
                    unsigned dst;
                    if constexpr (SL >= 4) {
                        constexpr int kPerRow = SL / 4;
                        dst = (i / kPerRow) * Cfg::kARowU4 + (i % kPerRow) * 64 + lane;
                    } else {
                        dst = (2 * i + (lane >> 5)) * Cfg::kARowU4 + (lane & 31u);
                    }
                    a_lds[dst] = astage[i];
                }
            }
        };
        if constexpr (AM > 0) {
            static_assert(AM * SL >= 4, "staged path needs at least one full wave-load per stage");
            issue_a_stage(sp_begin * (KS / SL), true);
        }
        ScaleRec<FMT, KS> srec[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            srec[nt] = load_scale_rec<FMT, KS>(s_rsrc, s_voff[nt], sp_begin * 64 * kRecBytes);
        u32x4 wring[D][NT];
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) // D <= KS tiles always exist
                wring[i][nt] = buf_load16(w_rsrc, w_voff[nt], (kt_begin + i) * kTileBytes, kAuxNt);

        u32x4 afrag[MT][4];
        // direct path: ring of PA fragment sets, slot T % PA holds tile T of the span
        u32x4 aring[AM == 0 ? PA : 1][MT][4];
        if constexpr (AM == 0) {
#pragma unroll
            for (int i = 0; i < PA; ++i)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if constexpr (ABL & 1)
                            aring[i][mt][j] = u32x4{lane, lane + j, 0x3f803f80u, 0x3f803f80u};
                        else
                            aring[i][mt][j] = buf_load16(a_rsrc, a_voff[mt] + j * 16, (kt_begin + i) * 256, kAuxDefault);
                    }
        }
        auto load_first_frags = [&]() { // fragments of tile 0 of the stage now in LDS
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (ABL & 1)
                    afrag[0][j] = u32x4{lane, lane + j, 0x3f803f80u, 0x3f803f80u};
                else
                    afrag[0][j] = a_lds[a_frag_base + j];
            }
        };
        if constexpr (AM > 0) {
            write_a_stage(std::false_type{}); // first stage's activations -> LDS
            load_first_frags();
            if constexpr (SL < KS)
                issue_a_stage(sp_begin * (KS / SL) + 1, true);
        }

        // One span (KS tiles).  kLast: the wave's final span -- nothing further is
        // prefetched, so a wave that owns a single span issues exactly the loads it
        // uses.  Within a step the order is: read this tile's fragments, unpack +
        // MFMA out of ring slot T % D, THEN refill that slot D tiles ahead; the
        // sched_barrier pins that order (hipcc otherwise sinks every refill to the end
        // of the span and copies the ring at the back edge, draining the pipeline).
        ScaleRec<FMT, KS> srec_next[NT];
        // Fp16Mx: is any scale byte of the span now in srec outside 114..140?  (n-tiles past N hold zeros: not looked at)
        auto mx_span_needs_fallback = [&]() -> bool {
            if constexpr (AT::kAdaptive) {
                unsigned bad = 0;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    if ((unsigned)nt < valid_nt)
                        bad = mx_rec_outside_f16<KS>(srec[nt], bad);
                return __builtin_amdgcn_ballot_w64(bad != 0) != 0;
            } else {
                return false;
            }
        };
        // Fp16Mx fallback, spans [sp_from, sp_end) of this wave: exact for ANY e8m0 scale, written for size, not speed (no checkpoint
        // should ever get here: DESIGN.md section 3.1): one k-tile per trip of a rolled loop, weights and activation fragments straight
        // from global / L2 (rows >= M are outside a_rsrc: zeros), the tile's scale byte loaded by itself; weights to bf16 (exact), every
        // fp16 fragment split into hi + lo bf16 in registers, two bf16 MFMAs per word.  (What the fast body had prefetched is dropped.)
        auto mx_fallback = [&](const unsigned sp_from) {
            if constexpr (AT::kAdaptive) {
#pragma unroll 1
                for (unsigned kt = sp_from * KS; kt < sp_end * KS; ++kt) {
                    u32x4 wt[NT], af[MT][4];
                    float sc[NT];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        wt[nt] = buf_load16(w_rsrc, w_voff[nt], kt * kTileBytes, kAuxNt);
                        const unsigned sb = __builtin_amdgcn_raw_buffer_load_b8(s_rsrc, s_voff[nt] + kt % KS, (kt / KS) * 64 * kRecBytes, kAuxDefault);
                        sc[nt] = __builtin_bit_cast(float, (sb & 0xffu) << 23);
                    }
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            af[mt][j] = buf_load16(a_rsrc, a_voff[mt] + j * 16, kt * 256, kAuxDefault);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            u32x4 hi, lo;
                            split_f16(af[mt][j], hi, lo);
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                acc[mt][nt][0] = mfma16_hilo(unpack_mx(Bf16{}, wt[nt][j], sc[nt]), hi, lo, acc[mt][nt][0]);
                        }
                }
            } else {
                (void)sp_from;
            }
        };
        auto span_body = [&](const unsigned sp, auto last_c, auto fb_c) {
            constexpr bool kLast = decltype(last_c)::value;
            // kFb: this span of a block-floating-point kernel holds raw bf16 activations (see write_a_stage)
            constexpr bool kFb = decltype(fb_c)::value;
            using ATX = std::conditional_t<kFb, Bf16, AT>;
            using FragX = typename ATX::frag;
            const unsigned kt0 = sp * KS;
            if constexpr (!kLast) {
                if constexpr (AM > 0 && SL == KS)
                    issue_a_stage(sp + 1, true); // held in VGPRs until this span's fragments are read
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    srec_next[nt] = load_scale_rec<FMT, KS>(s_rsrc, s_voff[nt], (sp + 1) * 64 * kRecBytes);
            }
            static_for<0, KS>([&](auto t_c) {
                constexpr int T = decltype(t_c)::value;
                constexpr int SLOT = T % D;
                constexpr bool kRefill = !kLast || (T + D < KS);
                constexpr bool kNextA = !kLast || (T + PA < KS); // direct path: refill the ring slot
                const unsigned kt = kt0 + T;
                // activation fragments: this step's from LDS (staged path), or the next
                // step's straight from L2 (direct path)
                u32x4 anext[MT][4];
                // does the next tile live in the stage that is in LDS now?
                constexpr bool kNextInStage = (T + 1 < KS) && ((T + 1) % SL != 0);
                if constexpr (AM > 0) {
                    // staged path: prefetch the NEXT tile's fragments now, so the LDS
                    // latency hides behind this tile's unpack instead of stalling each MFMA
                    if constexpr (kNextInStage) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            anext[0][j] = (ABL & 1) ? afrag[0][j] : a_lds[a_frag_base + ((T + 1) % SL) * 16 + j];
                        }
                        __builtin_amdgcn_sched_barrier(0); // keep the reads at the top of the step
                    }
                } else {
                    // direct path: this tile's fragments come out of ring slot T % PA
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            afrag[mt][j] = aring[T % PA][mt][j];
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    float s_lo, s_hi;
                    tile_scales<FMT, KS, T>(srec[nt], s_lo, s_hi);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const unsigned w = wring[SLOT][nt][j];
                        FragX wf;
                        if constexpr (ABL & 2) {
                            const unsigned sb = __builtin_bit_cast(unsigned, j < 2 ? s_lo : s_hi);
                            wf = __builtin_bit_cast(FragX, u32x4{w, w ^ sb, w, sb});
                        } else if constexpr (FMT == kFmtNv)
                            wf = unpack_nv(ATX{}, w, j < 2 ? s_lo : s_hi);
                        else
                            wf = unpack_mx(ATX{}, w, s_lo);
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            f32x4 &a = acc[mt][nt][j % NACC];
                            if constexpr (ABL & 4) {
                                const u32x4 wb = __builtin_bit_cast(u32x4, wf), ab = afrag[mt][j];
                                a += __builtin_bit_cast(f32x4, wb ^ ab);
                            } else
                                a = mfma16(wf, __builtin_bit_cast(FragX, afrag[mt][j]), a);
                        }
                    }
                }
                if constexpr (ABL & 16) { // tile consumed (its MFMAs issued): first and last tile of the wave
                    if (T == 0 && sp == sp_begin)
                        ts[1] = __builtin_amdgcn_s_memrealtime();
                    if constexpr (kLast && T == KS - 1)
                        ts[2] = __builtin_amdgcn_s_memrealtime();
                }
                if constexpr (kRefill) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        wring[SLOT][nt] = buf_load16(w_rsrc, w_voff[nt], (kt + D) * kTileBytes, kAuxNt);
                }
                if constexpr (AM == 0 && kNextA && (ABL & 1) == 0) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            aring[T % PA][mt][j] = buf_load16(a_rsrc, a_voff[mt] + j * 16, (kt + PA) * 256, kAuxDefault);
                }
                if constexpr (AM > 0 && kNextInStage) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        afrag[0][j] = anext[0][j];
                }
                if constexpr (AM > 0 && SL < KS && (T + 1) % SL == 0 && (T + 1 < KS || !kLast)) {
                    // stage boundary inside (or at the end of) the span: every fragment of the old
                    // stage has been read (LDS is in order within a wave), so the slice takes the
                    // stage held in VGPRs, and the one after it is requested
                    write_a_stage(fb_c);
                    load_first_frags();
                    const unsigned st_next = (kt + 1) / SL + 1;
                    if constexpr (!kLast || (T + 1 + SL < KS))
                        issue_a_stage(st_next, true);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            if constexpr (AT::kBfp) { // fold this span: total += 2^sh * (span partial sums)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                    for (int q = 0; q < NACC; ++q) {
                        total[nt] += acc[0][nt][q] * bfp_up;
                        acc[0][nt][q] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
            }
            if constexpr (!kLast) {
                // every fragment of this span has been read (LDS is in order within a
                // wave): the slice can take the next span's activations
                if constexpr (AM > 0 && SL == KS) {
                    write_a_stage(fb_c);
                    load_first_frags();
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    srec[nt] = srec_next[nt];
                if constexpr (AT::kAdaptive)
                    mx_fb = mx_span_needs_fallback();
            }
        };
        if constexpr (AT::kAdaptive) {
            // fast spans first; from the first span whose record holds a scale outside fp16's safe range the exact fallback body
            mx_fb = mx_span_needs_fallback();
            unsigned sp = sp_begin;
            for (; sp + 1 < sp_end && !mx_fb; ++sp)
                span_body(sp, std::false_type{}, std::false_type{});
            if (!mx_fb)
                span_body(sp_end - 1, std::true_type{}, std::false_type{});
            else
                mx_fallback(sp);
            // the join of the two bodies: see mfma_join_settle (device_common.hpp)
            static_for<0, 2>([&](auto pass) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int q = 0; q < NACC; ++q)
                            mfma_join_pin(acc[mt][nt][q]);
                if constexpr (decltype(pass)::value == 0)
                    mfma_join_settle();
            });
        } else if constexpr (AT::kBfp) {
            // exact block-floating-point spans first; from the first span that fails the range check (its flag is set
            // by the write_a_stage that put it in LDS: the prologue's, or the previous span's) the bf16 pipeline
            unsigned sp = sp_begin;
            for (; sp + 1 < sp_end && !bfp_fb; ++sp)
                span_body(sp, std::false_type{}, std::false_type{});
            if (!bfp_fb) {
                span_body(sp_end - 1, std::true_type{}, std::false_type{});
            } else {
                for (; sp + 1 < sp_end; ++sp)
                    span_body(sp, std::false_type{}, std::true_type{});
                span_body(sp_end - 1, std::true_type{}, std::true_type{});
            }
        } else {
            for (unsigned sp = sp_begin; sp + 1 < sp_end; ++sp)
                span_body(sp, std::false_type{}, std::false_type{});
            span_body(sp_end - 1, std::true_type{}, std::false_type{});
        }
    }

    // --- cross-wave K reduction through LDS, then the epilogue -----------------
    // (gpu/quantization/reduce.cuh:7-58 + qgemm.cuh:95-192 in the reference)
    const float gs = *p.gs;
    // one float4 = rows n..n+3 of column m:  x global scale, one RNE rounding,
    // one 8-byte store (or the fp32 slab of this K part when gridDim.z > 1)
    auto emit = [&](f32x4 v, unsigned iwn, unsigned imt, unsigned inn, unsigned il) {
        const unsigned m = m0 + imt * 16 + (il & 15u);
        const unsigned ntile = (block_x * WN + iwn) * NT + inn;
        const unsigned n = ntile * 16 + (il >> 4) * 4;
        if (m >= p.m || ntile >= ntiles)
            return;
        if (gridDim.z == 1) {
            *reinterpret_cast<uint2 *>((char *)p.c + ((size_t)m * p.n + n) * 2) = finish4<AT>(v, gs, p.bias, n);
        } else {
            float *slab = p.workspace + ((size_t)blockIdx.z * p.m + m) * p.n + n;
            if constexpr (kCombine) {
#if defined(__HIP_DEVICE_COMPILE__)
                // write-through to memory (sc0 sc1): visible to the other XCDs without a write-back of this XCD's whole L2
                asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(slab), "v"(v) : "memory");
#endif
            } else {
                *reinterpret_cast<f32x4 *>(slab) = v;
            }
        }
    };
    // SiLU-mul: logical tiles (inn, inn + 1) are the gate / up halves of output tile ntile / 2 (host guarantees
    // an even NT, an even N/16 and gridDim.z == 1)
    auto emit_pair = [&](f32x4 gate, f32x4 up, unsigned iwn, unsigned imt, unsigned inn, unsigned il) {
        const unsigned m = m0 + imt * 16 + (il & 15u);
        const unsigned ntile = (block_x * WN + iwn) * NT + inn;
        if (m >= p.m || ntile >= ntiles)
            return;
        const unsigned n_half = p.n >> 1;
        const unsigned n = (ntile >> 1) * 16 + (il >> 4) * 4;
        *reinterpret_cast<uint2 *>((char *)p.c + ((size_t)m * n_half + n) * 2) =
            finish4_silu_mul<AT>(gate, up, gs, p.bias, n, n_half);
    };

    f32x4 accs[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if constexpr (AT::kBfp) {
                accs[mt][nt] = total[nt];
            } else {
                accs[mt][nt] = acc[mt][nt][0];
                if constexpr (NACC == 2)
                    accs[mt][nt] += acc[mt][nt][1];
            }
        }
    if constexpr (WK == 1) {
        if (p.act) {
            if constexpr (NT % 2 == 0) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; nt += 2)
                        emit_pair(accs[mt][nt], accs[mt][nt + 1], wn, mt, nt, lane);
            }
        } else {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    emit(accs[mt][nt], wn, mt, nt, lane);
        }
    } else {
        constexpr int kItems = Cfg::kRedItems;
        f32x4 *const red = reinterpret_cast<f32x4 *>(smem + WN * WK * Cfg::kALdsU4);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                red[wk * kItems + ((wn * MT + mt) * NT + nt) * 64 + lane] = accs[mt][nt];
        __syncthreads();
        for (unsigned item = threadIdx.x; item < (unsigned)kItems; item += Cfg::kThreads) {
            const unsigned tile = item >> 6;
            if (p.act && (tile % NT) % 2 != 0)
                continue; // the up half is consumed together with its gate tile
            f32x4 v = red[item];
#pragma unroll
            for (int q = 1; q < WK; ++q)
                v += red[q * kItems + item];
            if (p.act) {
                if constexpr (NT % 2 == 0) {
                    f32x4 u = red[item + 64];
#pragma unroll
                    for (int q = 1; q < WK; ++q)
                        u += red[q * kItems + item + 64];
                    emit_pair(v, u, tile / (MT * NT), (tile / NT) % MT, tile % NT, item & 63u);
                }
            } else {
                emit(v, tile / (MT * NT), (tile / NT) % MT, tile % NT, item & 63u);
            }
        }
    }
    if constexpr (kCombine) {
        // Every thread's slab stores are made visible device-wide (release at agent scope: the other slices run on other XCDs, whose L2s
        // are not coherent with this one), then ONE thread takes the workgroup's ticket for this output tile; the workgroup that draws the
        // last ticket acquires, re-arms the counter for the next launch and sums the slabs in slice order.
        if (gridDim.z > 1) {
            // (the slab stores above are write-through and the last arriver's loads below bypass the caches: no agent-scope fence, whose
            //  L2 write-back / invalidate per workgroup measured 15-55 us per launch -- profiles/r04_ksplit_combine.json, first variant)
            __shared__ unsigned ticket;
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            __syncthreads();
            if (threadIdx.x == 0)
                ticket = __hip_atomic_fetch_add(arg_tickets + (blockIdx.y * gridDim.x + block_x), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            if (ticket == gridDim.z - 1) {
                if (threadIdx.x == 0)
                    __hip_atomic_store(arg_tickets + (blockIdx.y * gridDim.x + block_x), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (unsigned item = threadIdx.x; item < (unsigned)Cfg::kRedItems; item += Cfg::kThreads) {
                    const unsigned tile = item >> 6, il = item & 63u;
                    const unsigned iwn = tile / (MT * NT), imt = (tile / NT) % MT, inn = tile % NT;
                    const unsigned m = m0 + imt * 16 + (il & 15u);
                    const unsigned ntile = (block_x * WN + iwn) * NT + inn;
                    const unsigned n = ntile * 16 + (il >> 4) * 4;
                    if (m >= p.m || ntile >= ntiles)
                        continue;
                    const float *src = p.workspace + (size_t)m * p.n + n;
                    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
                    for (unsigned z = 0; z < gridDim.z; ++z) {
                        f32x4 part;
#if defined(__HIP_DEVICE_COMPILE__)
                        asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(part) : "v"(src + (size_t)z * p.m * p.n) : "memory");
#endif
                        v += part;
                    }
                    *reinterpret_cast<uint2 *>((char *)p.c + ((size_t)m * p.n + n) * 2) = finish4<AT>(v, gs, p.bias, n);
                }
            }
        }
    }
    if constexpr (ABL & 16) {
        ts[3] = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) {
            unsigned long long *dst = reinterpret_cast<unsigned long long *>(p.workspace) +
                                      ((size_t)block_x * (WN * WK) + wave) * 4;
            dst[0] = ts[0], dst[1] = ts[1], dst[2] = ts[2], dst[3] = ts[3];
        }
    }
}

template <class Cfg>
__global__ __launch_bounds__(Cfg::kThreads) void gemm_stream_kernel(const void *arg_w, const void *arg_s, const void *arg_a,
                                                                    unsigned arg_k, unsigned arg_n, unsigned arg_m,
                                                                    unsigned arg_spw, unsigned arg_act, void *arg_c,
                                                                    const float *arg_gs, const void *arg_bias,
                                                                    float *arg_workspace) {
    gemm_stream_body<Cfg>(arg_w, arg_s, arg_a, arg_k, arg_n, arg_m, arg_spw, arg_act, arg_c, arg_gs, arg_bias, arg_workspace, blockIdx.x);
}

// The same kernel with the cross-workgroup K split combined in the launch (kCombine above); tickets: one zero-initialised counter per
// (m-block, column block) workgroup position, left at zero by every launch.
template <class Cfg>
__global__ __launch_bounds__(Cfg::kThreads) void gemm_stream_combine_kernel(const void *arg_w, const void *arg_s, const void *arg_a,
                                                                            unsigned arg_k, unsigned arg_n, unsigned arg_m,
                                                                            unsigned arg_spw, unsigned arg_act, void *arg_c,
                                                                            const float *arg_gs, const void *arg_bias,
                                                                            float *arg_workspace, unsigned *arg_tickets) {
    gemm_stream_body<Cfg, true>(arg_w, arg_s, arg_a, arg_k, arg_n, arg_m, arg_spw, arg_act, arg_c, arg_gs, arg_bias, arg_workspace, blockIdx.x,
                                arg_tickets);
}

// Grouped form (see gemm_decode_grouped_kernel in gemm_decode.hpp): the members' grids concatenated along x; one m-block, no K
// split across workgroups (grid y = z = 1).
template <class Cfg>
__global__ __launch_bounds__(Cfg::kThreads) void gemm_stream_grouped_kernel(const GroupTable g, const void *arg_a, unsigned arg_k, unsigned arg_m,
                                                                            unsigned arg_spw) {
    unsigned idx = 0, base = 0;
    for (unsigned i = 0; i + 1 < g.count; ++i)
        if (blockIdx.x >= g.wg_end[i])
            idx = i + 1, base = g.wg_end[i];
    gemm_stream_body<Cfg>(g.w[idx], g.s[idx], arg_a, arg_k, g.n[idx], arg_m, arg_spw, 0u, g.c[idx], g.gs[idx], g.bias[idx], nullptr,
                          blockIdx.x - base);
}

// Second pass of the cross-workgroup split-K: sum the fp32 slabs in a fixed
// order (deterministic), apply the global scale, round once.
template <class AT>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(void *c, const float *ws, const float *gs_ptr, const void *bias,
                                                            unsigned m, unsigned n, unsigned parts) {
    const size_t total4 = (size_t)m * n / 4;
    const float gs = *gs_ptr;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
         i += (size_t)gridDim.x * blockDim.x) {
        // (the slabs of a batch are requested together -- a runtime-length loop of dependent adds would wait out one L2 / MALL round trip per
        //  slab -- and summed in slab order, the fixed order of the plain loop)
        const f32x4 *const src = reinterpret_cast<const f32x4 *>(ws) + i;
        const size_t slab4 = (size_t)m * n / 4;
        f32x4 v = src[0];
        unsigned q = 1;
        for (; q + 3 < parts; q += 4) {
            const f32x4 x0 = src[(size_t)q * slab4], x1 = src[(size_t)(q + 1) * slab4], x2 = src[(size_t)(q + 2) * slab4], x3 = src[(size_t)(q + 3) * slab4];
            v += x0, v += x1, v += x2, v += x3;
        }
        for (; q < parts; ++q)
            v += src[(size_t)q * slab4];
        reinterpret_cast<uint2 *>(c)[i] = finish4<AT>(v, gs, bias, (unsigned)((i * 4) % n));
    }
}

// The same second pass with the SiLU-mul epilogue: the slabs hold the plain [m][n] product (the kernels ran with act = 0), the output is
// [m][n / 2], c[r][j] = silu(y[r][j]) * y[r][j + n/2] with y = sum * gs + bias -- the arithmetic of finish4_silu_mul, one rounding.
template <class AT>
__global__ __launch_bounds__(256) void splitk_reduce_silu_kernel(void *c, const float *ws, const float *gs_ptr, const void *bias,
                                                                 unsigned m, unsigned n, unsigned parts) {
    const unsigned n_half = n >> 1, q_per_row = n_half / 4;
    const size_t total4 = (size_t)m * q_per_row;
    const float gs = *gs_ptr;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
         i += (size_t)gridDim.x * blockDim.x) {
        const unsigned row = (unsigned)(i / q_per_row), col = (unsigned)(i % q_per_row) * 4;
        const float *const src = ws + (size_t)row * n + col;
        f32x4 gate = *reinterpret_cast<const f32x4 *>(src), up = *reinterpret_cast<const f32x4 *>(src + n_half);
        const size_t slab = (size_t)m * n;
        unsigned q = 1;
        for (; q + 1 < parts; q += 2) { // (two slabs requested together, summed in slab order)
            const f32x4 g0 = *reinterpret_cast<const f32x4 *>(src + q * slab), u0 = *reinterpret_cast<const f32x4 *>(src + q * slab + n_half);
            const f32x4 g1 = *reinterpret_cast<const f32x4 *>(src + (q + 1) * slab), u1 = *reinterpret_cast<const f32x4 *>(src + (q + 1) * slab + n_half);
            gate += g0, gate += g1, up += u0, up += u1;
        }
        for (; q < parts; ++q) {
            gate += *reinterpret_cast<const f32x4 *>(src + q * slab);
            up += *reinterpret_cast<const f32x4 *>(src + q * slab + n_half);
        }
        reinterpret_cast<uint2 *>(c)[i] = finish4_silu_mul<AT>(gate, up, gs, bias, col, n_half);
    }
}

} // namespace petit_amd
