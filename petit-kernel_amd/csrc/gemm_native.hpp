// gemm_native.hpp -- the native-FP4 path: MXFP4 weights fed RAW to the CDNA4 block-scaled MFMA
// (v_mfma_scale_f32_16x16x128_f8f6f4), activations quantised on the fly to MXFP8 (e4m3 elements,
// one e8m0 scale per 32 k).  OPT-IN: quantising 16-bit activations to e4m3 costs ~2^-4 relative
// per element, which breaks the 1e-2 parity bar of the dequant kernels, so
// these kernels are never chosen by solution_id = -1 and are only enumerated after
// petit_enable_native_fp4(1).  Their own stated tolerance and exact-semantics test:
// tests/test_gpu_parity.py::test_native_mxfp4.  NVFP4 cannot use this instruction exactly (e4m3
// group-16 scales are not E8M0 block-32 scales), so the path exists for MXFP4 only.
//
// What the packed layout buys here (layout.h): a lane's 16 bytes are 32 consecutive k of one weight
// row = exactly the FP4 operand of the instruction, and the span record byte of the tile is exactly
// its per-lane E8M0 scale -- zero unpack VALU, one MFMA per 16x16x128 block instead of four plus 48
// VALU.
//
// Operand layout of the instruction, probed on gfx950 (tools/probes/mfma_scale_probe.hip):
//   FP4 operand : lane (row = l%16, g = l/16), regs 0-3: k = 32g + 8r + nibble (natural).
//   FP8 operand : lane (col = l%16, g = l/16), regs 0-3: k = 16g .. 16g+15, regs 4-7: k = 64+16g ..
//   scales      : block b (k in [32b, 32b+32)) takes its E8M0 byte from lane group g = b of the same
//                 row / column, whichever lane holds the data bytes; opsel picks the byte of the VGPR.
//
// Two launches: quantize_act_kernel (A -> fp8 bytes in the operand's order + scales, into the
// registered workspace) and gemm_native_kernel (tiled like gemm_tiled.hpp, A tile through LDS, written there by
// direct global -> LDS loads).
#pragma once

#include "gemm_tiled.hpp"

namespace petit_amd {

typedef __attribute__((ext_vector_type(8))) int i32x8;

// Pins an accumulator at this program point (device pass only: the host pass has no "a" registers).
// hipcc 7.2 does not treat the block-scaled MFMA builtin as convergent and sinks it into the
// lane-divergent `if (m < M)` of the epilogue; there EXEC masks off the lanes of the absent rows and
// the instruction, which reads the weight rows from ALL lanes, sees garbage for them (wrong results
// in every partial m-tile).  Found with per-step register dumps; tools/probes/mfma_scale_hazard.hip
// rules out the VALU -> scale-operand hazards.
__device__ __forceinline__ void pin_acc(f32x4 &x) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+a"(x));
#endif
}

// Workspace layout, K-TILE MAJOR: qa[K/128][M][128] bytes (inside a tile: byte 32g + 16c + j holds k = 64c + 16g + j), then
// qs[K/128][M][4] E8M0 bytes -- the activation tile and the scale dwords of a workgroup and k-tile are contiguous runs, so the
// direct-to-LDS wave-loads read whole cache lines (with row-major scales qs[M][K/32] the 64 lanes of a scale load touched
// 64 different lines for 256 useful bytes; see gemm_native32.hpp for the measurement that prompted this).  Rows >= M of the
// last m-block read the next tile's rows (or zeros past the end): they only feed output columns that are never stored.
__host__ __device__ inline size_t native_ws_bytes(unsigned m, unsigned k) { return (size_t)m * k + (size_t)m * (k / 32); }
// the same rounded up to 256 B: where the fp32 slabs of a K split start inside a call's workspace
__host__ __device__ inline size_t native_ws_aligned(unsigned m, unsigned k) { return (native_ws_bytes(m, k) + 255) & ~(size_t)255; }

// One thread = 8 consecutive k of one row; the 4 threads of a quad share one 32-k block.
template <class AT>
__global__ __launch_bounds__(256) void quantize_act_kernel(const void *a, unsigned char *ws, unsigned m, unsigned k) {
    const size_t units = (size_t)m * (k / 8);
    unsigned char *qa = ws;
    unsigned char *qs = ws + (size_t)m * k;
    for (size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x; u < units; u += (size_t)gridDim.x * blockDim.x) {
        const unsigned row = (unsigned)(u / (k / 8)), c8 = (unsigned)(u % (k / 8)); // 8-element column
        const u32x4 raw = reinterpret_cast<const u32x4 *>(a)[u];
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
        // block maximum over the quad (k/8 is a multiple of 4 and rows start on quad boundaries)
        amax = fmaxf(amax, __shfl_xor(amax, 1));
        amax = fmaxf(amax, __shfl_xor(amax, 2));
        // E8M0 scale 2^(E-7): the block maximum lands in [128, 256) <= 448 (e4m3 max)
        const unsigned ebits = (__builtin_bit_cast(unsigned, amax) >> 23) & 0xffu;
        unsigned sbyte = amax == 0.f ? 127u : (ebits > 7u ? ebits - 7u : 1u);
        sbyte = sbyte > 254u ? 254u : sbyte;
        const float inv = __builtin_bit_cast(float, (254u - sbyte) << 23); // 2^-(sbyte-127)
        int q0 = 0, q1 = 0;
        q0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[0] * inv, v[1] * inv, q0, false);
        q0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[2] * inv, v[3] * inv, q0, true);
        q1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[4] * inv, v[5] * inv, q1, false);
        q1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[6] * inv, v[7] * inv, q1, true);
        // position inside the 128-k tile in the FP8 operand's order
        const unsigned kt = c8 / 16, col16 = c8 % 16;
        const unsigned off = 32 * ((col16 & 7) >> 1) + 16 * (col16 >> 3) + 8 * (col16 & 1);
        uint2 o;
        o.x = (unsigned)q0, o.y = (unsigned)q1;
        *reinterpret_cast<uint2 *>(qa + ((size_t)kt * m + row) * 128 + off) = o;
        if ((c8 & 3) == 0)
            qs[((size_t)kt * m + row) * 4 + (c8 / 4) % 4] = (unsigned char)sbyte;
    }
}

//   MT, NTW, WAVES, D as in TiledCfg.  The A tile (BM rows x 128 fp8 bytes) and its scale bytes (BM x 4) go
//   global -> LDS directly (buffer_load ... lds, see gemm_tiled.hpp): rows are 8 un-padded 16-byte units with an
//   XOR swizzle (unit u of row r at position u ^ ((r / 2) % 8): the 16 rows of a fragment read spread over all
//   banks and the two units a lane needs stay one aligned 32-byte pair), scales are one dword per row.
template <class AT_, int KS_, int MT_, int NTW_, int WAVES_, int D_> struct NativeCfg {
    using AT = AT_;
    static constexpr int KS = KS_, MT = MT_, NTW = NTW_, WAVES = WAVES_, D = D_;
    static constexpr int kThreads = 64 * WAVES;
    static constexpr int BM = 16 * MT;
    static constexpr int kDataU4 = BM * 8;                   // one tile image
    static constexpr int kScaleU4 = kThreads / 4;            // one dword per thread (rows >= BM: junk zeros)
    static constexpr int kDataLoads = BM * 8 / 64 / WAVES;   // 1 KiB wave-loads per wave per tile
    static constexpr int kMinWavesPerSimd = MT * NTW > 16 ? 1 : 2; // (64 x 320: 80 accumulator registers, one wave per SIMD)
    static_assert(KS % D == 0, "ring depth must divide the span");
    static_assert((BM * 8) % (64 * WAVES) == 0 && BM <= kThreads, "A tile must split evenly over the waves");
    static_assert((8 * WAVES) % 16 == 0, "wave-loads must step by whole 16-row groups (swizzle term constant)");
};

template <class Cfg>
__global__ __launch_bounds__(Cfg::kThreads, Cfg::kMinWavesPerSimd) void gemm_native_kernel(const GemmArgs p, const unsigned char *ws) {
    using AT = typename Cfg::AT;
    constexpr int KS = Cfg::KS, MT = Cfg::MT, NTW = Cfg::NTW, WAVES = Cfg::WAVES, D = Cfg::D;
    constexpr unsigned kRecBytes = ScaleRec<kFmtMx, KS>::kBytes;
    constexpr unsigned kOob = 0x80000000u;

    // [tile image 0][tile image 1][scale dwords 0][scale dwords 1]
    __shared__ u32x4 smem[2 * Cfg::kDataU4 + 2 * Cfg::kScaleU4];

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
    // K slice of this workgroup (gridDim.z > 1: see gemm_tiled.hpp)
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
    const unsigned s_row_bytes = p.k / 2;
    // logical -> physical n-tiles (identity, or gate/up pairs for the SiLU-mul epilogue; device_common.hpp)
    const unsigned pt0 = valid_nt ? physical_tile(nt0, ntiles, p.act) : 0u;
    const unsigned span_tiles = !valid_nt ? 0u : p.act ? (valid_nt >> 1) + (ntiles >> 1) : valid_nt;
    const __amdgpu_buffer_rsrc_t w_rsrc =
        make_rsrc((const char *)p.w + (size_t)pt0 * w_row_bytes, span_tiles * w_row_bytes);
    const __amdgpu_buffer_rsrc_t s_rsrc =
        make_rsrc((const char *)p.s + (size_t)pt0 * s_row_bytes, span_tiles * s_row_bytes);
    // tile nt of this wave, relative to pt0 (scalar: rides in the SGPR offset of every load)
    auto rel_tile = [&](int nt) -> unsigned {
        return ((unsigned)nt < valid_nt) ? physical_tile(nt0 + nt, ntiles, p.act) - pt0 : span_tiles; // beyond: out of range
    };
    // quantised activations and their scales, k-tile major: tile kt of this m-block starts (kt M + m0) rows of 128 (data) /
    // 4 (scales) bytes into its region; the descriptors end with the region
    const size_t qa_bytes = (size_t)p.m * p.k, qs_bytes = (size_t)p.m * (p.k / 32);
    const __amdgpu_buffer_rsrc_t qa_rsrc = make_rsrc(ws + (size_t)m0 * 128, (unsigned)(qa_bytes - (size_t)m0 * 128));
    const __amdgpu_buffer_rsrc_t qs_rsrc = make_rsrc(ws + qa_bytes + (size_t)m0 * 4, (unsigned)(qs_bytes - (size_t)m0 * 4));
    const unsigned qa_tile = p.m * 128, qs_tile = p.m * 4; // bytes from one k-tile to the next

    // Per-lane offsets are ONE VGPR each; everything that varies with the n-tile, the k-step or the
    // staging unit rides in the SGPR offset, which gfx950 includes in the buffer range check (probed,
    // tools/probes): n-tiles beyond valid_nt and rows beyond M still read as zeros.  (The streaming
    // kernel keeps validity in the VGPR offset; here 12 VGPRs decide between one and two waves per SIMD.)
    const unsigned w_voff = lane * 16, s_voff = lane * kRecBytes;
    // direct-to-LDS staging.  Data: wave-load i of this wave covers rows 8*(i*WAVES + wave) .. +7, lane l -> row + l/8,
    // position l%8, which receives unit (l%8) ^ ((row/2)%8).  Scales: wave w covers rows 64w .. 64w+63, one dword
    // per lane (waves beyond BM read out of range and park zeros in the unused tail of the scale array).
    const unsigned dma_row0 = wave * 8 + (lane >> 3);
    const unsigned dma_voff = dma_row0 * 128 + (((lane & 7u) ^ ((dma_row0 >> 1) & 7u)) * 16);
    const unsigned qs_voff = (wave * 64 < (unsigned)Cfg::BM) ? (wave * 64 + lane) * 4 : kOob;
    auto dma_stage = [&](unsigned kt, unsigned buf) {
#if defined(__HIP_DEVICE_COMPILE__) // (the host pass knows neither the builtin nor the LDS address space)
        u32x4 *const data = smem + buf * Cfg::kDataU4;
#pragma unroll
        for (int i = 0; i < Cfg::kDataLoads; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(qa_rsrc, (__attribute__((address_space(3))) void *)(data + (i * WAVES + wave) * 64), 16,
                                                     dma_voff, i * (8 * WAVES) * 128 + kt * qa_tile, 0, 0);
        u32x4 *const sc = smem + 2 * Cfg::kDataU4 + buf * Cfg::kScaleU4;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(qs_rsrc, (__attribute__((address_space(3))) void *)(sc + wave * 16), 4, qs_voff, kt * qs_tile, 0, 0);
#else
        (void)kt, (void)buf;
#endif
    };
    const unsigned swz = (r >> 1) & 7u;
    const unsigned a_frag_lo = r * 8 + ((2 * g) ^ swz);      // + mt*16*8; the second half is the pair's other unit
    const unsigned a_frag_hi = r * 8 + ((2 * g + 1) ^ swz);
    const unsigned a_scale_byte = r * 4 + g;                 // + mt*16*4, byte index into the scale array

    dma_stage(kt_begin, 0); // (kt_begin is even)
    ScaleRec<kFmtMx, KS> srec[NTW], srec_next[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
        srec[nt] = load_scale_rec<kFmtMx, KS>(s_rsrc, s_voff, rel_tile(nt) * s_row_bytes + sp_begin * 64 * kRecBytes);
    u32x4 wring[D][NTW];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
            wring[i][nt] = buf_load16(w_rsrc, w_voff, rel_tile(nt) * w_row_bytes + (kt_begin + i) * kTileBytes, kAuxDefault);
    __syncthreads();

    auto span_body = [&](const unsigned sp, auto last_c) {
        constexpr bool kLast = decltype(last_c)::value;
        const unsigned kt0 = sp * KS;
        static_for<0, KS>([&](auto t_c) {
            constexpr int T = decltype(t_c)::value;
            constexpr int SLOT = T % D;
            constexpr bool kRefill = !kLast || (T + D < KS);
            constexpr bool kNextA = !kLast || (T + 1 < KS);
            const unsigned kt = kt0 + T;
            const unsigned cur = kt & 1u;
            const u32x4 *const a_cur = smem + cur * Cfg::kDataU4;
            const unsigned char *const a_cur_bytes = reinterpret_cast<const unsigned char *>(smem + 2 * Cfg::kDataU4 + cur * Cfg::kScaleU4);
            if constexpr (kNextA) {
                dma_stage(kt + 1, cur ^ 1u); // everybody left that buffer at the barrier that ended the previous step
                __builtin_amdgcn_sched_barrier(0); // (issued BEFORE this step's W refill: the wait that ends the step counts on it)
            }
            if constexpr (!kLast && T == KS - 1) { // next span's scale records: one step ahead is enough
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
                    srec_next[nt] = load_scale_rec<kFmtMx, KS>(s_rsrc, s_voff, rel_tile(nt) * s_row_bytes + (sp + 1) * 64 * kRecBytes);
            }
            // weights: the lane's uint4 IS the FP4 operand; its scale is byte T of the span record
            i32x8 wop[NTW];
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const u32x4 w = wring[SLOT][nt];
                wop[nt] = i32x8{(int)w[0], (int)w[1], (int)w[2], (int)w[3], 0, 0, 0, 0};
            }
            if constexpr (kRefill) {
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
                    wring[SLOT][nt] = buf_load16(w_rsrc, w_voff, rel_tile(nt) * w_row_bytes + (kt + D) * kTileBytes, kAuxDefault);
            }
            // the step's global loads are requested before anything else (hipcc otherwise sinks them
            // next to their LDS stores at the end of the step and exposes the whole HBM/L2 latency).
            // Not with 128 accumulator registers: holding the loads across the step (and the second
            // fragment set below) pushes the kernel past 256 VGPRs = one wave per SIMD, measured
            // 6-8 % slower at M = 2048 than letting the second wave hide the latency.
            constexpr bool kPrefetch = true;
            if constexpr (kPrefetch)
                __builtin_amdgcn_sched_barrier(0);
            if constexpr (kPrefetch) {
                // activation fragments: m-tile mt+1 is read from LDS while the MFMAs of m-tile mt run
                u32x4 flo[2], fhi[2];
                int fsc[2];
                auto read_frag = [&](int mt, int slot) {
                    flo[slot] = a_cur[a_frag_lo + mt * 16 * 8];
                    fhi[slot] = a_cur[a_frag_hi + mt * 16 * 8];
                    fsc[slot] = (int)a_cur_bytes[a_scale_byte + mt * 16 * 4];
                };
                read_frag(0, 0);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    if (mt + 1 < MT)
                        read_frag(mt + 1, (mt + 1) & 1);
                    const u32x4 lo = flo[mt & 1], hi = fhi[mt & 1];
                    const int ascale = fsc[mt & 1];
                    const i32x8 aop = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
                            wop[nt], aop, acc[mt][nt], 4 /* A = FP4 */, 0 /* B = FP8 e4m3 */, T % 4,
                            (int)srec[nt].d[T / 4], 0, ascale);
                    __builtin_amdgcn_sched_barrier(0); // keep the prefetch one m-tile ahead, no further
                }
            } else {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const u32x4 lo = a_cur[a_frag_lo + mt * 16 * 8];
                    const u32x4 hi = a_cur[a_frag_hi + mt * 16 * 8];
                    const int ascale = (int)a_cur_bytes[a_scale_byte + mt * 16 * 4];
                    const i32x8 aop = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
                            wop[nt], aop, acc[mt][nt], 4 /* A = FP4 */, 0 /* B = FP8 e4m3 */, T % 4,
                            (int)srec[nt].d[T / 4], 0, ascale);
                }
            }
            // The next activation tile must have landed -- nothing else.  Loads retire in issue order and the last loads of the
            // step are its W refill, which stays in flight: __syncthreads() (= vmcnt(0)) drained the refill at every step, so
            // the weight ring covered ONE step whatever its depth, and a step (16 MFMAs, ~0.25 us) took an L2 round trip
            // (~0.5 us: 64 steps x 7 rounds x 0.5 us = the 223 us measured at gate_up, M = 512).
            if constexpr (kNextA) {
#if defined(__HIP_DEVICE_COMPILE__)
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(kRefill ? NTW : 0) : "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
#endif
            }
            // Pin the step: left alone, LLVM sinks all MT*NTW*KS MFMAs of a span below its last barrier
            // and hoists every step's LDS traffic above them (hundreds of spilled VGPRs, no overlap).
            __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (!kLast) {
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
                srec[nt] = srec_next[nt];
        }
    };
    for (unsigned sp = sp_begin; sp + 1 < sp_end; ++sp)
        span_body(sp, std::false_type{});
    span_body(sp_end - 1, std::true_type{});

#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
            pin_acc(acc[mt][nt]); // every MFMA executes with all 64 lanes, before the divergent stores

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
