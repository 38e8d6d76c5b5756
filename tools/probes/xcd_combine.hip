// tools/probes/xcd_combine.hip -- VERDICT r05 item 6: what does a K-split combine INSIDE a launch cost when the partners sit on CUs of ONE XCD (they share an L2)?
// Every open item between M = 8 and M = 256 names "a combine of K slices cheaper than a launch"; round 4's ticketed slab read (agent scope: the partials travel
// through memory, r04_ksplit_combine.json) cost what a second launch costs.  Here: 256 workgroups (one per CU), groups of G = 2 / 4 workgroups own one C tile of
// `tile_f4` float4 (64 x 256 f32 = 4096 float4 = 64 KB); every workgroup produces a partial tile in registers, the non-last arrivers park theirs in a slab, the LAST
// arriver of the group (a ticket in L2) sums the slab(s) into its registers and writes the bf16 tile.  Variants:
//   partners   0 = same XCD (block ids b, b + 8, b + 16, ...: workgroups are dispatched round-robin over the 8 XCDs -- checked here by reading HW_REG_XCC_ID)
//              1 = neighbouring block ids (eight different XCDs)
//   scope      0 = workgroup-scope accesses (sc0: served by the XCD's L2, never leaves it) -- only meaningful for partners = 0
//              1 = agent-scope accesses (sc1: through memory / MALL), the only correct choice for partners = 1
//   baseline   no exchange at all: every workgroup writes its own (partial) tile -- what the same grid costs without the combine
// The probe returns the XCC id every block ran on, so that the "b % 8" assumption is MEASURED, not assumed.
//   hipcc -O3 -std=c++20 -shared -fPIC --offload-arch=gfx950 tools/probes/xcd_combine.hip -o tools/probes/libxcdcombine.so
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int SCOPE> __device__ __forceinline__ void st4(f32x4 *p, f32x4 v) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (SCOPE == 0)
        asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
    else
        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
#endif
}
template <int SCOPE> __device__ __forceinline__ f32x4 ld4(const f32x4 *p) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (SCOPE == 0)
        asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
    else
        asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
#endif
    return v;
}

// PER = float4 per thread (tile_f4 = 256 * PER)
template <int SCOPE, int PER>
__global__ __launch_bounds__(256) void combine_kernel(f32x4 *slabs, unsigned *tickets, uint2 *out, unsigned *xcc_of_block, int G, int partners, int baseline) {
    const unsigned b = blockIdx.x, tid = threadIdx.x;
    unsigned tile, rank;
    if (partners == 0) { // same XCD: blocks b, b + 8, ..., b + 8 (G - 1)
        tile = (b / (8 * G)) * 8 + (b % 8), rank = (b / 8) % G;
    } else {
        tile = b / G, rank = b % G;
    }
#if defined(__HIP_DEVICE_COMPILE__)
    if (tid == 0 && xcc_of_block)
        xcc_of_block[b] = __builtin_amdgcn_s_getreg(20 /* HW_REG_XCC_ID */ | (0 << 6) | ((4 - 1) << 11));
#endif
    f32x4 acc[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const float x = (float)((tid * PER + i) % 97) * 0.25f + (float)rank;
        acc[i] = f32x4{x, x + 1.f, x + 2.f, x + 3.f};
    }
    const unsigned tile_f4 = 256 * PER;
    if (!baseline) {
        __shared__ unsigned arrival;
        // park the partial (slot `rank` of the tile's slab), then take a ticket
#pragma unroll
        for (int i = 0; i < PER; ++i)
            st4<SCOPE>(slabs + ((size_t)tile * G + rank) * tile_f4 + i * 256 + tid, acc[i]);
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        __syncthreads();
        if (tid == 0) {
            if constexpr (SCOPE == 0)
                arrival = __hip_atomic_fetch_add(tickets + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else
                arrival = __hip_atomic_fetch_add(tickets + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (arrival != (unsigned)(G - 1))
            return;
        if (tid == 0)
            tickets[tile] = 0; // (the next launch starts from zero; stream order makes it visible)
        for (int r = 0; r < G; ++r) {
            if ((unsigned)r == rank)
                continue;
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const f32x4 v = ld4<SCOPE>(slabs + ((size_t)tile * G + r) * tile_f4 + i * 256 + tid);
#if defined(__HIP_DEVICE_COMPILE__)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
                acc[i] += v;
            }
        }
    }
    const size_t obase = baseline ? (size_t)b * tile_f4 : (size_t)tile * tile_f4;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
        typedef __attribute__((ext_vector_type(2))) float f32x2;
        const bf16x2 lo = __builtin_convertvector(f32x2{acc[i][0], acc[i][1]}, bf16x2), hi = __builtin_convertvector(f32x2{acc[i][2], acc[i][3]}, bf16x2);
        uint2 o;
        o.x = __builtin_bit_cast(unsigned, lo), o.y = __builtin_bit_cast(unsigned, hi);
        out[obase + i * 256 + tid] = o;
    }
}

// per = float4 per thread: 4 (16 KB tile = 16 x 256 f32), 16 (64 KB = 64 x 256 f32), 32 (128 KB)
extern "C" int xc_launch(int scope, int per, void *slabs, void *tickets, void *out, void *xcc, int blocks, int G, int partners, int baseline, void *stream) {
    hipStream_t st = (hipStream_t)stream;
#define GO(S, P) hipLaunchKernelGGL((combine_kernel<S, P>), dim3(blocks), dim3(256), 0, st, (f32x4 *)slabs, (unsigned *)tickets, (uint2 *)out, (unsigned *)xcc, G, partners, baseline)
    if (scope == 0 && per == 4) GO(0, 4);
    else if (scope == 0 && per == 16) GO(0, 16);
    else if (scope == 0 && per == 32) GO(0, 32);
    else if (scope == 1 && per == 4) GO(1, 4);
    else if (scope == 1 && per == 16) GO(1, 16);
    else if (scope == 1 && per == 32) GO(1, 32);
    else return 2;
#undef GO
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
