// tools/probes/ksplit_combine.hip -- VERDICT r03 item 5: what does a K split across workgroups cost at M = 8 / 16 when its partial sums are
// combined INSIDE the launch (slab + ticket, last arriver sums in slice order: gemm_stream_body<Cfg, true>) instead of in a second launch?
// The streaming kernel of the product (csrc/gemm_stream.hpp), bf16 x NVFP4, staged 16-row activations, in three forms:
//   variant 0  no K split                                  (what solution_id = -1 runs on `o` at M = 16)
//   variant 1  K split across gridDim.z + splitk_reduce_kernel   (two launches: what the library does today when a row asks for it)
//   variant 2  K split across gridDim.z, combined in the launch
// for column blocks of 32 / 64 / 128 (NT = 2 / 4 with WN = 1, and WN = 2 x NT = 4): a split of 2 with twice the column block keeps the
// grid at the unsplit kernel's size while every CU pulls in HALF of the activations (the activation ingest is what
// separates M = 16 from M = 1).
//   hipcc -O3 -std=c++20 -shared -fPIC --offload-arch=gfx950 -Ipetit-kernel_amd/csrc tools/probes/ksplit_combine.hip -o tools/probes/libksplitcombine.so
#include "gemm_stream.hpp"

using namespace petit_amd;

template <class Cfg>
static int launch(int variant, const void *w, const void *s, const void *a, void *c, const float *gs, float *ws, unsigned *tickets, unsigned m,
                  unsigned n, unsigned k, unsigned splitk, hipStream_t stream) {
    const unsigned ntiles = n / kTileN, per_wg = Cfg::WN * Cfg::NT;
    const unsigned nspans = k / (kTileK * Cfg::KS), kparts = splitk * Cfg::WK;
    const unsigned spw = (nspans + kparts - 1) / kparts;
    const dim3 grid((ntiles + per_wg - 1) / per_wg, 1, splitk);
    if (variant == 2)
        hipLaunchKernelGGL(gemm_stream_combine_kernel<Cfg>, grid, dim3(Cfg::kThreads), 0, stream, w, s, a, k, n, m, spw, 0u, c, gs, nullptr, ws, tickets);
    else
        hipLaunchKernelGGL(gemm_stream_kernel<Cfg>, grid, dim3(Cfg::kThreads), 0, stream, w, s, a, k, n, m, spw, 0u, c, gs, nullptr, ws);
    if (variant == 1 && splitk > 1) {
        const size_t total4 = (size_t)m * n / 4;
        hipLaunchKernelGGL(splitk_reduce_kernel<Bf16>, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, stream, c, ws, gs, nullptr, m, n, splitk);
    }
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// shape: 0 = 16x32 (NT 2), 1 = 16x64 (NT 4), 2 = 16x128 (WN 2 x NT 4); all WK = 4, staged AM = 16
extern "C" int ksc_launch(int shape, int variant, const void *w, const void *s, const void *a, void *c, const float *gs, float *ws,
                          unsigned *tickets, unsigned m, unsigned n, unsigned k, unsigned splitk, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    switch (shape) {
    case 0: return launch<StreamCfg<Bf16, kFmtNv, 8, 1, 2, 1, 4, 4, 16>>(variant, w, s, a, c, gs, ws, tickets, m, n, k, splitk, st);
    case 1: return launch<StreamCfg<Bf16, kFmtNv, 8, 1, 4, 1, 4, 2, 16>>(variant, w, s, a, c, gs, ws, tickets, m, n, k, splitk, st);
    case 2: return launch<StreamCfg<Bf16, kFmtNv, 8, 1, 4, 2, 4, 2, 16>>(variant, w, s, a, c, gs, ws, tickets, m, n, k, splitk, st);
    }
    return 2;
}
