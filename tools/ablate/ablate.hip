// ablate.hip -- measurement-only build of the streaming kernel with pieces removed
// (StreamCfg::ABL), to find out where the time of a launch goes.  Not shipped.
#include "../../petit-kernel_amd/csrc/gemm_stream.hpp"
using namespace petit_amd;

template <class AT, int MT, int NT, int WN, int WK, int D, int AM, int ABL> static void launch(const GemmArgs &a, hipStream_t st) {
    using Cfg = StreamCfg<AT, kFmtNv, 8, MT, NT, WN, WK, D, AM, ABL>;
    const unsigned ntiles = a.n / 16, per_wg = WN * NT;
    dim3 grid((ntiles + per_wg - 1) / per_wg, (a.m + 16 * MT - 1) / (16 * MT), 1);
    GemmArgs b = a;
    b.spans_per_wave = (a.k / 1024 + WK - 1) / WK;
    hipLaunchKernelGGL(gemm_stream_kernel<Cfg>, grid, dim3(Cfg::kThreads), 0, st, b.w, b.s, b.a, b.k, b.n, b.m, b.spans_per_wave,
                       b.act, b.c, b.gs, b.bias, b.workspace);
}

// variant: (AT,MT,NT,WN,WK,D,AM) 0 = bfp (1,1,1,8,8,1)  1 = bf16 (1,1,1,8,8,1)  2 = bf16 (1,2,1,4,4,16)  3 = bf16 (1,2,1,8,4,0)
//          4 = (1,2,1,4,8,16) deeper ring  5 = (1,1,1,8,8,16) the M = 1 wave geometry  6 = (1,2,1,8,4,16) eight K waves
extern "C" int ablate_launch(int variant, int abl, void *c, const void *a, const void *w, const void *s, const float *gs,
                             unsigned m, unsigned n, unsigned k, void *stream, void *stamps) {
    GemmArgs g{};
    g.workspace = (float *)stamps;
    g.c = c, g.a = a, g.w = w, g.s = s, g.gs = gs, g.m = m, g.n = n, g.k = k;
    hipStream_t st = (hipStream_t)stream;
#define CASE(V, AT, MT, NT, WN, WK, D, AM)                                 \
    if (variant == V) {                                               \
        switch (abl) {                                                \
        case 0: launch<AT, MT, NT, WN, WK, D, AM, 0>(g, st); return 0;        \
        case 1: launch<AT, MT, NT, WN, WK, D, AM, 1>(g, st); return 0;        \
        case 2: launch<AT, MT, NT, WN, WK, D, AM, 2>(g, st); return 0;        \
        case 3: launch<AT, MT, NT, WN, WK, D, AM, 3>(g, st); return 0;        \
        case 4: launch<AT, MT, NT, WN, WK, D, AM, 4>(g, st); return 0;        \
        case 5: launch<AT, MT, NT, WN, WK, D, AM, 5>(g, st); return 0;        \
        case 7: launch<AT, MT, NT, WN, WK, D, AM, 7>(g, st); return 0;        \
        case 8: launch<AT, MT, NT, WN, WK, D, AM, 8>(g, st); return 0;        \
        case 16: launch<AT, MT, NT, WN, WK, D, AM, 16>(g, st); return 0;      \
        case 32: launch<AT, MT, NT, WN, WK, D, AM, 32>(g, st); return 0;      \
        default: return -1;                                           \
        }                                                             \
    }
    CASE(0, Bf16Bfp, 1, 1, 1, 8, 8, 1)
    CASE(1, Bf16, 1, 1, 1, 8, 8, 1)
    CASE(2, Bf16, 1, 2, 1, 4, 4, 16)
    CASE(3, Bf16, 1, 2, 1, 8, 4, 0)
    CASE(4, Bf16, 1, 2, 1, 4, 8, 16)
    CASE(5, Bf16, 1, 1, 1, 8, 8, 16)
    CASE(6, Bf16, 1, 2, 1, 8, 4, 16)
    return -1;
}
