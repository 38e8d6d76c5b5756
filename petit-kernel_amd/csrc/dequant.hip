// dequant.hip -- stand-alone dense dequantisation of PACKED weights (debug / test aid).
//
// The reference keeps GPU kernels that expand its packed format back to a dense 16-bit matrix for its unit tests
// (DequantPetitFp4 / DequantPetitMxFp4, fp4/quantization_utils.cu:542-727, used by quantization_utils_fp4_test.cc).  The
// GEMM kernels here never materialise the dense matrix; this op does, through the SAME hardware converts and the same
// scale decode the kernels use (device_common.hpp), so "what do the kernels think this packed tensor holds" can be
// answered without a GEMM:  out[n][k] = fp4(w[n][k]) * scale[n][k / g] * global_scale,  f32 / bf16 / fp16, row-major.
#include <hip/hip_runtime.h>

#include "device_common.hpp"

namespace petit_amd {

// One thread = one packed uint4 = 32 consecutive k of one weight row (layout.h): 32 outputs.
template <int FMT, int OUT> // OUT: 0 f32, 1 bf16, 2 fp16
__global__ __launch_bounds__(256) void dequant_packed_kernel(void *out, const uint4 *w, const unsigned char *s, float gs, unsigned n, unsigned k) {
    const size_t total = (size_t)n * k / 32;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
        const unsigned lane = (unsigned)(o % 64);
        const size_t tile = o / 64;
        const unsigned kt = (unsigned)(tile % (k / kTileK)), nt = (unsigned)(tile / (k / kTileK));
        const unsigned r = lane % 16, g = lane / 16;
        const unsigned row = nt * 16 + r, k0 = kt * kTileK + g * kLaneK;
        const uint4 q = w[o];
        const unsigned words[4] = {q.x, q.y, q.z, q.w};
        float v[32];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float sc;
            if constexpr (FMT == kFmtNv) {
                const unsigned byte = s[packed_nvscale_byte_index(k, row, (k0 + 8 * j) / kNvGroup)];
                sc = e4m3_byte<0>(byte);
            } else {
                const unsigned byte = s[packed_mxscale_byte_index(k, row, k0 / kMxGroup)];
                sc = e8m0_byte<0>(byte);
            }
            const f32x2 p0 = cvt_fp4_f32<0>(words[j], 1.0f), p1 = cvt_fp4_f32<1>(words[j], 1.0f);
            const f32x2 p2 = cvt_fp4_f32<2>(words[j], 1.0f), p3 = cvt_fp4_f32<3>(words[j], 1.0f);
            const float e[8] = {p0.x, p0.y, p1.x, p1.y, p2.x, p2.y, p3.x, p3.y};
#pragma unroll
            for (int i = 0; i < 8; ++i)
                v[8 * j + i] = e[i] * sc * gs;
        }
        const size_t base = (size_t)row * k + k0;
        if constexpr (OUT == 0) {
            float4 *dst = reinterpret_cast<float4 *>((float *)out + base);
#pragma unroll
            for (int i = 0; i < 8; ++i)
                dst[i] = float4{v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]};
        } else {
            uint4 *dst = reinterpret_cast<uint4 *>((unsigned short *)out + base);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint4 pk;
                if constexpr (OUT == 1) {
                    pk.x = pack2(Bf16{}, v[8 * i], v[8 * i + 1]), pk.y = pack2(Bf16{}, v[8 * i + 2], v[8 * i + 3]);
                    pk.z = pack2(Bf16{}, v[8 * i + 4], v[8 * i + 5]), pk.w = pack2(Bf16{}, v[8 * i + 6], v[8 * i + 7]);
                } else {
                    pk.x = pack2(Fp16{}, v[8 * i], v[8 * i + 1]), pk.y = pack2(Fp16{}, v[8 * i + 2], v[8 * i + 3]);
                    pk.z = pack2(Fp16{}, v[8 * i + 4], v[8 * i + 5]), pk.w = pack2(Fp16{}, v[8 * i + 6], v[8 * i + 7]);
                }
                dst[i] = pk;
            }
        }
    }
}

int dequant_packed(void *out, const void *w, const void *s, float gs, unsigned n, unsigned k, int b_type, int out_kind, hipStream_t stream) {
    if (n == 0 || k == 0)
        return kOk;
    if (n % kTileN || k % 256)
        return kErrProblemShape;
    const bool mx = b_type == kDataTypeMxFp4e2m1;
    if ((!mx && b_type != kDataTypeFp4e2m1) || out_kind < 0 || out_kind > 2)
        return kErrBadArgument;
    const size_t items = (size_t)n * k / 32;
    size_t blocks = (items + 255) / 256;
    if (blocks > 4096)
        blocks = 4096;
    const dim3 grid((unsigned)blocks), block(256);
    const uint4 *w4 = (const uint4 *)w;
    const unsigned char *s1 = (const unsigned char *)s;
#define PETIT_DQ(F, O) hipLaunchKernelGGL((dequant_packed_kernel<F, O>), grid, block, 0, stream, out, w4, s1, gs, n, k)
    if (!mx && out_kind == 0) PETIT_DQ(kFmtNv, 0);
    if (!mx && out_kind == 1) PETIT_DQ(kFmtNv, 1);
    if (!mx && out_kind == 2) PETIT_DQ(kFmtNv, 2);
    if (mx && out_kind == 0) PETIT_DQ(kFmtMx, 0);
    if (mx && out_kind == 1) PETIT_DQ(kFmtMx, 1);
    if (mx && out_kind == 2) PETIT_DQ(kFmtMx, 2);
#undef PETIT_DQ
    return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

} // namespace petit_amd
