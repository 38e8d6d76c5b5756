// mfma32_fp6_probe.hip -- what the native FP4 x FP6 class (csrc/gemm_native32.hpp, ACT = 6) relies on, probed on gfx950:
//   1. v_cvt_scalef32_pk32_fp6_{bf16,f16}: element i of the 32 inputs lands in bits [6 i, 6 i + 6) of the 6 result dwords, as
//      RNE(src / scale) in E2M3 (sign, 2 exponent bits, 3 mantissa bits; bias 1; subnormal step 1/8; max 7.5, saturating);
//   2. v_mfma_scale_f32_32x32x64_f8f6f4 with A = FP4 (cbsz 4), B = FP6 E2M3 (blgp 2): lane (col = l % 32, h = l / 32) holds
//      k = 32 h .. 32 h + 31 in element order (the FP4 operand's "natural" layout), registers 0-5 of the operand;
//   3. the E8M0 scale byte of that block comes from the same lane (as for FP4 / FP8).
// Standalone: hipcc -O2 --offload-arch=gfx950 mfma32_fp6_probe.hip -o mfma32_fp6_probe && ./mfma32_fp6_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(32))) __bf16 bf16x32;
typedef __attribute__((ext_vector_type(32))) _Float16 f16x32;
typedef __attribute__((ext_vector_type(6))) unsigned u32x6;

static float e2m3(int c) {
    const int e = (c >> 3) & 3, m = c & 7;
    const float v = e ? ldexpf(1.f + m / 8.f, e - 1) : m / 8.f;
    return (c & 32) ? -v : v;
}
static int e2m3_rne(float x) { // nearest code, ties to even mantissa, saturating
    const float ax = fminf(fabsf(x), 7.5f);
    int best = 0;
    float bd = 1e30f;
    for (int c = 0; c < 32; ++c) {
        const float d = fabsf(e2m3(c) - ax);
        if (d < bd || (d == bd && (c & 1) == 0))
            bd = d, best = c;
    }
    return best | (signbit(x) ? 32 : 0);
}

__global__ void cvt_kernel(const float *src, float scale, unsigned *out_bf16, unsigned *out_f16) {
    const int l = threadIdx.x;
    bf16x32 vb;
    f16x32 vh;
    for (int i = 0; i < 32; ++i) {
        vb[i] = (__bf16)src[l * 32 + i];
        vh[i] = (_Float16)src[l * 32 + i];
    }
    const u32x6 a = __builtin_amdgcn_cvt_scalef32_pk32_fp6_bf16(vb, scale);
    const u32x6 b = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(vh, scale);
    for (int r = 0; r < 6; ++r)
        out_bf16[l * 6 + r] = a[r], out_f16[l * 6 + r] = b[r];
}

typedef __attribute__((ext_vector_type(16))) float f32x16v;
__global__ void cvt2x_kernel(const float *src, float scale, unsigned *out) {
    const int l = threadIdx.x;
    f32x16v a, b;
    for (int i = 0; i < 16; ++i)
        a[i] = src[l * 32 + i], b[i] = src[l * 32 + 16 + i];
    const u32x6 r = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a, b, scale);
    for (int j = 0; j < 6; ++j)
        out[l * 6 + j] = r[j];
}

__global__ void mfma_kernel(const int *a_words, const int *b_words, const int *sa, const int *sb, float *out) {
    const int l = threadIdx.x;
    i32x8 a, b;
    for (int r = 0; r < 8; ++r)
        a[r] = a_words[l * 8 + r], b[r] = b_words[l * 8 + r];
    f32x16 c = {};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4 /* A = FP4 */, 2 /* B = FP6 E2M3 */, 0, sa[l], 0, sb[l]);
    for (int i = 0; i < 16; ++i)
        out[l * 16 + i] = c[i];
}

static void put6(std::vector<int> &words, int lane, int i, int code) { // element i of the lane's operand: bits [6 i, 6 i + 6)
    const int bit = 6 * i;
    uint64_t v = (uint64_t)(code & 63) << (bit % 32);
    words[lane * 8 + bit / 32] |= (int)(uint32_t)v;
    if (bit % 32 > 26)
        words[lane * 8 + bit / 32 + 1] |= (int)(uint32_t)(v >> 32);
}

int main() {
    int fails = 0;
    // ---- 1. the conversion
    {
        std::vector<float> src(64 * 32);
        for (int l = 0; l < 64; ++l)
            for (int i = 0; i < 32; ++i) {
                const int c = (l * 7 + i * 3) % 64;
                // lanes 0-31: exactly representable values; lanes 32-63: in-between values and values beyond the range (rounding, saturation)
                src[l * 32 + i] = l < 32 ? e2m3(c) : e2m3(c) * 1.06f + ((i & 1) ? 0.03f : -0.02f) + (i == 5 ? 20.f : 0.f);
            }
        float *dsrc;
        unsigned *d1, *d2;
        hipMalloc(&dsrc, src.size() * 4), hipMalloc(&d1, 64 * 6 * 4), hipMalloc(&d2, 64 * 6 * 4);
        for (float scale : {1.0f, 0.25f, 4.0f}) {
            hipMemcpy(dsrc, src.data(), src.size() * 4, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(cvt_kernel, dim3(1), dim3(64), 0, 0, dsrc, scale, d1, d2);
            std::vector<unsigned> o1(64 * 6), o2(64 * 6);
            hipMemcpy(o1.data(), d1, o1.size() * 4, hipMemcpyDeviceToHost), hipMemcpy(o2.data(), d2, o2.size() * 4, hipMemcpyDeviceToHost);
            int bad1 = 0, bad2 = 0;
            for (int l = 0; l < 64; ++l)
                for (int i = 0; i < 32; ++i) {
                    auto get = [&](const std::vector<unsigned> &o) {
                        const int bit = 6 * i;
                        uint64_t w = o[l * 6 + bit / 32];
                        if (bit / 32 + 1 < 6)
                            w |= (uint64_t)o[l * 6 + bit / 32 + 1] << 32;
                        return (int)((w >> (bit % 32)) & 63);
                    };
                    // the hardware rounds the 16-bit input: emulate on what it saw
                    const float xb = (float)(__bf16)src[l * 32 + i] / scale, xh = (float)(_Float16)src[l * 32 + i] / scale;
                    const int wb = e2m3_rne(xb), wh = e2m3_rne(xh);
                    const int gb = get(o1), gh = get(o2);
                    if (e2m3(gb) != e2m3(wb) && bad1++ < 5)
                        printf("  cvt bf16 scale %g lane %d elem %d: x %g got code %d (%g) want %d (%g)\n", scale, l, i, xb, gb, e2m3(gb), wb, e2m3(wb));
                    if (e2m3(gh) != e2m3(wh) && bad2++ < 5)
                        printf("  cvt f16 scale %g lane %d elem %d: x %g got code %d (%g) want %d (%g)\n", scale, l, i, xh, gh, e2m3(gh), wh, e2m3(wh));
                }
            printf("cvt_scalef32_pk32_fp6 scale %g: bf16 mismatches %d, f16 mismatches %d (element i at bits [6i, 6i+6), dst = RNE(src / scale))\n", scale, bad1, bad2);
            fails += bad1 + bad2;
        }
    }
    // ---- 1b. v_cvt_scalef32_2xpk16_fp6_f32(a, b, scale): where do a[i] and b[i] land?
    {
        std::vector<float> src(64 * 32);
        for (int l = 0; l < 64; ++l)
            for (int i = 0; i < 32; ++i)
                src[l * 32 + i] = e2m3((i + 1 + l) % 32 == 0 ? 33 : (i + 1 + l) % 32); // distinct magnitudes per input slot (never zero)
        float *dsrc;
        unsigned *dout;
        hipMalloc(&dsrc, src.size() * 4), hipMalloc(&dout, 64 * 6 * 4);
        hipMemcpy(dsrc, src.data(), src.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(cvt2x_kernel, dim3(1), dim3(64), 0, 0, dsrc, 1.0f, dout);
        std::vector<unsigned> o(64 * 6);
        hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
        int seq = 0, inter = 0;
        for (int l = 0; l < 64; ++l)
            for (int e = 0; e < 32; ++e) {
                const int bit = 6 * e;
                uint64_t w = o[l * 6 + bit / 32];
                if (bit / 32 + 1 < 6)
                    w |= (uint64_t)o[l * 6 + bit / 32 + 1] << 32;
                const float got = e2m3((int)((w >> (bit % 32)) & 63));
                seq += got == src[l * 32 + e];                                  // element e = input slot e (a then b)
                inter += got == src[l * 32 + (e % 2) * 16 + e / 2];             // element e = a[e / 2] (even) / b[e / 2] (odd)
            }
        printf("cvt_scalef32_2xpk16_fp6_f32: %d of 2048 elements match 'a then b', %d match 'interleaved a0 b0 a1 b1 ...'\n", seq, inter);
        if (seq != 2048 && inter != 2048)
            fails++;
    }
    // ---- 2. the MFMA's FP6 operand layout: A (FP4) one-hot per row n at k = (n + shift) % 64, B element (lane (m, h), i) carries the code 32 h + i
    {
        int *da, *db, *dsa, *dsb;
        float *dout;
        hipMalloc(&da, 2048), hipMalloc(&db, 2048), hipMalloc(&dsa, 256), hipMalloc(&dsb, 256), hipMalloc(&dout, 4096);
        std::vector<int> sa(64, 127), sb(64, 127);
        hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice), hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
        int bad = 0;
        for (int shift = 0; shift < 64; shift += 32) {
            std::vector<int> a(512, 0), b(512, 0);
            for (int n = 0; n < 32; ++n) {
                const int k = (n + shift) % 64, lane = n + 32 * (k / 32), q = k % 32; // FP4 operand: lane (row, h), nibble q = k - 32 h
                a[lane * 8 + q / 8] |= 2 << (4 * (q % 8));                           // code 2 = 1.0
            }
            for (int l = 0; l < 64; ++l)
                for (int i = 0; i < 32; ++i)
                    put6(b, l, i, (32 * (l / 32) + i + l % 32) % 64); // code names (k + m) % 64: differs per column too
            hipMemcpy(da, a.data(), 2048, hipMemcpyHostToDevice), hipMemcpy(db, b.data(), 2048, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(mfma_kernel, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dout);
            std::vector<float> out(1024);
            hipMemcpy(out.data(), dout, 4096, hipMemcpyDeviceToHost);
            // accumulator: lane (m = l % 32, hh = l / 32), v = 4 u + e <-> row n = 8 u + 4 hh + e
            for (int l = 0; l < 64; ++l)
                for (int v = 0; v < 16; ++v) {
                    const int m = l % 32, n = 8 * (v / 4) + 4 * (l / 32) + v % 4, k = (n + shift) % 64;
                    const float want = e2m3((k + m) % 64);
                    if (out[l * 16 + v] != want && bad++ < 8)
                        printf("  mfma: D[n %d][m %d] = %g, natural layout predicts %g (k = %d)\n", n, m, out[l * 16 + v], want, k);
                }
        }
        printf("mfma 32x32x64 A = FP4, B = FP6: natural-layout mismatches %d of 2048\n", bad);
        fails += bad;
        // ---- 3. the B scale byte: lane (m, h) scales ITS block; set 2^1 for h = 0 lanes of even columns, 2^-1 for h = 1 lanes of columns % 4 == 1
        {
            std::vector<int> a(512, 0), b(512, 0);
            for (int l = 0; l < 64; ++l) {
                for (int q = 0; q < 32; ++q)
                    a[l * 8 + q / 8] |= 2 << (4 * (q % 8)); // W = 1.0 everywhere
                for (int i = 0; i < 32; ++i)
                    put6(b, l, i, 8);                        // 1.0 everywhere
                sb[l] = 127 + ((l / 32 == 0 && (l % 2) == 0) ? 1 : 0) - ((l / 32 == 1 && (l % 4) == 1) ? 1 : 0);
            }
            hipMemcpy(da, a.data(), 2048, hipMemcpyHostToDevice), hipMemcpy(db, b.data(), 2048, hipMemcpyHostToDevice);
            hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(mfma_kernel, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dout);
            std::vector<float> out(1024);
            hipMemcpy(out.data(), dout, 4096, hipMemcpyDeviceToHost);
            int sbad = 0;
            for (int l = 0; l < 64; ++l) {
                const int m = l % 32;
                const float want = 32.f * ((m % 2) == 0 ? 2.f : 1.f) + 32.f * ((m % 4) == 1 ? 0.5f : 1.f);
                for (int v = 0; v < 16; ++v)
                    if (out[l * 16 + v] != want && sbad++ < 4)
                        printf("  scale: column %d got %g want %g\n", m, out[l * 16 + v], want);
            }
            printf("mfma B scale byte per (column, block) from lane (column, block): mismatches %d\n", sbad);
            fails += sbad;
        }
    }
    printf(fails ? "FAILED\n" : "OK\n");
    return fails ? 1 : 0;
}
