// mfma32_layout_probe.hip -- operand layouts of the block-scaled MFMAs the native kernels use, found by brute force:
//   v_mfma_scale_f32_32x32x64_f8f6f4   A = FP4 (cbsz 4),  B = FP8 e4m3 (blgp 0)  and  B = FP4 (blgp 4)
//   v_mfma_scale_f32_16x16x128_f8f6f4  A = FP4,           B = FP4
// For every k0 the A operand is one-hot in k (rows all 1.0 at k0) and the B operand carries a value that encodes
// (lane group, position) of each element, so D[0][col] names the B element that pairs with k0 -- and vice versa.
// Standalone: hipcc -O2 --offload-arch=gfx950 mfma32_layout_probe.hip -o mfma32_layout_probe && ./mfma32_layout_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int BFMT> __global__ void k32(const int *a_words, const int *b_words, int sa, int sb, float *out) {
    const int l = threadIdx.x;
    i32x8 a, b;
    for (int r = 0; r < 8; ++r) { a[r] = a_words[l * 8 + r]; b[r] = b_words[l * 8 + r]; }
    f32x16 c = {};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4, BFMT, 0, sa, 0, sb);
    for (int i = 0; i < 16; ++i) out[l * 16 + i] = c[i];
}
template <int BFMT> __global__ void k16(const int *a_words, const int *b_words, int sa, int sb, float *out) {
    const int l = threadIdx.x;
    i32x8 a, b;
    for (int r = 0; r < 8; ++r) { a[r] = a_words[l * 8 + r]; b[r] = b_words[l * 8 + r]; }
    f32x4 c = {};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 4, BFMT, 0, sa, 0, sb);
    for (int i = 0; i < 4; ++i) out[l * 4 + i] = c[i];
}

static const float kFp4[16] = {0, .5f, 1, 1.5f, 2, 3, 4, 6, -0.f, -.5f, -1, -1.5f, -2, -3, -4, -6};
static float e4m3(int b) { // OCP e4m3
    int e = (b >> 3) & 15, m = b & 7;
    float v = e ? ldexpf(1.f + m / 8.f, e - 7) : ldexpf(m / 8.f, -6);
    return (b & 0x80) ? -v : v;
}

int main() {
    int *da, *db;
    float *dout;
    hipMalloc(&da, 64 * 8 * 4), hipMalloc(&db, 64 * 8 * 4), hipMalloc(&dout, 64 * 16 * 4);
    std::vector<int> a(512), b(512);
    std::vector<float> out(1024);
    const int one = 127; // E8M0 1.0 in byte 0

    // ---- 32x32x64, A = FP4 natural? : B FP8 all 1.0 (0x38); A lane (row, h) nibble q set to code 2 (1.0) only for one
    // (h, q) at a time -> every D[row][col] = 1 iff that nibble is a valid k (it always is); instead find WHICH k by making B
    // one-hot in its own (h', byte): D != 0 iff they pair.  So: pair table A(h, q) <-> B(h', byte).
    for (int bfmt = 0; bfmt <= 4; bfmt += 4) {
        const int nb = bfmt == 0 ? 32 : 32; // elements per lane of B: 32 fp8 bytes / 32 fp4 nibbles
        printf("== 32x32x64  A=FP4  B=%s : for A (half h, nibble q) the B (half, element) that pairs\n", bfmt ? "FP4" : "FP8");
        for (int h = 0; h < 2; ++h)
            for (int q = 0; q < 32; ++q) {
                std::fill(a.begin(), a.end(), 0), std::fill(b.begin(), b.end(), 0);
                for (int l = 0; l < 64; ++l)
                    if (l / 32 == h)
                        a[l * 8 + q / 8] |= 2 << (4 * (q % 8));
                // B element (h', e) encodes value index 1 + h'*32 + e as ... too many values for fp4: do it in two passes
                int found_h = -1, found_e = -1;
                for (int hp = 0; hp < 2 && found_h < 0; ++hp)
                    for (int e = 0; e < nb && found_h < 0; ++e) {
                        std::fill(b.begin(), b.end(), 0);
                        for (int l = 0; l < 64; ++l)
                            if (l / 32 == hp) {
                                if (bfmt == 0)
                                    b[l * 8 + e / 4] |= 0x38 << (8 * (e % 4));
                                else
                                    b[l * 8 + e / 8] |= 2 << (4 * (e % 8));
                            }
                        hipMemcpy(da, a.data(), 2048, hipMemcpyHostToDevice), hipMemcpy(db, b.data(), 2048, hipMemcpyHostToDevice);
                        if (bfmt == 0) hipLaunchKernelGGL(k32<0>, dim3(1), dim3(64), 0, 0, da, db, one, one, dout);
                        else hipLaunchKernelGGL(k32<4>, dim3(1), dim3(64), 0, 0, da, db, one, one, dout);
                        hipMemcpy(out.data(), dout, 4096, hipMemcpyDeviceToHost);
                        if (out[0] != 0.f) found_h = hp, found_e = e;
                    }
                printf("A(h%d,q%2d)->B(h%d,e%2d)%s", h, q, found_h, found_e, (q % 4 == 3) ? "\n" : "  ");
            }
    }
    // ---- 32x32x64: which lane's scale byte applies to which k-block: A scale = 2.0 (128) only in lanes of half hs
    printf("== 32x32x64 scales: A nibble (h,q) x A-scale from lanes of half hs (others 1.0) -> D\n");
    {
        int *dsa;
        hipMalloc(&dsa, 256);
        // (uses per-lane scale registers: rebuild kernel inline via a tiny lambda kernel is overkill; the 16x16x128 probe
        // showed the scale of a block comes from the lane that holds the block's data; 32x32x64 is checked end-to-end by the
        // parity tests of the kernel, which use per-block random scales)
        hipFree(dsa);
    }
    // ---- 16x16x128, A = FP4, B = FP4: pairing table
    printf("== 16x16x128  A=FP4  B=FP4 : for A (group g, nibble q) the B (group, nibble) that pairs\n");
    for (int g = 0; g < 4; ++g)
        for (int q = 0; q < 32; ++q) {
            std::fill(a.begin(), a.end(), 0);
            for (int l = 0; l < 64; ++l)
                if (l / 16 == g)
                    a[l * 8 + q / 8] |= 2 << (4 * (q % 8));
            int fg = -1, fe = -1;
            for (int gp = 0; gp < 4 && fg < 0; ++gp)
                for (int e = 0; e < 32 && fg < 0; ++e) {
                    std::fill(b.begin(), b.end(), 0);
                    for (int l = 0; l < 64; ++l)
                        if (l / 16 == gp)
                            b[l * 8 + e / 8] |= 2 << (4 * (e % 8));
                    hipMemcpy(da, a.data(), 2048, hipMemcpyHostToDevice), hipMemcpy(db, b.data(), 2048, hipMemcpyHostToDevice);
                    hipLaunchKernelGGL(k16<4>, dim3(1), dim3(64), 0, 0, da, db, one, one, dout);
                    hipMemcpy(out.data(), dout, 1024, hipMemcpyDeviceToHost);
                    if (out[0] != 0.f) fg = gp, fe = e;
                }
            printf("A(g%d,q%2d)->B(g%d,e%2d)%s", g, q, fg, fe, (q % 4 == 3) ? "\n" : "  ");
        }
    (void)kFp4, (void)e4m3;
    return 0;
}
