// mfma_scale_align.hip -- how exact is the sum inside ONE v_mfma_scale_f32_{32x32x64,16x16x128}_f8f6f4 on gfx950?
//
// The native class's parity tests carried an empirical "cancellation allowance" (1e-5 -> 2e-5 -> 4e-5 of sum |a||w|, raised whenever a fuzz run
// found a worse case).  This probe replaces it by a measured property of the instruction: with A = FP4 (the weights) and
// B = FP8 e4m3 / FP6 e2m3 / FP4 (the quantised activations), each with E8M0 block scales, it builds ONE output element whose exact value is known
//      D[0][0] = C + sum_k A[0][k] * B[k][0]          (A = 1.0 at a few chosen k, 0 elsewhere)
// and asks how many bits of a SMALL term survive next to LARGE terms that cancel:
//   same    +big, -big, small in ONE 32-element block (one scale): the gap is limited by the element format's own range;
//   cross   +big, -big in block 0, small in block 1 of the same instruction, block scales 2^G apart: any gap;
//   acc     +big, -big as products, small arrives in the accumulator C;
//   round   big + small without cancellation: is the final rounding to nearest or truncation?
// For every case and gap G = log2(big / small): the result, the exact value, and the surviving bits of `small` (24 = exact to f32).
// Prints one line per measurement plus a summary: the largest gap at which `small` is still exact and the gap beyond which it is lost altogether,
// per (instruction, B format, case).  Standalone: hipcc -O2 --offload-arch=gfx950 mfma_scale_align.hip -o mfma_scale_align && ./mfma_scale_align
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <vector>

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// A = FP4 (cbsz 4); B format BLGP: 0 = FP8 e4m3, 2 = FP6 e2m3, 4 = FP4 e2m1
template <int BLGP> __global__ void k32(const int *a_words, const int *b_words, const int *sa, const int *sb, const float *c_in, float *out) {
    const int l = threadIdx.x;
    i32x8 a, b;
    for (int r = 0; r < 8; ++r)
        a[r] = a_words[l * 8 + r], b[r] = b_words[l * 8 + r];
    f32x16 c;
    for (int i = 0; i < 16; ++i)
        c[i] = c_in[l * 16 + i];
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4, BLGP, 0, sa[l], 0, sb[l]);
    for (int i = 0; i < 16; ++i)
        out[l * 16 + i] = c[i];
}
template <int BLGP> __global__ void k16(const int *a_words, const int *b_words, const int *sa, const int *sb, const float *c_in, float *out) {
    const int l = threadIdx.x;
    i32x8 a, b;
    for (int r = 0; r < 8; ++r)
        a[r] = a_words[l * 8 + r], b[r] = b_words[l * 8 + r];
    f32x4 c;
    for (int i = 0; i < 4; ++i)
        c[i] = c_in[l * 16 + i];
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 4, BLGP, 0, sa[l], 0, sb[l]);
    for (int i = 0; i < 4; ++i)
        out[l * 16 + i] = c[i];
}

static float dec_fp4(int c) {
    static const float t[8] = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f};
    return (c & 8) ? -t[c & 7] : t[c & 7];
}
static float dec_fp6(int c) { // e2m3, bias 1
    const int e = (c >> 3) & 3, m = c & 7;
    const float v = e ? ldexpf(1.f + m / 8.f, e - 1) : m / 8.f;
    return (c & 32) ? -v : v;
}
static float dec_fp8(int c) { // OCP e4m3fn, bias 7
    const int e = (c >> 3) & 15, m = c & 7;
    const float v = e ? ldexpf(1.f + m / 8.f, e - 7) : ldexpf(m / 8.f, -6);
    return (c & 128) ? -v : v;
}
static int enc(int fmt, float x) { // exact encoder: the code whose value is x (aborts when x is not representable)
    const int n = fmt == 0 ? 256 : fmt == 2 ? 64 : 16;
    for (int c = 0; c < n; ++c) {
        const float v = fmt == 0 ? dec_fp8(c) : fmt == 2 ? dec_fp6(c) : dec_fp4(c);
        if (fmt == 0 && (c & 127) == 127)
            continue; // NaN
        if (v == x && (x != 0.f || c == 0))
            return c;
    }
    fprintf(stderr, "value %g not representable in format %d\n", x, fmt);
    exit(2);
}

struct Operands {
    std::vector<int> a, b, sa, sb;
    std::vector<float> c;
    Operands() : a(64 * 8, 0), b(64 * 8, 0), sa(64, 127), sb(64, 127), c(64 * 16, 0.f) {}
};
// element i of lane `lane`'s operand, `bits` wide, LSB first
static void put(std::vector<int> &w, int lane, int i, int bits, int code) {
    const int bit = bits * i;
    const uint64_t v = (uint64_t)(code & ((1 << bits) - 1)) << (bit % 32);
    w[lane * 8 + bit / 32] |= (int)(uint32_t)v;
    if (bit % 32 + bits > 32)
        w[lane * 8 + bit / 32 + 1] |= (int)(uint32_t)(v >> 32);
}
static int bits_of(int fmt) { return fmt == 0 ? 8 : fmt == 2 ? 6 : 4; }

struct Runner {
    int *da, *db, *dsa, *dsb;
    float *dc, *dout;
    Runner() {
        hipMalloc(&da, 64 * 8 * 4), hipMalloc(&db, 64 * 8 * 4), hipMalloc(&dsa, 64 * 4), hipMalloc(&dsb, 64 * 4);
        hipMalloc(&dc, 64 * 16 * 4), hipMalloc(&dout, 64 * 16 * 4);
    }
    float run(const Operands &o, int shape, int fmt) { // -> D[0][0]
        hipMemcpy(da, o.a.data(), 64 * 8 * 4, hipMemcpyHostToDevice), hipMemcpy(db, o.b.data(), 64 * 8 * 4, hipMemcpyHostToDevice);
        hipMemcpy(dsa, o.sa.data(), 64 * 4, hipMemcpyHostToDevice), hipMemcpy(dsb, o.sb.data(), 64 * 4, hipMemcpyHostToDevice);
        hipMemcpy(dc, o.c.data(), 64 * 16 * 4, hipMemcpyHostToDevice);
        if (shape == 32) {
            if (fmt == 0) hipLaunchKernelGGL(k32<0>, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dc, dout);
            if (fmt == 2) hipLaunchKernelGGL(k32<2>, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dc, dout);
            if (fmt == 4) hipLaunchKernelGGL(k32<4>, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dc, dout);
        } else {
            if (fmt == 0) hipLaunchKernelGGL(k16<0>, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dc, dout);
            if (fmt == 2) hipLaunchKernelGGL(k16<2>, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dc, dout);
            if (fmt == 4) hipLaunchKernelGGL(k16<4>, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dc, dout);
        }
        float out[16];
        hipMemcpy(out, dout, sizeof(out), hipMemcpyDeviceToHost);
        return out[0];
    }
};

// lane of (row-or-column 0, k-block `blk`): 32x32x64 has two 32-element blocks per operand row (lanes 0 and 32), 16x16x128 four (lanes 0, 16, 32, 48);
// the E8M0 scale byte of block b comes from that lane, for every element format (csrc/gemm_native32.hpp, layout note)
static int lane_of_block(int shape, int blk) { return shape == 32 ? 32 * blk : 16 * blk; }
// where element k of row / column 0 sits: FP4 and FP6 operands hold 32 consecutive k per lane (lane group = k / 32); an FP8 operand's lane group g holds
// k = 16 g .. 16 g + 15 in registers 0-3 and HALF + 16 g .. HALF + 16 g + 15 in registers 4-7 (HALF = 32 for 32x32x64, 64 for 16x16x128): probed on gfx950
// (tools/probes/mfma32_layout_probe.hip), and checked again by layout_selfcheck() below before anything is measured
static void locate(int shape, int fmt, int k, int *lane, int *elem) {
    const int step = shape == 32 ? 32 : 16, half = shape == 32 ? 32 : 64;
    if (fmt != 0) {
        *lane = step * (k / 32), *elem = k % 32;
    } else if (k < half) {
        *lane = step * (k / 16), *elem = k % 16;
    } else {
        *lane = step * ((k - half) / 16), *elem = 16 + (k - half) % 16;
    }
}
// one term: weight `wgt` (FP4) at k of row 0 of A; activation `act` at k of column 0 of B
static void term(Operands &o, int shape, int fmt, int k, float act, float wgt = 1.0f) {
    int lane, elem;
    locate(shape, 4, k, &lane, &elem);
    put(o.a, lane, elem, 4, enc(4, wgt));
    locate(shape, fmt, k, &lane, &elem);
    put(o.b, lane, elem, bits_of(fmt), enc(fmt, act));
}

static double surviving_bits(double got, double exact, double small) {
    const double err = fabs(got - exact);
    if (err == 0.0)
        return 24.0;
    const double b = -log2(err / fabs(small));
    return b < 0 ? 0.0 : b > 24 ? 24.0 : b;
}

int main() {
    Runner R;
    const char *fname[5] = {"fp8_e4m3", "", "fp6_e2m3", "", "fp4_e2m1"};
    // layout self-check: every single k, alone, must contribute exactly its product (and its block's scales)
    for (int shape : {32, 16})
        for (int fmt : {0, 2, 4}) {
            int bad = 0;
            const int K = shape == 32 ? 64 : 128;
            for (int k = 0; k < K; ++k) {
                Operands o;
                term(o, shape, fmt, k, 1.5f, 2.0f);
                o.sb[lane_of_block(shape, k / 32)] = 127 + 3, o.sa[lane_of_block(shape, k / 32)] = 127 - 1;
                const float got = R.run(o, shape, fmt);
                if (got != 1.5f * 2.0f * 8.0f * 0.5f && bad++ < 4)
                    printf("LAYOUT %dx%d %s k=%d: got %g want 12\n", shape, shape, fname[fmt], k, got);
            }
            printf("layout self-check %dx%d %s: %d of %d positions wrong\n", shape, shape, fname[fmt], bad, K);
        }
    printf("# instr bfmt case gap_log2 big small c_in got exact bits_of_small_surviving\n");
    for (int shape : {32, 16}) {
        for (int fmt : {0, 2, 4}) {
            const float big = fmt == 0 ? 448.f : fmt == 2 ? 7.5f : 6.f;       // the format's largest value
            const float full = fmt == 0 ? 1.875f : fmt == 2 ? 1.875f : 1.5f;  // a value with every mantissa bit set
            const int nblk = shape == 32 ? 2 : 4;
            struct Sum { int exact_upto = -1, lost_from = 1 << 30; } sum_same, sum_cross[4], sum_acc, sum_acc_nocancel;
            // ---- same block: gap from the element format alone (small = full * 2^-e as far as the format goes)
            for (int e = 0; e < 16; ++e) {
                const float small = ldexpf(full, -e);
                bool ok = true;
                const int n = fmt == 0 ? 256 : fmt == 2 ? 64 : 16;
                ok = false;
                for (int c = 0; c < n; ++c)
                    if ((fmt == 0 ? dec_fp8(c) : fmt == 2 ? dec_fp6(c) : dec_fp4(c)) == small && !(fmt == 0 && (c & 127) == 127))
                        ok = true;
                if (!ok)
                    continue;
                for (int pos_small : {2, 17, 31}) {
                    Operands o;
                    term(o, shape, fmt, 0, big), term(o, shape, fmt, 1, -big), term(o, shape, fmt, pos_small, small);
                    const float got = R.run(o, shape, fmt);
                    const double gap = log2((double)big / small);
                    const double bits = surviving_bits(got, small, small);
                    printf("%dx%d %s same(pos%d) %.2f %g %g 0 %.9g %.9g %.1f\n", shape, shape, fname[fmt], pos_small, gap, big, small, got, small, bits);
                    if (bits >= 24 && (int)gap > sum_same.exact_upto) sum_same.exact_upto = (int)gap;
                    if (bits <= 0 && (int)gap < sum_same.lost_from) sum_same.lost_from = (int)gap;
                }
            }
            // ---- same block, the widest gap the formats allow: weights 6 on the big pair and 0.5 on the small term; the pair adjacent (k = 0, 1) and far apart (k = 0, 31)
            for (int e = 0; e < 16; ++e) {
                const float small = ldexpf(full, -e);
                bool ok = false;
                const int n = fmt == 0 ? 256 : fmt == 2 ? 64 : 16;
                for (int c = 0; c < n; ++c)
                    if ((fmt == 0 ? dec_fp8(c) : fmt == 2 ? dec_fp6(c) : dec_fp4(c)) == small && !(fmt == 0 && (c & 127) == 127))
                        ok = true;
                if (!ok)
                    continue;
                for (int far = 0; far < 2; ++far) {
                    Operands o;
                    term(o, shape, fmt, 0, big, 6.0f), term(o, shape, fmt, far ? 31 : 1, -big, 6.0f), term(o, shape, fmt, 16, small, 0.5f);
                    const float got = R.run(o, shape, fmt);
                    const double exact = 0.5 * small, gap = log2(6.0 * big / exact);
                    printf("%dx%d %s same6(%s) %.2f %g %g 0 %.9g %.9g %.1f\n", shape, shape, fname[fmt], far ? "pair k=0,31" : "pair k=0,1", gap, 6.0 * big, exact, got, exact,
                           surviving_bits(got, exact, exact));
                }
            }
            // ---- pair sums without cancellation: 6 * big at k = 0 plus 0.5 * small at k = d.  The exact sum needs (gap + 4) bits: how many does the instruction keep,
            //      and does it depend on how far apart the two k are (i.e. on which terms are added first)?
            for (int d : {1, 2, 3, 4, 5, 8, 15, 16, 17, 31}) {
                int kept_min = 99;
                for (int e = 0; e < 16; ++e) {
                    const float small = ldexpf(full, -e);
                    bool ok = false;
                    const int n = fmt == 0 ? 256 : fmt == 2 ? 64 : 16;
                    for (int c = 0; c < n; ++c)
                        if ((fmt == 0 ? dec_fp8(c) : fmt == 2 ? dec_fp6(c) : dec_fp4(c)) == small && !(fmt == 0 && (c & 127) == 127))
                            ok = true;
                    if (!ok)
                        continue;
                    Operands o;
                    term(o, shape, fmt, 0, big, 6.0f), term(o, shape, fmt, d, small, 0.5f);
                    const float got = R.run(o, shape, fmt);
                    const double exact = 6.0 * big + 0.5 * small;
                    const double err = fabs((double)got - exact);
                    // bits of the sum kept below its leading bit: 24 = an exactly rounded f32
                    const int kept = err == 0.0 ? 24 : (int)floor(log2(fabs(exact)) - log2(err));
                    if (kept < kept_min) kept_min = kept;
                    printf("%dx%d %s pairsum(d=%d) gap %.2f exact %.9g got %.9g bits_kept %d\n", shape, shape, fname[fmt], d, log2(6.0 * big / (0.5 * small)), exact, (double)got, kept);
                }
                printf("PAIRSUM %dx%d %s d=%d: fewest significant bits of big + small kept = %d\n", shape, shape, fname[fmt], d, kept_min);
            }
            // ---- random blocks: the error of ONE instruction against the exact sum, in units of 2^(E - 24), E = floor(log2(largest |term|)), terms = the products and C.
            //      (i) one block non-zero; (ii) every block non-zero, block scales within +-S binades; C random of the size of the sum or zero
            {
                unsigned long long rng = 0x9E3779B97F4A7C15ull ^ (unsigned long long)(shape * 131 + fmt);
                auto next = [&]() { rng ^= rng << 13, rng ^= rng >> 7, rng ^= rng << 17; return rng; };
                const int ncodes = fmt == 0 ? 256 : fmt == 2 ? 64 : 16, K = shape == 32 ? 64 : 128;
                for (int mode = 0; mode < 4; ++mode) { // 0: one block, C = 0; 1: one block, C random; 2: all blocks scales +-4, C random; 3: all blocks scales +-12, C random
                    double worst = 0.0, worst_rel = 0.0;
                    for (int trial = 0; trial < 1500; ++trial) {
                        Operands o;
                        double exact = 0.0, maxterm = 0.0, sumabs = 0.0;
                        const int nblk_used = mode < 2 ? 1 : nblk;
                        for (int b = 0; b < nblk_used; ++b) {
                            const int sb = mode == 2 ? (int)(next() % 9) - 4 : mode == 3 ? (int)(next() % 25) - 12 : 0;
                            o.sb[lane_of_block(shape, b)] = 127 + sb;
                            for (int i = 0; i < 32; ++i) {
                                int code;
                                do
                                    code = (int)(next() % ncodes);
                                while (fmt == 0 && (code & 127) == 127);
                                const float act = fmt == 0 ? dec_fp8(code) : fmt == 2 ? dec_fp6(code) : dec_fp4(code);
                                const float wgt = dec_fp4((int)(next() % 16));
                                int lane, elem;
                                locate(shape, 4, 32 * b + i, &lane, &elem);
                                put(o.a, lane, elem, 4, enc(4, wgt == 0.f ? 0.f : wgt));
                                locate(shape, fmt, 32 * b + i, &lane, &elem);
                                put(o.b, lane, elem, bits_of(fmt), code);
                                const double p = (double)act * wgt * ldexp(1.0, sb);
                                exact += p, sumabs += fabs(p);
                                if (fabs(p) > maxterm) maxterm = fabs(p);
                            }
                        }
                        (void)K;
                        float cin = 0.f;
                        if (mode >= 1) {
                            cin = (float)(((double)(next() % 2000001) / 1000000.0 - 1.0) * (sumabs / 8.0 + 1e-3));
                            o.c[0] = cin;
                            exact += cin;
                            if (fabs((double)cin) > maxterm) maxterm = fabs((double)cin);
                        }
                        if (maxterm == 0.0)
                            continue;
                        const float got = R.run(o, shape, fmt);
                        const double unit = ldexp(1.0, (int)floor(log2(maxterm)) - 24);
                        const double err = fabs((double)got - exact);
                        // (the final f32 rounding of the result itself is not the instruction's internal error: subtract half an ulp of the result)
                        const double res_ulp = exact == 0.0 ? 0.0 : ldexp(1.0, (int)floor(log2(fabs(exact))) - 23);
                        const double internal = err > res_ulp ? err - res_ulp : 0.0;
                        if (internal / unit > worst) worst = internal / unit;
                        if (err / sumabs > worst_rel) worst_rel = err / sumabs;
                    }
                    printf("RANDOM %dx%d %s mode %d (%s): worst internal error = %.2f units of 2^(E-24); worst |err| / sum|terms| = %.3g\n", shape, shape, fname[fmt], mode,
                           mode == 0 ? "one block, C = 0" : mode == 1 ? "one block, C random" : mode == 2 ? "all blocks, scales +-4 binades, C random" : "all blocks, scales +-12 binades, C random",
                           worst, worst_rel);
                }
            }
            // ---- cross block: big pair in block 0 (scale 2^G), small in block b (scale 1)
            for (int b = 1; b < nblk; ++b)
                for (int G = 0; G <= 64; ++G) {
                    Operands o;
                    term(o, shape, fmt, 0, big), term(o, shape, fmt, 1, -big), term(o, shape, fmt, 32 * b + 5, full);
                    o.sb[lane_of_block(shape, 0)] = 127 + G; // activations of block 0 scaled by 2^G
                    const float got = R.run(o, shape, fmt);
                    const double gap = G + log2((double)big / full);
                    const double bits = surviving_bits(got, full, full);
                    printf("%dx%d %s cross(blk%d) %.2f %g %g 0 %.9g %.9g %.1f\n", shape, shape, fname[fmt], b, gap, ldexp((double)big, G), full, got, (double)full, bits);
                    if (bits >= 24 && (int)gap > sum_cross[b].exact_upto) sum_cross[b].exact_upto = (int)gap;
                    if (bits <= 0 && (int)gap < sum_cross[b].lost_from) sum_cross[b].lost_from = (int)gap;
                }
            // ---- accumulator: +big, -big as products (scale 2^G), C = a full-mantissa f32
            for (int G = 0; G <= 64; ++G) {
                Operands o;
                term(o, shape, fmt, 0, big), term(o, shape, fmt, 1, -big);
                o.sb[lane_of_block(shape, 0)] = 127 + G;
                const float cin = 1.2345678f;
                o.c[0] = cin;
                const float got = R.run(o, shape, fmt);
                const double gap = G + log2((double)big / cin);
                const double bits = surviving_bits(got, cin, cin);
                printf("%dx%d %s acc %.2f %g 0 %.9g %.9g %.9g %.1f\n", shape, shape, fname[fmt], gap, ldexp((double)big, G), cin, got, (double)cin, bits);
                if (bits >= 24 && (int)gap > sum_acc.exact_upto) sum_acc.exact_upto = (int)gap;
                if (bits <= 0 && (int)gap < sum_acc.lost_from) sum_acc.lost_from = (int)gap;
            }
            // ---- rounding: big * 2^G + small, no cancellation (exact in double); and C + product
            int rne = 0, trunc = 0, other = 0;
            for (int G = 0; G <= 30; ++G)
                for (int b = 0; b < nblk; ++b) {
                    Operands o;
                    term(o, shape, fmt, 0, big), term(o, shape, fmt, 32 * b + 7, full);
                    o.sb[lane_of_block(shape, 0)] = 127 + G;
                    if (b == 0) { // same block: both scaled
                    }
                    const double exact = ldexp((double)big, G) + (b == 0 ? ldexp((double)full, G) : (double)full);
                    const float got = R.run(o, shape, fmt);
                    const float want_rne = (float)exact;
                    float want_trunc = want_rne;
                    if (fabs((double)want_rne) > fabs(exact))
                        want_trunc = nextafterf(want_rne, 0.f);
                    if (got == want_rne) ++rne;
                    if (got == want_trunc) ++trunc;
                    if (got != want_rne && got != want_trunc) {
                        ++other;
                        printf("%dx%d %s round(blk%d) G=%d got %.9g rne %.9g trunc %.9g exact %.12g\n", shape, shape, fname[fmt], b, G, got, want_rne, want_trunc, exact);
                    }
                }
            printf("SUMMARY %dx%d %s: same-block small exact up to gap 2^%d (format range ends there unless lost_from is set: %d); ", shape, shape, fname[fmt], sum_same.exact_upto,
                   sum_same.lost_from == 1 << 30 ? -1 : sum_same.lost_from);
            for (int b = 1; b < nblk; ++b)
                printf("cross-block(%d) exact up to 2^%d, lost from 2^%d; ", b, sum_cross[b].exact_upto, sum_cross[b].lost_from == 1 << 30 ? -1 : sum_cross[b].lost_from);
            printf("accumulator exact up to 2^%d, lost from 2^%d; no-cancel sums: %d match RNE, %d match truncation, %d neither\n", sum_acc.exact_upto,
                   sum_acc.lost_from == 1 << 30 ? -1 : sum_acc.lost_from, rne, trunc, other);
        }
    }
    return 0;
}
