/*
 * petit_oracle.c -- CPU restatement of the reference's FP4 GEMM path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under petit-kernel_amd/ may include,
 * link or call this file.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, and only as the checker.
 *
 * Every function cites the reference file:line (paths relative to
 * /root/reference) whose behaviour it restates.  The restatement is pinned
 * against (a) golden vectors produced by importing the reference's own Python
 * oracle (tests/golden/, generator tests/golden/make_golden.py), (b) the
 * reference's host-side software float types compiled from where they lie
 * (oracle/_ref, see oracle/Makefile) and (c) the reference's known-answer
 * tables (16 fp4 codes x 126 e4m3 scales, 16 codes x e8m0 1..237).
 *
 * Plain C11, no dependencies beyond libm / OpenMP (optional).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------
 * Scalar number formats
 * --------------------------------------------------------------------- */

static inline float bits_to_f32(uint32_t u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static inline uint32_t f32_to_bits(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}

/* FP4 E2M1 code -> value.  Table of tests/ops/test_fp4_gemm_quark.py:10-14 and
 * lib/gemm/rocm/quantization/fp4/quantization_utils_fp4_test.cc:259-262. */
static const float kFp4Values[16] = {
    0.0f,  0.5f,  1.0f,  1.5f,  2.0f,  3.0f,  4.0f,  6.0f,
    -0.0f, -0.5f, -1.0f, -1.5f, -2.0f, -3.0f, -4.0f, -6.0f,
};

float po_fp4_to_f32(unsigned code) { return kFp4Values[code & 15]; }

/* OCP FP8 E4M3 (fn) -> f32.  Follows lib/tests/floating_points.h:21-75
 * (CvtFp32Fp8<false,false>): bias 7, no infinities, 0x7f/0xff are NaN,
 * subnormals are m * 2^-9. */
float po_e4m3_to_f32(uint8_t x) {
    uint32_t sign = x >> 7;
    uint32_t man = x & 7;
    int exp = (x & 0x7f) >> 3;
    if ((x & 0x7f) == 0x7f)
        return NAN;
    if ((x & 0x7f) == 0)
        return sign ? -0.0f : 0.0f;
    float mag;
    if (exp == 0)
        mag = ldexpf((float)man, -9); /* 2^(1-7) * man/8 */
    else
        mag = ldexpf(1.0f + (float)man / 8.0f, exp - 7);
    return sign ? -mag : mag;
}

/* OCP FP8 E5M2 -> f32 (floating_points.h:21-75 with kIsBf8 = true). Used only
 * to prove the PetitFormat truth table (each re-encoded byte decodes to
 * fp4 * 2^-14). */
float po_e5m2_to_f32(uint8_t x) {
    uint32_t sign = x >> 7;
    uint32_t man = x & 3;
    int exp = (x & 0x7f) >> 2;
    if (exp == 31)
        return man ? NAN : (sign ? -INFINITY : INFINITY);
    float mag;
    if (exp == 0)
        mag = ldexpf((float)man, -16); /* 2^(1-15) * man/4 */
    else
        mag = ldexpf(1.0f + (float)man / 4.0f, exp - 15);
    return sign ? -mag : mag;
}

/* E8M0 block scale -> f32.  lib/gemm/rocm/quantization/dequant.cuh:198-203
 * places the byte in the bf16 exponent field, i.e. 2^(s-127); s = 0 therefore
 * decodes to 0.0 (exponent field 0, mantissa 0) and s = 255 to +inf.  The
 * reference's own tests only draw 1..237
 * (quantization_utils_fp4_test.cc:266-271). */
float po_e8m0_to_f32(uint8_t s) { return bits_to_f32((uint32_t)s << 23); }

/* f32 -> bf16, round-to-nearest-even, NaN kept quiet.
 * lib/tests/floating_points.h:79-112 (CvtBf16Fp32). */
uint16_t po_f32_to_bf16(float f) {
    uint32_t x = f32_to_bits(f);
    if ((x & 0x7f800000u) != 0x7f800000u) {
        x += 0x7fffu + ((x >> 16) & 1u);
    } else if (x & 0xffffu) {
        x |= 0x10000u;
    }
    return (uint16_t)(x >> 16);
}
float po_bf16_to_f32(uint16_t h) { return bits_to_f32((uint32_t)h << 16); }

/* f32 <-> IEEE binary16, round-to-nearest-even (what half_float::half(float)
 * does in lib/tests/floating_points.h:145-150 and what `.to(torch.float16)`
 * does in tests/ops/test_fp4_gemm_quark.py:24). */
uint16_t po_f32_to_f16(float f) {
    uint32_t x = f32_to_bits(f);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t ax = x & 0x7fffffffu;
    if (ax >= 0x7f800000u) /* inf / nan */
        return (uint16_t)(sign | 0x7c00u | ((ax > 0x7f800000u) ? 0x200u : 0u));
    if (ax >= 0x477ff000u) /* rounds to >= 65520 -> inf */
        return (uint16_t)(sign | 0x7c00u);
    if (ax < 0x33000001u) /* < 2^-25 (or exactly 2^-25: ties to even 0) */
        return (uint16_t)sign;
    int e = (int)(ax >> 23) - 127;
    uint32_t m = (ax & 0x7fffffu) | 0x800000u;
    int shift;
    uint32_t he;
    if (e < -14) { /* subnormal half */
        shift = 13 + (-14 - e);
        he = 0;
    } else {
        shift = 13;
        he = (uint32_t)(e + 15);
    }
    uint32_t q = m >> shift;
    uint32_t rem = m & ((1u << shift) - 1u);
    uint32_t half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u)))
        q++;
    uint32_t r;
    if (he == 0)
        r = q; /* may carry into exponent 1: still the right bit pattern */
    else
        r = ((he - 1) << 10) + q; /* q includes the hidden bit (0x400) */
    return (uint16_t)(sign | r);
}
float po_f16_to_f32(uint16_t h) {
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    int exp = (h >> 10) & 31;
    uint32_t man = h & 0x3ffu;
    float mag;
    if (exp == 31)
        return bits_to_f32(sign | 0x7f800000u | (man << 13));
    if (exp == 0)
        mag = ldexpf((float)man, -24);
    else
        mag = ldexpf(1.0f + (float)man / 1024.0f, exp - 15);
    return sign ? -mag : mag;
}

/* ------------------------------------------------------------------------
 * Dense dequantisation of the *native* checkpoint formats
 * --------------------------------------------------------------------- */

/* tests/ops/test_fp4_gemm_quark.py:9-20 (_dequant_nvfp4):
 *   q   : uint8 [n][k/2], low nibble = even k, high nibble = odd k
 *   s   : e4m3  [n][k/16]
 *   out : f32   [n][k] = LUT[q] * float(s)                                 */
void po_dequant_nvfp4(const uint8_t *q, const uint8_t *s, int n, int k,
                      float *out) {
#pragma omp parallel for schedule(static)
    for (int r = 0; r < n; ++r) {
        const uint8_t *qr = q + (size_t)r * (k / 2);
        const uint8_t *sr = s + (size_t)r * (k / 16);
        float *o = out + (size_t)r * k;
        for (int c = 0; c < k; c += 2) {
            float sc = po_e4m3_to_f32(sr[c / 16]);
            uint8_t b = qr[c / 2];
            o[c] = kFp4Values[b & 15] * sc;
            o[c + 1] = kFp4Values[b >> 4] * sc;
        }
    }
}

/* MXFP4 twin: group 32 (lib/pybind/fp4.cc:132-135), scale 2^(e-127)
 * (dequant.cuh:198-203).  tests/ops/test_fp4_gemm_quark.py:83 delegates this
 * to amd-quark's dq_mxfp4 (unpinned, not in /root/reference): "parity
 * unpinned" at that boundary; pinned instead by the in-tree semantics above
 * and by torch.float8_e8m0fnu (tests/test_oracle.py).                       */
void po_dequant_mxfp4(const uint8_t *q, const uint8_t *s, int n, int k,
                      float *out) {
#pragma omp parallel for schedule(static)
    for (int r = 0; r < n; ++r) {
        const uint8_t *qr = q + (size_t)r * (k / 2);
        const uint8_t *sr = s + (size_t)r * (k / 32);
        float *o = out + (size_t)r * k;
        for (int c = 0; c < k; c += 2) {
            float sc = po_e8m0_to_f32(sr[c / 32]);
            uint8_t b = qr[c / 2];
            o[c] = kFp4Values[b & 15] * sc;
            o[c + 1] = kFp4Values[b >> 4] * sc;
        }
    }
}

/* ------------------------------------------------------------------------
 * Reference GEMM
 * --------------------------------------------------------------------- */

/* tests/ops/test_fp4_gemm_quark.py:23-24,51-53:
 *     b_ref = dequant * global_scale            (f32, on the weights)
 *     c     = (a.float() @ b_ref.T.float()).to(a.dtype)
 * a_kind: 0 = fp16 bits, 1 = bf16 bits.  The product is accumulated in
 * double so the oracle does not carry an accumulation-order artefact of its
 * own; the result is rounded once to f32 and once to the 16-bit type.       */
void po_gemm_ref(const uint16_t *a, int a_kind, const float *b_dq,
                 float global_scale, int m, int n, int k, uint16_t *c,
                 float *c_f32 /* optional, may be NULL */) {
    float *af = (float *)malloc((size_t)m * k * sizeof(float));
    for (size_t i = 0; i < (size_t)m * k; ++i)
        af[i] = a_kind ? po_bf16_to_f32(a[i]) : po_f16_to_f32(a[i]);
#pragma omp parallel for schedule(static)
    for (int j = 0; j < n; ++j) {
        const float *br = b_dq + (size_t)j * k;
        for (int i = 0; i < m; ++i) {
            const float *ar = af + (size_t)i * k;
            double acc = 0.0;
            for (int t = 0; t < k; ++t)
                acc += (double)ar[t] * (double)(br[t] * global_scale);
            float r = (float)acc;
            if (c_f32)
                c_f32[(size_t)i * n + j] = r;
            c[(size_t)i * n + j] = a_kind ? po_f32_to_bf16(r) : po_f32_to_f16(r);
        }
    }
    free(af);
}

/* Whole reference CPU path in one call: dequant + matmul, as timed for
 * bench.py's cpu_baseline ("port").  fmt: 0 = NVFP4 (g=16), 1 = MXFP4 (g=32).
 * Returns 0, or -1 when the scratch allocation fails.                       */
int po_fp4_gemm_cpu(const uint16_t *a, int a_kind, const uint8_t *q,
                    const uint8_t *s, float global_scale, int fmt, int m, int n,
                    int k, uint16_t *c) {
    float *dq = (float *)malloc((size_t)n * k * sizeof(float));
    if (!dq)
        return -1;
    if (fmt == 0)
        po_dequant_nvfp4(q, s, n, k, dq);
    else
        po_dequant_mxfp4(q, s, n, k, dq);
    po_gemm_ref(a, a_kind, dq, global_scale, m, n, k, c, NULL);
    free(dq);
    return 0;
}

int po_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------
 * The reference's on-disk ("Petit") formats -- restated so that the
 * reference's own repack invariants can be checked without a GPU:
 *     petit_dequant(repack(x)) == native_dequant(x)         (bit exact)
 * lib/gemm/rocm/quantization/fp4/quantization_utils_fp4_test.cc:103-133.
 * --------------------------------------------------------------------- */

/* PetitFormat: per-u32 nibble re-encode.
 * lib/gemm/rocm/quantization/fp4/quantization_utils.cu:183-206.
 * Nibbles 0..3 go to the 0x8e positions of bytes [1,3,0,2] (sign at bit
 * 15,31,7,23; value three bits below the sign ... one above bit 0);
 * nibbles 4..7 are written bit-reversed into the 0x71 positions so that a
 * later bitreverse32 lands them in the 0x8e positions.  -0 becomes +0.      */
static uint32_t bitrev3(uint32_t v) { /* reverse a 3-bit field */
    return ((v & 1u) << 2) | (v & 2u) | ((v >> 2) & 1u);
}
uint32_t po_petit_format(uint32_t v) {
    /* sign-bit position per source nibble (derived from
     * the off_s / off_d arithmetic of :186-192) */
    static const int sign_lo[4] = {15, 31, 7, 23}; /* nibbles 0..3 */
    static const int sign_hi[4] = {16, 0, 24, 8};  /* nibbles 4..7 */
    uint32_t r = 0;
    for (int i = 0; i < 8; ++i) {
        uint32_t code = (v >> (4 * i)) & 0xfu;
        uint32_t mag = code & 7u;
        uint32_t sgn = mag ? (code >> 3) : 0u; /* -0 -> +0, :195-198 */
        if (i < 4)
            r |= (sgn << sign_lo[i]) | (mag << (sign_lo[i] - 6));
        else
            r |= (sgn << sign_hi[i - 4]) | (bitrev3(mag) << (sign_hi[i - 4] + 4));
    }
    return r;
}

static uint32_t bitrev32(uint32_t x) {
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0f0f0f0fu) | ((x & 0x0f0f0f0fu) << 4);
    x = ((x >> 8) & 0x00ff00ffu) | ((x & 0x00ff00ffu) << 8);
    return (x >> 16) | (x << 16);
}

/* Inverse view of a Petit word: the eight values fp4 * 2^-14 as the kernel
 * sees them.  lib/gemm/rocm/quantization/dequant.cuh:113-125 (Fp4ToBf8:
 * lo = q & 0x8e8e8e8e, hi = bitrev(q) & 0x8e8e8e8e) -- byte order within
 * lo is elements [2,0,3,1], within hi elements [6,4,7,5]; each byte read as
 * OCP e5m2 is fp4 * 2^-14.  out[i] = element i scaled back by 2^14.        */
void po_petit_word_decode(uint32_t w, float out[8]) {
    uint32_t lo = w & 0x8e8e8e8eu;
    uint32_t hi = bitrev32(w) & 0x8e8e8e8eu;
    static const int lo_elem[4] = {2, 0, 3, 1};
    static const int hi_elem[4] = {6, 4, 7, 5};
    for (int b = 0; b < 4; ++b) {
        out[lo_elem[b]] = ldexpf(po_e5m2_to_f32((uint8_t)(lo >> (8 * b))), 14);
        out[hi_elem[b]] = ldexpf(po_e5m2_to_f32((uint8_t)(hi >> (8 * b))), 14);
    }
}

/* Packed-weight location of native element (n, k), reference layout
 * RepackQWeightLayout64x32 (quantization_utils.cu:20-87,178; kernel
 * :208-253; launcher :729-746):  uint4 out[K/64][N/32][64 lanes],
 *   lane = ((k%64)/16)*16 + n%16,  word = 2*((n%32)/16) + (k%16)/8.
 * Returns the u32 index into the packed buffer.                             */
size_t po_petit_weight_word_index(int n_total, int n, int k) {
    size_t tile = (size_t)(k / 64) * (size_t)(n_total / 32) + (size_t)(n / 32);
    int lane = ((k % 64) / 16) * 16 + (n % 16);
    int word = 2 * ((n % 32) / 16) + ((k % 16) / 8);
    return (tile * 64 + (size_t)lane) * 4 + (size_t)word;
}

/* RepackNvFp4ToPetitFp4Weights (quantization_utils.cu:729-746).
 * in: u32 [n][k/8];  out: same byte count.  k % 64 == 0, n % 32 == 0 here
 * (the reference launches K/256 x N/32 blocks and silently drops remainders;
 * the restatement refuses them instead).                                    */
int po_petit_repack_weights(const uint32_t *in, int n, int k, uint32_t *out) {
    if (n % 32 || k % 64)
        return -1;
    for (int r = 0; r < n; ++r)
        for (int k8 = 0; k8 < k / 8; ++k8)
            out[po_petit_weight_word_index(n, r, k8 * 8)] =
                po_petit_format(in[(size_t)r * (k / 8) + k8]);
    return 0;
}

/* e4m3 -> "e5m3" byte: fp16 bits of (scale * 2^7) shifted right by 7.
 * quantization_utils.cu:143-162 (RepackScaleLayout::Transform) and
 * lib/tests/quantization.cc:29-38 (UpscaleFp8e4m3ToE5m3).  Valid for
 * non-negative, non-NaN scales only.            */
uint8_t po_e4m3_to_e5m3(uint8_t e4m3) {
    float f = po_e4m3_to_f32(e4m3) * 128.0f;
    return (uint8_t)((po_f32_to_f16(f) >> 7) & 0xff);
}
/* In-kernel decode of that byte: byte << 7 is an fp16 holding scale * 2^7
 * (dequant.cuh:161-164).  Returns the scale itself.                         */
float po_e5m3_to_f32(uint8_t b) {
    return po_f16_to_f32((uint16_t)((uint16_t)b << 7)) / 128.0f;
}

/* Packed NV-scale location, RepackScaleLayout64x32
 * (quantization_utils.cu:89-141,180; kernel :255-304; launcher :748-760):
 * u16 out[K/64][N/32][64], lane = ((k%64)/16)*16 + n%16, low byte = row
 * n%32 < 16, high byte = the row 16 below.  Returns the byte index.        */
size_t po_petit_nvscale_byte_index(int n_total, int n, int k) {
    size_t tile = (size_t)(k / 64) * (size_t)(n_total / 32) + (size_t)(n / 32);
    int lane = ((k % 64) / 16) * 16 + (n % 16);
    return (tile * 64 + (size_t)lane) * 2 + (size_t)((n % 32) / 16);
}
int po_petit_repack_nvscales(const uint8_t *in, int n, int k, uint8_t *out) {
    if (n % 64 || k % 64) /* kernel tiles 64(K) x 64(N): launcher :755 */
        return -1;
    for (int r = 0; r < n; ++r)
        for (int g = 0; g < k / 16; ++g)
            out[po_petit_nvscale_byte_index(n, r, g * 16)] =
                po_e4m3_to_e5m3(in[(size_t)r * (k / 16) + g]);
    return 0;
}

/* Packed MX-scale location, RepackMxScaleLayout64x32
 * (quantization_utils.cu:165-181; launcher :762-773; in-kernel fetch
 * warp_schedule_fp16.cuh:54-60): u16 out[K/64][N/32][32],
 * u16 index = h*16 + n%16 with h = (k%64)/32, low/high byte as above.      */
size_t po_petit_mxscale_byte_index(int n_total, int n, int k) {
    size_t tile = (size_t)(k / 64) * (size_t)(n_total / 32) + (size_t)(n / 32);
    int idx = ((k % 64) / 32) * 16 + (n % 16);
    return (tile * 32 + (size_t)idx) * 2 + (size_t)((n % 32) / 16);
}
int po_petit_repack_mxscales(const uint8_t *in, int n, int k, uint8_t *out) {
    if (n % 32 || k % 256) /* launcher grid (K/256, N/32), :768 */
        return -1;
    for (int r = 0; r < n; ++r)
        for (int g = 0; g < k / 32; ++g)
            out[po_petit_mxscale_byte_index(n, r, g * 32)] =
                in[(size_t)r * (k / 32) + g];
    return 0;
}

/* Dense dequant of the *packed* reference format (the CPU counterpart of
 * DequantPetitFp4 / DequantPetitMxFp4, quantization_utils.cu:673-727).
 * fmt 0: NV (e5m3 scales), 1: MX (raw e8m0).  out f32 [n][k].               */
int po_petit_dequant(const uint32_t *w, const uint8_t *s, int fmt, int n, int k,
                     float *out) {
    if (n % 32 || k % 64)
        return -1;
    for (int r = 0; r < n; ++r)
        for (int k8 = 0; k8 < k / 8; ++k8) {
            float v[8];
            po_petit_word_decode(w[po_petit_weight_word_index(n, r, k8 * 8)], v);
            float sc;
            if (fmt == 0)
                sc = po_e5m3_to_f32(s[po_petit_nvscale_byte_index(n, r, k8 * 8)]);
            else
                sc = po_e8m0_to_f32(s[po_petit_mxscale_byte_index(n, r, k8 * 8)]);
            for (int i = 0; i < 8; ++i)
                out[(size_t)r * k + (size_t)k8 * 8 + i] = v[i] * sc;
        }
    return 0;
}
