// probes.hip -- tiny hardware-semantics probes run once on the MI355X box; results are
// recorded in DESIGN.md.  Not part of the product library.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;

// out[i*2..] = cvt_scalef32_pk_f32_fp4(word, scale[i]) for byte 0
__global__ void probe_cvt_scale(const unsigned *words, const float *scales, float *out_f32, unsigned *out_bf16, int n) {
    int i = threadIdx.x;
    if (i >= n) return;
    f32x2 f = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(words[i], scales[i], 0);
    out_f32[2 * i] = f.x;
    out_f32[2 * i + 1] = f.y;
    bf16x2 b = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(words[i], scales[i], 0);
    out_bf16[i] = __builtin_bit_cast(unsigned, b);
}

// raw buffer bounds semantics: is soffset part of the range check?
__global__ void probe_buffer_oob(const unsigned *buf, unsigned num_bytes, unsigned voff, unsigned soff, unsigned *out) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)buf, 0, (int)num_bytes, 0x00020000);
    out[threadIdx.x] = __builtin_amdgcn_raw_buffer_load_b32(r, voff + threadIdx.x * 4, soff, 0);
}

extern "C" void run_probe_cvt_scale(const unsigned *w, const float *s, float *of, unsigned *ob, int n, void *stream) {
    hipLaunchKernelGGL(probe_cvt_scale, dim3(1), dim3(64), 0, (hipStream_t)stream, w, s, of, ob, n);
}
extern "C" void run_probe_buffer_oob(const unsigned *buf, unsigned num_bytes, unsigned voff, unsigned soff, unsigned *out, void *stream) {
    hipLaunchKernelGGL(probe_buffer_oob, dim3(1), dim3(64), 0, (hipStream_t)stream, buf, num_bytes, voff, soff, out);
}
