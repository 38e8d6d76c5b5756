// examples/cxx_gemm.cc -- a C++ caller of the reference's namespace API (llama.cpp style), served by
// libpetit_amd.so through include/causalflow/petit/gemm.h.  Reads a problem written by
// tests/test_gpu_parity.py::test_cxx_api_end_to_end, repacks on the device, runs the GEMM, writes C.
//
//   g++ -std=c++17 -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ examples/cxx_gemm.cc \
//       petit-kernel_amd/lib/libpetit_amd.so -L/opt/rocm/lib -lamdhip64 -o cxx_gemm
//   ./cxx_gemm problem.bin out.bin
//
// problem.bin: u32 {kind (0 nv, 1 mx), is_bf16, m, n, k}, f32 global_scale, then A (m*k u16), native
// weights (n*k/2 bytes), native scales (n*k/16 e4m3 or n*k/32 e8m0 bytes).
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "causalflow/petit/gemm.h"

using namespace causalflow::petit::rocm::quantization;

#define HIP_OK(x)                                                              \
    do {                                                                       \
        if ((x) != hipSuccess) {                                               \
            std::fprintf(stderr, "HIP error at %s:%d\n", __FILE__, __LINE__);  \
            return 2;                                                          \
        }                                                                      \
    } while (0)

int main(int argc, char **argv) {
    if (argc != 3)
        return 64;
    FILE *f = std::fopen(argv[1], "rb");
    if (!f)
        return 66;
    unsigned hdr[5];
    float gs;
    if (std::fread(hdr, 4, 5, f) != 5 || std::fread(&gs, 4, 1, f) != 1)
        return 65;
    const unsigned kind = hdr[0], is_bf16 = hdr[1], m = hdr[2], n = hdr[3], k = hdr[4];
    const size_t a_bytes = (size_t)m * k * 2, w_bytes = (size_t)n * k / 2, s_bytes = (size_t)n * k / (kind ? 32 : 16);
    std::vector<unsigned char> a(a_bytes), w(w_bytes), s(s_bytes), c((size_t)m * n * 2);
    if (std::fread(a.data(), 1, a_bytes, f) != a_bytes || std::fread(w.data(), 1, w_bytes, f) != w_bytes ||
        std::fread(s.data(), 1, s_bytes, f) != s_bytes)
        return 65;
    std::fclose(f);

    unsigned *d_a, *d_w, *d_s, *d_pw, *d_ps, *d_c;
    float *d_gs;
    HIP_OK(hipMalloc((void **)&d_a, a_bytes));
    HIP_OK(hipMalloc((void **)&d_w, w_bytes));
    HIP_OK(hipMalloc((void **)&d_s, s_bytes));
    HIP_OK(hipMalloc((void **)&d_pw, w_bytes));
    HIP_OK(hipMalloc((void **)&d_ps, s_bytes));
    HIP_OK(hipMalloc((void **)&d_c, c.size()));
    HIP_OK(hipMalloc((void **)&d_gs, 4));
    HIP_OK(hipMemcpy(d_a, a.data(), a_bytes, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_w, w.data(), w_bytes, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_s, s.data(), s_bytes, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_gs, &gs, 4, hipMemcpyHostToDevice));
    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));

    // the reference's call sequence: repack once (fp4/quantization_utils.cu:729-773), then GEMMs
    fp4::RepackNvFp4ToPetitFp4Weights(d_pw, d_w, /*in_chan=*/k, /*out_chan=*/n, stream);
    if (kind == 0)
        fp4::RepackNvFp4ToPetitFp4Scales(d_ps, d_s, k, n, stream);
    else
        fp4::RepackMxFp4ToPetitFp4Scales(d_ps, d_s, k, n, stream);

    const DataType at = is_bf16 ? kDataTypeBf16 : kDataTypeFp16;
    PetitSolutionHints hints{at, kind ? kDataTypeMxFp4e2m1 : kDataTypeFp4e2m1, at, false};
    unsigned n_sols = 0;
    if (fp4::GemmGetSolutions(hints, m, n, k, nullptr, &n_sols) != 0 || n_sols == 0)
        return 3;
    std::vector<SolutionId> sols(n_sols);
    if (fp4::GemmGetSolutions(hints, m, n, k, sols.data(), &n_sols) != 0)
        return 3;
    // once with the library's choice, once with an explicit id from the enumeration (last one wins the output)
    int err = kind ? fp4::GemmMxFp4Fp16Grid(d_c, d_a, d_pw, d_ps, d_gs, m, n, k, hints, -1ul, stream)
                   : fp4::GemmFp4Fp16Grid(d_c, d_a, d_pw, d_ps, d_gs, m, n, k, hints, -1ul, stream);
    if (err)
        return 10 + err;
    const unsigned long id = sols[n_sols / 2].Repr();
    err = kind ? fp4::GemmMxFp4Fp16Grid(d_c, d_a, d_pw, d_ps, d_gs, m, n, k, hints, id, stream)
               : fp4::GemmFp4Fp16Grid(d_c, d_a, d_pw, d_ps, d_gs, m, n, k, hints, id, stream);
    if (err)
        return 20 + err;
    // an id nobody enumerated must be refused with the reference's code (gemm.h:108)
    if (fp4::GemmFp4Fp16Grid(d_c, d_a, d_pw, d_ps, d_gs, m, n, k, hints, 0x1234ul, stream) != kErrorKernelShape)
        return 30;
    HIP_OK(hipStreamSynchronize(stream));
    HIP_OK(hipMemcpy(c.data(), d_c, c.size(), hipMemcpyDeviceToHost));
    FILE *o = std::fopen(argv[2], "wb");
    if (!o || std::fwrite(c.data(), 1, c.size(), o) != c.size())
        return 73;
    std::fclose(o);
    std::printf("ok: %u solutions, ran id 0x%lx (element_b %u, mfma_type %u)\n", n_sols, id,
                (unsigned)sols[n_sols / 2].element_b, (unsigned)sols[n_sols / 2].mfma_type);
    return 0;
}
