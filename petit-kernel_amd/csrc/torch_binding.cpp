// torch_binding.cpp -- the compiled torch operator layer: `torch.ops.petit_kernel.*` over the C ABI of libpetit_amd.so.
//
// The reference ships its operators as a compiled ATen extension (lib/pybind/fp4.cc:38-283, pybind.cc:8-26); this is the
// same thing for the gfx950 build, registered through torch.library (TORCH_LIBRARY) instead of pybind so that the ops
// are visible to the dispatcher (torch.compile / graph capture see them as opaque custom ops).  Same checks, same
// error texts, same output shapes and dtypes as petit_kernel/ops.py (the ctypes layer), which stays as the
// dependency-free binding; both end in the same C entry points, there is no other compute path.
// torch is used for device memory, the current stream and the per-call scratch allocation only.
//
// Built by petit-kernel_amd/build.py with the host compiler (no device code here) into lib/libpetit_torch.so.
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h> // (torch-ROCm devices are "cuda" devices: the masquerading forms)
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/library.h>

#include <string>

#include "../../include/petit_amd.h"

namespace {

constexpr int64_t kLayoutN = 16, kLayoutM = 128, kPack = 8; // fp4.cc:17-19
constexpr int kCxxFp4 = 3, kCxxFp16 = 4, kCxxBf16 = 5, kCxxMxFp4 = 7; // quantization/types.h:4-13

void *stream_of(const at::Tensor &t) { return c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream(); }

at::Tensor repack_nvfp4(const at::Tensor &b_q_weight, int64_t size_n, int64_t size_k) {
    TORCH_CHECK(size_k % kLayoutM == 0, "size_k = ", size_k, " is not divisible by tile_k_size = ", kLayoutM);
    TORCH_CHECK(size_n % kLayoutN == 0, "size_n = ", size_n, " is not divisible by tile_n_size = ", kLayoutN);
    TORCH_CHECK(b_q_weight.dim() == 2 && size_k / kPack == b_q_weight.size(1), "Shape mismatch: b_q_weight.size(1) = ",
                b_q_weight.size(-1), ", size_k = ", size_k, ", pack_factor = ", kPack);
    TORCH_CHECK(b_q_weight.size(0) == size_n, "b_q_weight.size(0) = ", b_q_weight.size(0), " is not size_n = ", size_n);
    TORCH_CHECK(b_q_weight.is_cuda(), "b_q_weight is not on GPU");
    TORCH_CHECK(b_q_weight.is_contiguous(), "b_q_weight is not contiguous");
    TORCH_CHECK(b_q_weight.scalar_type() == at::kInt, "b_q_weight type is not kInt");
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(b_q_weight.device());
    at::Tensor out = at::empty({size_n / kLayoutN, size_k * kLayoutN / kPack}, b_q_weight.options());
    const int rc = petit_repack_nvfp4_weights((unsigned *)out.data_ptr(), (const unsigned *)b_q_weight.data_ptr(), (unsigned)size_k,
                                              (unsigned)size_n, stream_of(out));
    TORCH_CHECK(rc == PETIT_OK, "repack_nvfp4: ", petit_error_string(rc));
    return out;
}

at::Tensor process_scales(const at::Tensor &scales, int64_t size_n, int64_t size_k, bool mx) {
    const int64_t group = mx ? 32 : 16;
    TORCH_CHECK(size_k % (2 * kLayoutM) == 0, "size_k = ", size_k, " is not divisible by tile_k_size = ", 2 * kLayoutM);
    TORCH_CHECK(size_n % kLayoutN == 0, "size_n = ", size_n, " is not divisible by tile_n_size = ", kLayoutN);
    TORCH_CHECK(scales.dim() == 2 && scales.size(1) > 0 && size_k / scales.size(1) == group && size_k % scales.size(1) == 0,
                "Only groupsize = ", group, " is supported.");
    TORCH_CHECK(scales.size(0) == size_n, "scales.size(0) = ", scales.size(0), " is not size_n = ", size_n);
    TORCH_CHECK(scales.is_cuda(), "scales is not on GPU");
    TORCH_CHECK(scales.is_contiguous(), "scales is not contiguous");
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(scales.device());
    at::Tensor out;
    int rc;
    if (mx) {
        TORCH_CHECK(scales.scalar_type() == at::kByte, "scales type is not uint8");
        TORCH_CHECK(size_n % 32 == 0, "size_n = ", size_n, " is not divisible by the MX scale tile (32)");
        out = at::empty({size_n / 32, size_k}, scales.options());
        rc = petit_repack_mxfp4_scales((unsigned *)out.data_ptr(), (const unsigned *)scales.data_ptr(), (unsigned)size_k, (unsigned)size_n,
                                       stream_of(out));
    } else {
        TORCH_CHECK(scales.scalar_type() == at::kFloat8_e4m3fn, "scales type is not float8_e4m3fn");
        out = at::empty({scales.size(0), scales.size(1)}, scales.options());
        rc = petit_repack_nvfp4_scales((unsigned *)out.data_ptr(), (const unsigned *)scales.data_ptr(), (unsigned)size_k, (unsigned)size_n,
                                       stream_of(out));
    }
    TORCH_CHECK(rc == PETIT_OK, mx ? "process_mxfp4_scales: " : "process_nvfp4_scales: ", petit_error_string(rc));
    return out;
}
at::Tensor process_nvfp4_scales(const at::Tensor &s, int64_t n, int64_t k) { return process_scales(s, n, k, false); }
at::Tensor process_mxfp4_scales(const at::Tensor &s, int64_t n, int64_t k) { return process_scales(s, n, k, true); }

// activation: 0 none, 1 silu_mul (PETIT_ACTIVATION_*)
at::Tensor mul_a16(bool mx, const at::Tensor &A, const at::Tensor &B, const at::Tensor &s, const at::Tensor &global_scale, int64_t size_m,
                   int64_t size_n, int64_t size_k, int64_t solution_id, const std::optional<at::Tensor> &bias, int64_t activation) {
    // (check order as in the reference's MulNvFp4A16 / MulMxFp4A16, fp4.cc:163-260: the scale / weight tensor contracts first)
    if (mx) {
        TORCH_CHECK(B.dim() == 2 && B.size(0) == size_n / kLayoutN, "B.size(0) = ", B.size(0), " is not size_n / 16 = ", size_n / kLayoutN);
        TORCH_CHECK(B.size(1) == size_k * kLayoutN / kPack, "B.size(1) = ", B.size(1), " is not packed size = ", size_k * kLayoutN / kPack);
        TORCH_CHECK(s.dim() == 2 && s.size(0) == size_n / 32, "s.size(0) = ", s.size(0), " is not size_n / 32 = ", size_n / 32);
        TORCH_CHECK(s.size(1) == size_k, "s.size(1) = ", s.size(1), " is not size_k = ", size_k);
    } else {
        TORCH_CHECK(s.dim() == 2 && s.size(1) != 0 && size_k / s.size(1) == 16, "Only groupsize = 16 is supported. size_k = ", size_k,
                    ", s.size(1) = ", s.size(-1));
        TORCH_CHECK(s.numel() == size_n * size_k / 16, "s does not hold size_n * size_k / 16 scales");
    }
    TORCH_CHECK(A.scalar_type() == at::kBFloat16 || A.scalar_type() == at::kHalf, "A must be bfloat16 or float16.");
    TORCH_CHECK(A.is_cuda() && B.is_cuda() && s.is_cuda() && global_scale.is_cuda(), "all tensors must be on GPU");
    TORCH_CHECK(A.is_contiguous() && A.numel() == size_m * size_k, "A must be a contiguous [size_m, size_k] tensor");
    TORCH_CHECK(B.is_contiguous() && B.numel() * B.element_size() == size_n * size_k / 2, "B does not hold size_n * size_k packed 4-bit weights");
    TORCH_CHECK(global_scale.scalar_type() == at::kFloat && global_scale.numel() >= 1, "global_scale must be float32");
    TORCH_CHECK(activation == 0 || activation == 1, "activation must be 0 (none) or 1 (silu_mul)");
    if (activation)
        TORCH_CHECK(size_n % 32 == 0, "silu_mul needs size_n % 32 == 0 (gate / up halves of whole tiles), got ", size_n);
    if (bias.has_value())
        TORCH_CHECK(bias->is_cuda() && bias->device() == A.device() && bias->scalar_type() == A.scalar_type() && bias->is_contiguous() &&
                        bias->numel() == size_n,
                    "bias must be a contiguous [size_n] tensor of A's dtype on A's device");
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(A.device());
    at::Tensor c = at::empty({size_m, activation ? size_n / 2 : size_n}, A.options());
    const int a_type = A.scalar_type() == at::kBFloat16 ? kCxxBf16 : kCxxFp16;
    const petit_solution_hints hints{a_type, mx ? kCxxMxFp4 : kCxxFp4, a_type, 0};
    // ids are 64-bit patterns whose top nibble is the K split: a split of 8..15 sets bit 63, and the schema's `int` is a
    // signed int64 -- such an id arrives as its two's-complement value (petit_kernel/compiled.py maps it) and is
    // reinterpreted here.  A small negative value is "library default", as in the reference (fp4.cc:189-191,240: solution_id < 0) -- including
    // -2 / -3, which name the native class only on ITS entry point (mul_mxfp4_native): these two ops are the reference's, and exact
    uint64_t sid = (solution_id < 0 && solution_id >= -4096) ? PETIT_SOLUTION_AUTO : (uint64_t)solution_id;
    // NVFP4 weights with an MFMA-native image attached (petit_nvfp4_native_attach) have opted into the native class: -2 / -3 / -4 name it then
    if (!mx && solution_id <= -2 && solution_id >= -4 && petit_nvfp4_native_attached(B.data_ptr()))
        sid = solution_id == -2 ? PETIT_SOLUTION_AUTO_NATIVE_MXFP8 : solution_id == -3 ? PETIT_SOLUTION_AUTO_NATIVE_MXFP4 : PETIT_SOLUTION_AUTO_NATIVE_MXFP6;
    const petit_epilogue epi{bias.has_value() ? bias->data_ptr() : nullptr, (int32_t)activation, 0};
    // per-call scratch from the caching allocator (stream-ordered, capture-safe): K-split slabs / native-FP4 activations
    const uint64_t ws_bytes = petit_gemm_workspace_bytes_ex(&hints, (unsigned)size_m, (unsigned)size_n, (unsigned)size_k, sid,
                                                            (bias.has_value() || activation) ? &epi : nullptr);
    at::Tensor ws;
    if (ws_bytes)
        ws = at::empty({(int64_t)ws_bytes}, A.options().dtype(at::kByte));
    auto fn = mx ? petit_gemm_mxfp4_fp16_grid_ws : petit_gemm_fp4_fp16_grid_ws;
    const int rc = fn((unsigned *)c.data_ptr(), (const unsigned *)A.data_ptr(), (const unsigned *)B.data_ptr(), (const unsigned *)s.data_ptr(),
                      (const float *)global_scale.data_ptr(), (unsigned)size_m, (unsigned)size_n, (unsigned)size_k, &hints, sid,
                      (bias.has_value() || activation) ? &epi : nullptr, ws_bytes ? ws.data_ptr() : nullptr, ws_bytes, stream_of(A));
    TORCH_CHECK(rc != PETIT_ERROR_PROBLEM_SHAPE, "Incompatible problem shape (m=", size_m, ", n=", size_n, ", k=", size_k, ")");
    TORCH_CHECK(rc != PETIT_ERROR_KERNEL_SHAPE, "No kernel implementation for solution_id=", sid == PETIT_SOLUTION_AUTO ? "-1" : std::to_string((int64_t)sid), ".");
    TORCH_CHECK(rc == PETIT_OK, mx ? "mul_mxfp4_a16: " : "mul_nvfp4_a16: ", petit_error_string(rc));
    return c;
}
at::Tensor mul_nvfp4_a16(const at::Tensor &A, const at::Tensor &B, const at::Tensor &s, const at::Tensor &gs, int64_t m, int64_t n, int64_t k,
                         int64_t solution_id, const std::optional<at::Tensor> &bias, int64_t activation) {
    return mul_a16(false, A, B, s, gs, m, n, k, solution_id, bias, activation);
}
at::Tensor mul_mxfp4_a16(const at::Tensor &A, const at::Tensor &B, const at::Tensor &s, const at::Tensor &gs, int64_t m, int64_t n, int64_t k,
                         int64_t solution_id, const std::optional<at::Tensor> &bias, int64_t activation) {
    return mul_a16(true, A, B, s, gs, m, n, k, solution_id, bias, activation);
}

// Shape functions for the Meta key (FakeTensor / torch.compile tracing, torch.export): outputs of the right shape, dtype and
// device, nothing launched -- the ops trace as opaque calls instead of breaking the graph.
at::Tensor repack_nvfp4_meta(const at::Tensor &q, int64_t n, int64_t k) { return at::empty({n / kLayoutN, k * kLayoutN / kPack}, q.options()); }
at::Tensor process_nvfp4_scales_meta(const at::Tensor &s, int64_t n, int64_t k) { return at::empty({n, k / 16}, s.options()); }
at::Tensor process_mxfp4_scales_meta(const at::Tensor &s, int64_t n, int64_t k) { return at::empty({n / 32, k}, s.options()); }
at::Tensor mul_a16_meta(const at::Tensor &A, const at::Tensor &, const at::Tensor &, const at::Tensor &, int64_t m, int64_t n, int64_t, int64_t,
                        const std::optional<at::Tensor> &, int64_t activation) {
    TORCH_CHECK(A.scalar_type() == at::kBFloat16 || A.scalar_type() == at::kHalf, "A must be bfloat16 or float16.");
    return at::empty({m, activation ? n / 2 : n}, A.options());
}

} // namespace

// Schemas first, then one implementation per backend key.  The CPU key gets the SAME functions as the GPU key: a CPU
// tensor then reaches the checks above and gets the reference's own error text ("... is not on GPU", fp4.cc:50-52)
// instead of a dispatcher message.  (torch-ROCm devices dispatch on the CUDA key.)
TORCH_LIBRARY(petit_kernel, m) {
    m.def("repack_nvfp4(Tensor b_q_weight, int size_n, int size_k) -> Tensor");
    m.def("process_nvfp4_scales(Tensor scales, int size_n, int size_k) -> Tensor");
    m.def("process_mxfp4_scales(Tensor scales, int size_n, int size_k) -> Tensor");
    m.def("mul_nvfp4_a16(Tensor A, Tensor B, Tensor s, Tensor global_scale, int size_m, int size_n, int size_k, int solution_id, "
          "Tensor? bias=None, int activation=0) -> Tensor");
    m.def("mul_mxfp4_a16(Tensor A, Tensor B, Tensor s, Tensor global_scale, int size_m, int size_n, int size_k, int solution_id, "
          "Tensor? bias=None, int activation=0) -> Tensor");
    // round 3's op name (scales promised inside fp16's range): an alias of mul_mxfp4_a16 for one more round -- the kernels test the range themselves
    m.def("mul_mxfp4_a16_f16range(Tensor A, Tensor B, Tensor s, Tensor global_scale, int size_m, int size_n, int size_k, int solution_id, "
          "Tensor? bias=None, int activation=0) -> Tensor");
}
#define PETIT_IMPL_REAL(m)                                      \
    m.impl("repack_nvfp4", &repack_nvfp4);                      \
    m.impl("process_nvfp4_scales", &process_nvfp4_scales);      \
    m.impl("process_mxfp4_scales", &process_mxfp4_scales);      \
    m.impl("mul_nvfp4_a16", &mul_nvfp4_a16);                    \
    m.impl("mul_mxfp4_a16", &mul_mxfp4_a16);                    \
    m.impl("mul_mxfp4_a16_f16range", &mul_mxfp4_a16);
TORCH_LIBRARY_IMPL(petit_kernel, CUDA, m) { PETIT_IMPL_REAL(m) }
TORCH_LIBRARY_IMPL(petit_kernel, CPU, m) { PETIT_IMPL_REAL(m) }
TORCH_LIBRARY_IMPL(petit_kernel, Meta, m) {
    m.impl("repack_nvfp4", &repack_nvfp4_meta);
    m.impl("process_nvfp4_scales", &process_nvfp4_scales_meta);
    m.impl("process_mxfp4_scales", &process_mxfp4_scales_meta);
    m.impl("mul_nvfp4_a16", &mul_a16_meta);
    m.impl("mul_mxfp4_a16", &mul_a16_meta);
    m.impl("mul_mxfp4_a16_f16range", &mul_a16_meta);
}
