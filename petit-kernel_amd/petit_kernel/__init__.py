"""petit_kernel -- drop-in Python surface of the MI355X (gfx950) build.

Same names, signatures and tensor contracts as the reference package
(petit_kernel/__init__.py:8-79 of causalflow-ai/petit-kernel v0.0.3):
SGLang / vLLM call sites (`repack_nvfp4`, `process_nvfp4_scales`,
`mul_nvfp4_a16(..., solution_id=-1)`) work unchanged.  The packed tensors
returned by `repack_*` / `process_*_scales` keep the reference's shapes and
dtypes but use the gfx950 layout (petit-kernel_amd/csrc/layout.h); they are
opaque and only meaningful to `mul_*_a16` of this build.
"""
import enum

import torch

from . import compiled, offline, ops, tuning
from .ops import QuantizedActivations, mul_fp4_a16_grouped, mul_mxfp4_native, quantize_activations
from .ops import attach_nvfp4_native, mul_nvfp4_native, nvfp4_native_image
from .tuning import tune, tune_tensors
from .ops import SOLUTION_AUTO, SOLUTION_AUTO_NATIVE_MXFP4, SOLUTION_AUTO_NATIVE_MXFP6, SOLUTION_AUTO_NATIVE_MXFP8, PetitSolutionHints
from ._lib import MXFP4_F16RANGE_SCALE_MAX, MXFP4_F16RANGE_SCALE_MIN

# operator layer: the compiled torch.library binding when it is built and loads (csrc/torch_binding.cpp), else the
# ctypes layer; both are thin shims over the same C ABI (there is no other compute path)
_impl = compiled if compiled.available() else ops


class DataType(enum.Enum):
    # numbering of the reference's Python enum (petit_kernel/__init__.py:8-15)
    int4 = 0
    float8_e4m3fn = 1
    float4_e2m1 = 2
    float16 = 3
    bfloat16 = 4
    float8_e5m2fn = 5
    mxfloat4_e2m1 = 6


# Round 3's extension value for PetitSolutionHints.b_type ("MXFP4 whose every e8m0 scale byte lies in 114..140"): still accepted, and means plain
# MXFP4 -- the fp16 x MXFP4 kernels test the range themselves (include/petit_amd.h), there is nothing to promise any more.
DTYPE_MXFP4_E2M1_F16RANGE = 8


def repack_nvfp4(qw: torch.Tensor, size_n: int, size_k: int) -> torch.Tensor:
    return _impl.repack_nvfp4(qw, size_n, size_k)


def process_nvfp4_scales(scales: torch.Tensor, size_n: int, size_k: int) -> torch.Tensor:
    return _impl.process_nvfp4_scales(scales, size_n, size_k)


def repack_mxfp4(qw: torch.Tensor, size_n: int, size_k: int) -> torch.Tensor:
    # weight packing is format-independent, as in the reference (:27-28)
    return _impl.repack_nvfp4(qw, size_n, size_k)


def process_mxfp4_scales(scales: torch.Tensor, size_n: int, size_k: int) -> torch.Tensor:
    # (no look at the values, no host sync, nothing attached to the tensor: the GEMM kernels decide per wave and span whether the scales they
    # hold allow the single-MFMA fp16 body -- nn.Parameter-wrapped, copied or overwritten scale tensors all get the right kernel)
    return _impl.process_mxfp4_scales(scales, size_n, size_k)


def mxfp4_scales_in_fp16_range(raw_scales: torch.Tensor) -> bool:
    """A diagnostic, not an input of any call: True when every e8m0 byte of the RAW (unprocessed) MXFP4 scale tensor lies in 114..140, i.e. when
    fp16 activations run the single-MFMA body in every wave (a wave that meets a byte outside that range finishes its K range in the exact
    fallback body, which is slower).  Synchronises with the device."""
    if raw_scales.numel() == 0:
        return False
    lo, hi = torch.aminmax(raw_scales)
    return bool(int(lo) >= MXFP4_F16RANGE_SCALE_MIN and int(hi) <= MXFP4_F16RANGE_SCALE_MAX)


def mul_nvfp4_a16(a: torch.Tensor, b: torch.Tensor, s: torch.Tensor, global_scale: torch.Tensor,
                  size_m: int, size_n: int, size_k: int, solution_id: int = -1, *, bias: torch.Tensor = None,
                  activation: str = None) -> torch.Tensor:
    # `bias` / `activation` are extensions (keyword-only, default None = the reference's behaviour):
    #   bias       [size_n] of a.dtype, added before the single rounding to 16 bit
    #   activation "silu_mul": returns [size_m, size_n/2] = silu(y[:, :n/2]) * y[:, n/2:]  (gate_up of a gated MLP)
    return _impl.mul_nvfp4_a16(a, b, s, global_scale, size_m, size_n, size_k, solution_id, bias, activation)


def mul_mxfp4_a16(a: torch.Tensor, b: torch.Tensor, s: torch.Tensor, global_scale: torch.Tensor,
                  size_m: int, size_n: int, size_k: int, solution_id: int = -1, *, bias: torch.Tensor = None,
                  activation: str = None, f16_range: bool = None) -> torch.Tensor:
    # fp16 activations are an extension (the reference's MXFP4 path takes bf16 only, gemm_fp4_fp16_grid.cc:55-64); any negative solution_id
    # is the library default as in the reference (fp4.cc:240) -- the native class is reached through mul_mxfp4_native only.
    # f16_range: round 3's promise "every block scale lies in 114..140"; ignored since round 4 (the kernels test it), accepted for one more round
    if f16_range is not None:
        import warnings
        warnings.warn("mul_mxfp4_a16(f16_range=...) is ignored: the fp16 x MXFP4 kernels test the scale range themselves", DeprecationWarning, stacklevel=2)
    return _impl.mul_mxfp4_a16(a, b, s, global_scale, size_m, size_n, size_k, solution_id, bias, activation)


def get_fp4_solutions(size_m: int, size_n: int, size_k: int, a_type, c_type) -> list:
    return ops.get_fp4_solutions(size_m, size_n, size_k, a_type, c_type)


__all__ = [
    "repack_nvfp4",
    "repack_mxfp4",
    "process_nvfp4_scales",
    "process_mxfp4_scales",
    "mul_nvfp4_a16",
    "mul_mxfp4_a16",
    "get_fp4_solutions",
    "mxfp4_scales_in_fp16_range",
    "DataType",
    "PetitSolutionHints",
    "tune",
    "tune_tensors",
    "quantize_activations",
    "mul_fp4_a16_grouped",
    "mul_mxfp4_native",
    "nvfp4_native_image",
    "attach_nvfp4_native",
    "mul_nvfp4_native",
    "QuantizedActivations",
    "SOLUTION_AUTO",
    "SOLUTION_AUTO_NATIVE_MXFP8",
    "SOLUTION_AUTO_NATIVE_MXFP4",
    "SOLUTION_AUTO_NATIVE_MXFP6",
]
