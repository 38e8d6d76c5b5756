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
from .tuning import tune, tune_tensors
from .ops import SOLUTION_AUTO, SOLUTION_AUTO_NATIVE_MXFP4, SOLUTION_AUTO_NATIVE_MXFP8, PetitSolutionHints
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


# Extension, as PetitSolutionHints.b_type (a raw value of the C++ numbering; the reference's Python enum above stays as it is): MXFP4 whose every
# e8m0 scale byte lies in 114..140 (PETIT_DTYPE_MXFP4_E2M1_F16RANGE, include/petit_amd.h) -- enumerates / resolves the fp16 single-MFMA family.
DTYPE_MXFP4_E2M1_F16RANGE = 8


def repack_nvfp4(qw: torch.Tensor, size_n: int, size_k: int) -> torch.Tensor:
    return _impl.repack_nvfp4(qw, size_n, size_k)


def process_nvfp4_scales(scales: torch.Tensor, size_n: int, size_k: int) -> torch.Tensor:
    return _impl.process_nvfp4_scales(scales, size_n, size_k)


def repack_mxfp4(qw: torch.Tensor, size_n: int, size_k: int) -> torch.Tensor:
    # weight packing is format-independent, as in the reference (:27-28)
    return _impl.repack_nvfp4(qw, size_n, size_k)


def process_mxfp4_scales(scales: torch.Tensor, size_n: int, size_k: int) -> torch.Tensor:
    out = _impl.process_mxfp4_scales(scales, size_n, size_k)
    # One look at the raw scales, at load time: do all e8m0 bytes lie in 114..140 (e2m1 x scale a normal fp16 number)?  Real checkpoints do.
    # mul_mxfp4_a16 reads the mark off THIS tensor object and then gives fp16 activations the single-MFMA family
    # (PETIT_DTYPE_MXFP4_E2M1_F16RANGE, include/petit_amd.h); a tensor without the mark (re-wrapped, copied, traced) takes the exact split path.
    try:
        if not torch.compiler.is_compiling() and scales.is_cuda:
            out.petit_scales_in_fp16_range = mxfp4_scales_in_fp16_range(scales)
    except Exception:  # noqa: BLE001  (fake tensors, exotic subclasses: no mark, the exact path)
        pass
    return out


def mxfp4_scales_in_fp16_range(raw_scales: torch.Tensor) -> bool:
    """True when every e8m0 byte of the RAW (unprocessed) MXFP4 scale tensor lies in 114..140 -- the condition under which fp16 activations may
    take the single-MFMA family (PETIT_DTYPE_MXFP4_E2M1_F16RANGE).  For pipelines that process scales offline (petit_kernel.offline) and load the
    packed tensors later: record this bit next to them and pass it as mul_mxfp4_a16(..., scales_in_fp16_range=bit)."""
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
                  activation: str = None, scales_in_fp16_range: bool = None) -> torch.Tensor:
    # scales_in_fp16_range (extension, fp16 activations only): every e8m0 scale byte in 114..140 -- None = the mark
    # process_mxfp4_scales left on `s` (False when there is none); True is the caller's promise, False forces the exact split path.
    if scales_in_fp16_range is None:   # (explicit kernel ids keep the plain MXFP4 meaning they were enumerated with)
        scales_in_fp16_range = solution_id == -1 and getattr(s, "petit_scales_in_fp16_range", False)
    fast = bool(scales_in_fp16_range) and a.dtype == torch.float16
    if fast:
        return _impl.mul_mxfp4_a16(a, b, s, global_scale, size_m, size_n, size_k, solution_id, bias, activation, True)
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
    "QuantizedActivations",
    "SOLUTION_AUTO",
    "SOLUTION_AUTO_NATIVE_MXFP8",
    "SOLUTION_AUTO_NATIVE_MXFP4",
]
