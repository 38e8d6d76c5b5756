"""petit_kernel.compiled -- the compiled operator layer: `torch.ops.petit_kernel.*` (csrc/torch_binding.cpp,
lib/libpetit_torch.so), the counterpart of the reference's ATen extension (lib/pybind/fp4.cc, pybind.cc:8-26).

Same functions and argument order as petit_kernel.ops (the ctypes layer); both call the same C ABI of libpetit_amd.so.
The package front end (petit_kernel/__init__.py) uses this layer when the library is present (it costs a third of
the ctypes layer's host time per call, see bench.py `host_us_per_call`) and the ctypes layer otherwise;
$PETIT_AMD_BINDING=ctypes|compiled forces one.
"""
from __future__ import annotations

import os
from pathlib import Path

import torch

from . import _lib  # loads libpetit_amd.so first (the binding links against it)

LIB_PATH = Path(_lib.LIB_PATH).parent / "libpetit_torch.so"
_loaded = False
_error = None


def available() -> bool:
    """True once lib/libpetit_torch.so is loaded into torch's dispatcher."""
    global _loaded, _error
    if _loaded:
        return True
    if _error is not None or os.environ.get("PETIT_AMD_BINDING", "") == "ctypes":
        return False
    if not LIB_PATH.exists():
        _error = f"{LIB_PATH} not built (python petit-kernel_amd/build.py)"
        return False
    try:
        torch.ops.load_library(str(LIB_PATH))
        _loaded = True
    except Exception as exc:  # noqa: BLE001 -- e.g. a torch ABI mismatch: the ctypes layer still works
        _error = repr(exc)
    return _loaded


def why_unavailable() -> str:
    return _error or ""


_ACT = {None: 0, "none": 0, "silu_mul": 1}


def _act(activation) -> int:
    if activation not in _ACT:
        raise RuntimeError(f"activation must be one of {sorted(k for k in _ACT if k)} or None")
    return _ACT[activation]


def _sid(solution_id: int) -> int:
    """Python id -> the signed 64-bit value the op schema carries (`int` = int64).  Ids are unsigned 64-bit patterns with
    the K split in bits 60-63, so a split of 8..15 does not fit a signed int64 as such: it crosses as its two's-complement
    value and the binding reinterprets it.  Any negative id means "library default" (reference: fp4.cc:189-191)."""
    solution_id = int(solution_id)
    if solution_id < 0:
        # -2 / -3 / -4 cross as they are: the binding reads them as the native-class sentinels ONLY for NVFP4 weights that have an MFMA-native image
        # attached (attach_nvfp4_native: the caller's opt-in), as the library default everywhere else -- the reference's meaning of any negative id
        return solution_id if solution_id >= -4 else -1
    if solution_id >= 1 << 64:
        raise RuntimeError(f"No kernel implementation for solution_id={solution_id}.")
    return solution_id - (1 << 64) if solution_id >= 1 << 63 else solution_id


def repack_nvfp4(b_q_weight, size_n, size_k):
    return torch.ops.petit_kernel.repack_nvfp4(b_q_weight, size_n, size_k)


def process_nvfp4_scales(scales, size_n, size_k):
    return torch.ops.petit_kernel.process_nvfp4_scales(scales, size_n, size_k)


def process_mxfp4_scales(scales, size_n, size_k):
    return torch.ops.petit_kernel.process_mxfp4_scales(scales, size_n, size_k)


def mul_nvfp4_a16(A, B, s, global_scale, size_m, size_n, size_k, solution_id, bias=None, activation=None):
    return torch.ops.petit_kernel.mul_nvfp4_a16(A, B, s, global_scale, size_m, size_n, size_k, _sid(solution_id), bias, _act(activation))


def mul_mxfp4_a16(A, B, s, global_scale, size_m, size_n, size_k, solution_id, bias=None, activation=None):
    return torch.ops.petit_kernel.mul_mxfp4_a16(A, B, s, global_scale, size_m, size_n, size_k, _sid(solution_id), bias, _act(activation))
