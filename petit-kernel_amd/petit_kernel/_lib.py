"""ctypes loader for libpetit_amd.so (the C ABI of include/petit_amd.h).

The product path has no CPU fallback: if the HIP library is missing this
module raises at import time, loudly.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_PKG_ROOT = Path(__file__).resolve().parent.parent  # petit-kernel_amd/
LIB_PATH = Path(os.environ.get("PETIT_AMD_LIB", _PKG_ROOT / "lib" / "libpetit_amd.so"))

PETIT_OK = 0
PETIT_ERROR_PROBLEM_SHAPE = 1
PETIT_ERROR_KERNEL_SHAPE = 2
PETIT_ERROR_LAUNCH = 3
PETIT_ERROR_BAD_ARGUMENT = 4
PETIT_SOLUTION_AUTO = 0xFFFFFFFFFFFFFFFF
PETIT_SOLUTION_AUTO_NATIVE_MXFP8 = PETIT_SOLUTION_AUTO - 1   # Python surface: solution_id = -2
PETIT_SOLUTION_AUTO_NATIVE_MXFP4 = PETIT_SOLUTION_AUTO - 2   # Python surface: solution_id = -3
PETIT_SOLUTION_AUTO_NATIVE_MXFP6 = PETIT_SOLUTION_AUTO - 3   # Python surface: solution_id = -4

# C++ DataType numbering of the reference (quantization/types.h:4-13)
CXX_DTYPE_FP4_E2M1 = 3
CXX_DTYPE_FP16 = 4
CXX_DTYPE_BF16 = 5
CXX_DTYPE_MXFP4_E2M1 = 7
CXX_DTYPE_MXFP4_E2M1_F16RANGE = 8   # extension: MXFP4 with every e8m0 scale in 114..140 (include/petit_amd.h)
MXFP4_F16RANGE_SCALE_MIN, MXFP4_F16RANGE_SCALE_MAX = 114, 140
PETIT_DTYPE_FP32 = 100   # petit_dequant_packed_weights only


class SolutionHints(C.Structure):
    """petit_solution_hints (include/petit_amd.h)."""
    _fields_ = [("a_type", C.c_int32), ("b_type", C.c_int32), ("c_type", C.c_int32),
                ("require_high_precision", C.c_int32)]


class Epilogue(C.Structure):
    """petit_epilogue (include/petit_amd.h)."""
    _fields_ = [("bias", C.c_void_p), ("activation", C.c_int32), ("reserved", C.c_int32)]


class GroupMember(C.Structure):
    """petit_group_member (include/petit_amd.h)."""
    _fields_ = [("c", C.c_void_p), ("b", C.c_void_p), ("scales", C.c_void_p), ("global_scale", C.c_void_p), ("bias", C.c_void_p),
                ("n", C.c_uint32), ("reserved", C.c_uint32)]


class NativeArgs(C.Structure):
    """petit_native_args (include/petit_amd.h)."""
    _fields_ = [("struct_bytes", C.c_uint32), ("a_format", C.c_int32), ("out_format", C.c_int32), ("reserved", C.c_int32)]


class TuneParams(C.Structure):
    """petit_tune_params (include/petit_amd.h)."""
    _fields_ = [("struct_bytes", C.c_uint32), ("klass", C.c_int32), ("n_copies", C.c_uint32), ("launches", C.c_uint32),
                ("b", C.POINTER(C.c_void_p)), ("scales", C.POINTER(C.c_void_p)), ("rotate_bytes", C.c_uint64),
                ("samples", C.c_uint32), ("tolerance", C.c_float), ("persist", C.c_int32), ("m_lo", C.c_uint32), ("m_hi", C.c_uint32),
                ("reserved", C.c_uint32)]


# every symbol include/petit_amd.h declares, with its signature
_SIGNATURES = {
    "petit_gemm_fp4_fp16_grid_ex": (C.c_int, [C.c_void_p] * 5 + [C.c_uint] * 3 +
                                    [C.POINTER(SolutionHints), C.c_uint64, C.POINTER(Epilogue), C.c_void_p]),
    "petit_gemm_mxfp4_fp16_grid_ex": (C.c_int, [C.c_void_p] * 5 + [C.c_uint] * 3 +
                                      [C.POINTER(SolutionHints), C.c_uint64, C.POINTER(Epilogue), C.c_void_p]),
    "petit_gemm_fp4_fp16_grid_ws": (C.c_int, [C.c_void_p] * 5 + [C.c_uint] * 3 +
                                    [C.POINTER(SolutionHints), C.c_uint64, C.POINTER(Epilogue), C.c_void_p, C.c_uint64, C.c_void_p]),
    "petit_gemm_mxfp4_fp16_grid_ws": (C.c_int, [C.c_void_p] * 5 + [C.c_uint] * 3 +
                                      [C.POINTER(SolutionHints), C.c_uint64, C.POINTER(Epilogue), C.c_void_p, C.c_uint64, C.c_void_p]),
    "petit_gemm_workspace_bytes": (C.c_uint64, [C.POINTER(SolutionHints), C.c_uint, C.c_uint, C.c_uint, C.c_uint64]),
    "petit_gemm_workspace_bytes_ex": (C.c_uint64, [C.POINTER(SolutionHints), C.c_uint, C.c_uint, C.c_uint, C.c_uint64, C.POINTER(Epilogue)]),
    "petit_gemm_resolve_solution": (C.c_uint64, [C.POINTER(SolutionHints), C.c_uint, C.c_uint, C.c_uint, C.c_uint64, C.POINTER(Epilogue), C.c_uint64]),
    "petit_gemm_fp4_fp16_grid": (C.c_int, [C.c_void_p] * 5 + [C.c_uint] * 3 +
                                 [C.POINTER(SolutionHints), C.c_uint64, C.c_void_p]),
    "petit_gemm_mxfp4_fp16_grid": (C.c_int, [C.c_void_p] * 5 + [C.c_uint] * 3 +
                                   [C.POINTER(SolutionHints), C.c_uint64, C.c_void_p]),
    "petit_gemm_get_solutions": (C.c_int, [C.POINTER(SolutionHints), C.c_uint, C.c_uint, C.c_uint,
                                           C.POINTER(C.c_uint64), C.POINTER(C.c_uint)]),
    "petit_gemm_default_solution": (C.c_uint64, [C.POINTER(SolutionHints), C.c_uint, C.c_uint, C.c_uint]),
    "petit_gemm_auto_row_split": (C.c_uint, [C.POINTER(SolutionHints), C.c_uint, C.c_uint, C.c_uint, C.c_void_p]),
    "petit_gemm_row_split": (C.c_uint, [C.POINTER(SolutionHints), C.c_uint, C.c_uint, C.c_uint, C.c_uint64, C.POINTER(Epilogue)]),
    "petit_raster_tile": (None, [C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.POINTER(C.c_uint), C.POINTER(C.c_uint)]),
    "petit_repack_nvfp4_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_void_p]),
    "petit_repack_nvfp4_scales": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_void_p]),
    "petit_repack_mxfp4_scales": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_void_p]),
    "petit_repack_nvfp4_weights_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint]),
    "petit_repack_nvfp4_scales_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint]),
    "petit_repack_mxfp4_scales_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint]),
    "petit_convert_reference_weights_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint]),
    "petit_convert_reference_nvfp4_scales_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint]),
    "petit_convert_reference_mxfp4_scales_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint]),
    "petit_dequant_packed_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_uint, C.c_uint, C.c_int, C.c_int, C.c_void_p]),
    "petit_set_workspace": (C.c_int, [C.c_void_p, C.c_uint64]),
    "petit_workspace_bytes": (C.c_uint64, [C.c_uint64, C.c_uint, C.c_uint]),
    "petit_enable_native_fp4": (C.c_int, [C.c_int]),
    "petit_set_mxfp4_default_class": (C.c_int, [C.c_int]),
    "petit_get_mxfp4_default_class": (C.c_int, []),
    "petit_native_workspace_bytes": (C.c_uint64, [C.c_uint, C.c_uint]),
    "petit_gemm_fp4_fp16_grouped": (C.c_int, [C.POINTER(GroupMember), C.c_uint, C.c_void_p, C.c_uint, C.c_uint, C.POINTER(SolutionHints), C.c_uint64,
                                              C.c_void_p]),
    "petit_gemm_mxfp4_native": (C.c_int, [C.c_void_p] * 5 + [C.c_uint] * 3 + [C.POINTER(SolutionHints), C.c_uint64, C.POINTER(Epilogue),
                                          C.POINTER(NativeArgs), C.c_void_p, C.c_uint64, C.c_void_p]),
    "petit_nvfp4_native_image_bytes": (C.c_uint64, [C.c_uint, C.c_uint]),
    "petit_nvfp4_native_image": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_void_p]),
    "petit_nvfp4_native_image_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_uint]),
    "petit_nvfp4_native_image_dequant_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint]),
    "petit_nvfp4_native_attach": (C.c_int, [C.c_void_p, C.c_void_p]),
    "petit_nvfp4_native_attached": (C.c_void_p, [C.c_void_p]),
    "petit_gemm_nvfp4_native": (C.c_int, [C.c_void_p] * 4 + [C.c_uint] * 3 + [C.POINTER(SolutionHints), C.c_uint64, C.POINTER(Epilogue),
                                          C.POINTER(NativeArgs), C.c_void_p, C.c_uint64, C.c_void_p]),
    "petit_gemm_native_workspace_bytes": (C.c_uint64, [C.POINTER(SolutionHints), C.c_uint, C.c_uint, C.c_uint, C.c_uint64, C.POINTER(Epilogue),
                                                       C.POINTER(NativeArgs)]),
    "petit_quantized_activation_bytes": (C.c_uint64, [C.c_uint, C.c_uint, C.c_int]),
    "petit_quantize_activations": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_int, C.c_int, C.c_void_p]),
    "petit_gemm_tune": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_uint, C.POINTER(SolutionHints),
                                  C.POINTER(TuneParams), C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_float)]),
    "petit_tune_reserve": (C.c_int, [C.c_void_p, C.c_uint64]),
    "petit_tune_insert": (C.c_int, [C.POINTER(SolutionHints), C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_uint64]),
    "petit_tune_save": (C.c_int, [C.c_char_p]),
    "petit_tune_generation": (C.c_uint64, []),
    "petit_error_string": (C.c_char_p, [C.c_int]),
    "petit_layout_tag": (C.c_char_p, []),
    "petit_version": (C.c_char_p, []),
    "petit_describe_solution": (C.c_int, [C.c_uint64, C.c_char_p, C.c_uint]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def load() -> C.CDLL:
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python petit-kernel_amd/build.py` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    return lib


lib = load()


def error_string(code: int) -> str:
    return lib.petit_error_string(code).decode()


def describe_solution(solution_id: int) -> str:
    buf = C.create_string_buffer(256)
    lib.petit_describe_solution(C.c_uint64(solution_id & PETIT_SOLUTION_AUTO), buf, 256)
    return buf.value.decode()
