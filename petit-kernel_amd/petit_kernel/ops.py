"""petit_kernel.ops -- the torch-facing operator layer.

Mirrors the reference's pybind extension `petit_kernel.ops`
(lib/pybind/pybind.cc:8-26, lib/pybind/fp4.cc): same function names, argument
order and meaning, same output shapes / dtypes, same error classes
(RuntimeError with the reference's message texts).  torch is used only for
device memory and the current stream; the compute is libpetit_amd.so.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import SolutionHints as _CHints

_LAYOUT_N = 16        # kLayoutN, fp4.cc:18
_LAYOUT_M = 128       # kLayoutM, fp4.cc:17
_PACK = 8             # kPackFactor, fp4.cc:19


def _check(cond: bool, msg: str) -> None:
    if not cond:
        raise RuntimeError(msg)  # what TORCH_CHECK / AT_ERROR surface as in Python


def _stream(t: torch.Tensor) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _ptr(t: torch.Tensor) -> C.c_void_p:
    return C.c_void_p(t.data_ptr())


def _raise_on(code: int, what: str) -> None:
    if code != _lib.PETIT_OK:
        raise RuntimeError(f"{what}: {_lib.error_string(code)}")


class PetitSolutionHints:
    """PetitSolutionHints (quantization/gemm.h:112-117; bound at pybind.cc:19-25).

    The reference binds the struct but never registers its enum type, so its
    fields cannot actually be used from Python.
    Here they accept a petit_kernel.DataType, a torch dtype or a raw C++ enum
    integer.
    """

    def __init__(self) -> None:
        self.a_type = None
        self.b_type = None
        self.c_type = None
        self.require_high_precision = False

    def __repr__(self) -> str:
        return (f"PetitSolutionHints(a_type={self.a_type}, b_type={self.b_type}, "
                f"c_type={self.c_type}, require_high_precision={self.require_high_precision})")


def _cxx_dtype(x, default=None) -> int:
    """Anything that names an element type -> the reference's C++ DataType value."""
    if x is None:
        if default is None:
            raise RuntimeError("PetitSolutionHints field is not set")
        return default
    if isinstance(x, torch.dtype):
        if x == torch.bfloat16:
            return _lib.CXX_DTYPE_BF16
        if x == torch.float16:
            return _lib.CXX_DTYPE_FP16
        raise RuntimeError("A must be bfloat16 or float16.")
    name = getattr(x, "name", None)
    if name is not None:  # petit_kernel.DataType (Python numbering, __init__.py:8-15)
        table = {"float16": _lib.CXX_DTYPE_FP16, "bfloat16": _lib.CXX_DTYPE_BF16,
                 "float4_e2m1": _lib.CXX_DTYPE_FP4_E2M1, "mxfloat4_e2m1": _lib.CXX_DTYPE_MXFP4_E2M1,
                 "int4": 0, "float8_e4m3fn": 1, "float8_e5m2fn": 6}
        return table[name]
    return int(x)


def _c_hints(h: PetitSolutionHints) -> _CHints:
    a = _cxx_dtype(h.a_type)
    return _CHints(a, _cxx_dtype(h.b_type, _lib.CXX_DTYPE_FP4_E2M1), _cxx_dtype(h.c_type, a),
                   int(bool(h.require_high_precision)))


def repack_nvfp4(b_q_weight: torch.Tensor, size_n: int, size_k: int) -> torch.Tensor:
    """fp4.cc:38-78 (RepackNvFp4).  int32 [N, K/8] -> int32 [N/16, 2K]."""
    _check(size_k % _LAYOUT_M == 0, f"size_k = {size_k} is not divisible by tile_k_size = {_LAYOUT_M}")
    _check(size_n % _LAYOUT_N == 0, f"size_n = {size_n} is not divisible by tile_n_size = {_LAYOUT_N}")
    _check(b_q_weight.dim() == 2 and size_k // _PACK == b_q_weight.size(1),
           f"Shape mismatch: b_q_weight.size(1) = {b_q_weight.size(-1)}, size_k = {size_k}, pack_factor = {_PACK}")
    _check(b_q_weight.size(0) == size_n, f"b_q_weight.size(0) = {b_q_weight.size(0)} is not size_n = {size_n}")
    _check(b_q_weight.is_cuda, "b_q_weight is not on GPU")
    _check(b_q_weight.is_contiguous(), "b_q_weight is not contiguous")
    _check(b_q_weight.dtype == torch.int32, "b_q_weight type is not kInt")
    out = torch.empty((size_n // _LAYOUT_N, size_k * _LAYOUT_N // _PACK), dtype=torch.int32,
                      device=b_q_weight.device)
    with torch.cuda.device(b_q_weight.device):
        rc = _lib.lib.petit_repack_nvfp4_weights(_ptr(out), _ptr(b_q_weight), size_k, size_n, _stream(out))
    _raise_on(rc, "repack_nvfp4")
    return out


def process_nvfp4_scales(scales: torch.Tensor, size_n: int, size_k: int) -> torch.Tensor:
    """fp4.cc:80-121 (ProcessNvFp4Scales).  e4m3 [N, K/16] -> e4m3 [N, K/16]."""
    group_m = 2 * _LAYOUT_M
    _check(size_k % group_m == 0, f"size_k = {size_k} is not divisible by tile_k_size = {group_m}")
    _check(size_n % _LAYOUT_N == 0, f"size_n = {size_n} is not divisible by tile_n_size = {_LAYOUT_N}")
    _check(scales.dim() == 2 and scales.size(1) > 0 and size_k // scales.size(1) == 16 and
           size_k % scales.size(1) == 0, "Only groupsize = 16 is supported.")
    _check(scales.size(0) == size_n, f"scales.size(0) = {scales.size(0)} is not size_n = {size_n}")
    _check(scales.is_cuda, "scales is not on GPU")
    _check(scales.is_contiguous(), "scales is not contiguous")
    _check(scales.dtype == torch.float8_e4m3fn, "scales type is not float8_e4m3fn")
    out = torch.empty((scales.size(0), scales.size(1)), dtype=scales.dtype, device=scales.device)
    with torch.cuda.device(scales.device):
        rc = _lib.lib.petit_repack_nvfp4_scales(_ptr(out), _ptr(scales), size_k, size_n, _stream(out))
    _raise_on(rc, "process_nvfp4_scales")
    return out


def process_mxfp4_scales(scales: torch.Tensor, size_n: int, size_k: int) -> torch.Tensor:
    """fp4.cc:123-161 (ProcessMxFp4Scales).  uint8 [N, K/32] -> uint8 [N/32, K]."""
    group_m = 2 * _LAYOUT_M
    _check(size_k % group_m == 0, f"size_k = {size_k} is not divisible by tile_k_size = {group_m}")
    _check(size_n % _LAYOUT_N == 0, f"size_n = {size_n} is not divisible by tile_n_size = {_LAYOUT_N}")
    _check(scales.dim() == 2 and scales.size(1) > 0 and size_k // scales.size(1) == 32 and
           size_k % scales.size(1) == 0, "Only groupsize = 32 is supported.")
    _check(scales.size(0) == size_n, f"scales.size(0) = {scales.size(0)} is not size_n = {size_n}")
    _check(scales.is_cuda, "scales is not on GPU")
    _check(scales.is_contiguous(), "scales is not contiguous")
    _check(scales.dtype == torch.uint8, "scales type is not uint8")
    # The reference computes size_n / 32 with integer division (fp4.cc:145-147) and
    # would silently drop the last 16 rows; refuse instead.
    _check(size_n % 32 == 0, f"size_n = {size_n} is not divisible by the MX scale tile (32)")
    out = torch.empty((size_n // 32, size_k), dtype=scales.dtype, device=scales.device)
    with torch.cuda.device(scales.device):
        rc = _lib.lib.petit_repack_mxfp4_scales(_ptr(out), _ptr(scales), size_k, size_n, _stream(out))
    _raise_on(rc, "process_mxfp4_scales")
    return out


_ACTIVATIONS = {None: 0, "none": 0, "silu_mul": 1}   # PETIT_ACTIVATION_* (include/petit_amd.h)

# solution_id of the Python surface: any negative value = "library default" as in the reference (fp4.cc:189-191), with two
# values reserved for the default pick INSIDE the opt-in native-FP4 class (MXFP4 weights only; petit_amd.h)
SOLUTION_AUTO = -1
SOLUTION_AUTO_NATIVE_MXFP8 = -2
SOLUTION_AUTO_NATIVE_MXFP4 = -3
SOLUTION_AUTO_NATIVE_MXFP6 = -4


def _c_solution_id(solution_id: int, native_ok: bool = False) -> int:
    """Python id -> the C ABI's uint64.  Any negative id is the library default, as in the reference (fp4.cc:189,240); the native-class
    sentinels (-2 / -3 / -4) mean themselves only where the caller has opted into that class (mul_mxfp4_native, the resolve / workspace queries)."""
    solution_id = int(solution_id)
    if native_ok and solution_id == SOLUTION_AUTO_NATIVE_MXFP8:
        return _lib.PETIT_SOLUTION_AUTO_NATIVE_MXFP8
    if native_ok and solution_id == SOLUTION_AUTO_NATIVE_MXFP4:
        return _lib.PETIT_SOLUTION_AUTO_NATIVE_MXFP4
    if native_ok and solution_id == SOLUTION_AUTO_NATIVE_MXFP6:
        return _lib.PETIT_SOLUTION_AUTO_NATIVE_MXFP6
    return _lib.PETIT_SOLUTION_AUTO if solution_id < 0 else solution_id

_ws_need_cache = {}


def _workspace_need(a_type: int, b_type: int, m: int, n: int, k: int, sid: int, act: int = 0) -> int:
    """petit_gemm_workspace_bytes_ex, memoised per problem (a pure function of its arguments; of the epilogue only the
    activation matters)."""
    key = (a_type, b_type, m, n, k, sid, act, _lib.lib.petit_tune_generation())   # (a tuned row may change what AUTO needs)
    need = _ws_need_cache.get(key)
    if need is None:
        hints = _CHints(a_type, b_type, a_type, 0)
        epi = _lib.Epilogue(None, act, 0)
        need = int(_lib.lib.petit_gemm_workspace_bytes_ex(C.byref(hints), m, n, k, C.c_uint64(sid), C.byref(epi) if act else None))
        if len(_ws_need_cache) > 4096:
            _ws_need_cache.clear()
        _ws_need_cache[key] = need
    return need


def _mul(kind: str, A, B, s, global_scale, size_m, size_n, size_k, solution_id, bias=None, activation=None) -> torch.Tensor:
    if A.dtype != torch.bfloat16 and A.dtype != torch.float16:
        raise RuntimeError("A must be bfloat16 or float16.")
    # Checks the reference leaves out but whose
    # violation would read out of bounds:
    _check(A.is_cuda and B.is_cuda and s.is_cuda and global_scale.is_cuda, "all tensors must be on GPU")
    _check(A.is_contiguous() and A.numel() == size_m * size_k, "A must be a contiguous [size_m, size_k] tensor")
    _check(B.is_contiguous() and B.numel() * B.element_size() == size_n * size_k // 2,
           "B does not hold size_n * size_k packed 4-bit weights")
    _check(global_scale.dtype == torch.float32 and global_scale.numel() >= 1, "global_scale must be float32")
    _check(activation in _ACTIVATIONS, f"activation must be one of {sorted(k for k in _ACTIVATIONS if k)} or None")
    act = _ACTIVATIONS[activation]
    if act:
        _check(size_n % 32 == 0, f"silu_mul needs size_n % 32 == 0 (gate / up halves of whole tiles), got {size_n}")
    c = torch.empty((size_m, size_n // 2 if act else size_n), dtype=A.dtype, device=A.device)
    a_type = _lib.CXX_DTYPE_BF16 if A.dtype == torch.bfloat16 else _lib.CXX_DTYPE_FP16
    b_type = _lib.CXX_DTYPE_FP4_E2M1 if kind == "nv" else _lib.CXX_DTYPE_MXFP4_E2M1
    # require_high_precision: the reference turns it on for arch <= gfx90a when
    # solution_id < 0 (fp4.cc:24-34,189-191); gfx950 -> False.
    hints = _CHints(a_type, b_type, a_type, 0)
    # (NVFP4 weights with an MFMA-native image attached -- attach_nvfp4_native -- have opted into the native class: -2 / -3 / -4 then name it)
    sid = _c_solution_id(solution_id, native_ok=(kind == "nv" and int(solution_id) in (-2, -3, -4) and B.data_ptr() in _attached_images))
    epi = None
    if bias is not None or act:
        # fused epilogue (include/petit_amd.h, petit_epilogue): c = round16(act(acc * gs + bias[n]))
        if bias is not None:
            _check(bias.is_cuda and bias.device == A.device and bias.dtype == A.dtype and bias.is_contiguous() and
                   bias.numel() == size_n, "bias must be a contiguous [size_n] tensor of A's dtype on A's device")
        epi = _lib.Epilogue(bias.data_ptr() if bias is not None else None, act, 0)
    # Scratch for kernels that need it (cross-workgroup K split, native-FP4 path): ALWAYS per call, from torch's
    # stream-ordered caching allocator -- safe with several streams and under graph capture, and the same rule as the
    # compiled binding (a workspace registered with set_workspace() serves raw C-ABI callers only: it binds to one
    # stream, and a call from a second stream -- e.g. the side stream torch.cuda.graph captures on after an eager
    # warm-up -- would otherwise drop silently to a slower no-scratch kernel).
    ws = None
    ws_bytes = _workspace_need(a_type, b_type, size_m, size_n, size_k, sid, act)
    if ws_bytes:
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=A.device)
    fn = _lib.lib.petit_gemm_fp4_fp16_grid_ws if kind == "nv" else _lib.lib.petit_gemm_mxfp4_fp16_grid_ws
    with torch.cuda.device(A.device):
        err = fn(_ptr(c), _ptr(A), _ptr(B), _ptr(s), _ptr(global_scale), size_m, size_n, size_k,
                 C.byref(hints), C.c_uint64(sid), C.byref(epi) if epi is not None else None,
                 _ptr(ws) if ws is not None else None, C.c_uint64(ws_bytes), _stream(A))
    if err == _lib.PETIT_ERROR_PROBLEM_SHAPE:
        raise RuntimeError(f"Incompatible problem shape (m={size_m}, n={size_n}, k={size_k})")
    if err == _lib.PETIT_ERROR_KERNEL_SHAPE:
        raise RuntimeError(f"No kernel implementation for solution_id={solution_id}.")
    _raise_on(err, "mul_%sfp4_a16" % kind)
    return c


def mul_nvfp4_a16(A, B, s, global_scale, size_m, size_n, size_k, solution_id, bias=None, activation=None) -> torch.Tensor:
    """fp4.cc:163-209 (MulNvFp4A16); `bias` / `activation` (optional, not in the reference) are fused into the epilogue."""
    if s.dim() != 2 or s.size(1) == 0 or size_k // s.size(1) != 16:
        raise RuntimeError(f"Only groupsize = 16 is supported. size_k = {size_k}, s.size(1) = {s.size(-1)}")
    _check(s.numel() == size_n * size_k // 16, "s does not hold size_n * size_k / 16 scales")
    return _mul("nv", A, B, s, global_scale, size_m, size_n, size_k, solution_id, bias, activation)


def mul_mxfp4_a16(A, B, s, global_scale, size_m, size_n, size_k, solution_id, bias=None, activation=None, f16_range=None) -> torch.Tensor:
    """fp4.cc:211-260 (MulMxFp4A16); `bias` / `activation` (optional, not in the reference) are fused into the epilogue.
    `f16_range` (round 3: "every e8m0 block scale lies in 114..140") is accepted and ignored, with a DeprecationWarning: the fp16 x MXFP4 kernels test
    the scale range themselves."""
    if f16_range is not None:
        import warnings
        warnings.warn("mul_mxfp4_a16(f16_range=...) is ignored: the fp16 x MXFP4 kernels test the scale range themselves", DeprecationWarning, stacklevel=2)
    _check(B.size(0) == size_n // _LAYOUT_N, f"B.size(0) = {B.size(0)} is not size_n / 16 = {size_n // _LAYOUT_N}")
    _check(B.size(1) == size_k * _LAYOUT_N // _PACK,
           f"B.size(1) = {B.size(1)} is not packed size = {size_k * _LAYOUT_N // _PACK}")
    _check(s.size(0) == size_n // 32, f"s.size(0) = {s.size(0)} is not size_n / 32 = {size_n // 32}")
    _check(s.size(1) == size_k, f"s.size(1) = {s.size(1)} is not size_k = {size_k}")
    return _mul("mx", A, B, s, global_scale, size_m, size_n, size_k, solution_id, bias, activation)


def get_fp4_solutions(*args) -> list:
    """fp4.cc:262-283 (GetNvFp4Solutions), bound twice (pybind.cc:14-17).

    Accepts both call shapes that exist in the reference: the bound C++ order
    `(hints, size_m, size_n, size_k)` and the Python wrapper's
    `(size_m, size_n, size_k, a_type, c_type)` (petit_kernel/__init__.py:63-66),
    which the reference itself cannot serve.
    """
    if len(args) == 4 and isinstance(args[0], PetitSolutionHints):
        hints, m, n, k = args
        ch = _c_hints(hints)
    elif len(args) == 5:
        m, n, k, a_type, c_type = args
        a = _cxx_dtype(a_type)
        ch = _CHints(a, _lib.CXX_DTYPE_FP4_E2M1, _cxx_dtype(c_type, a), 0)
    else:
        raise TypeError("get_fp4_solutions(hints, m, n, k) or get_fp4_solutions(m, n, k, a_type, c_type)")
    count = C.c_uint(0)
    err = _lib.lib.petit_gemm_get_solutions(C.byref(ch), m, n, k, None, C.byref(count))
    if err != 0:
        raise RuntimeError(f"Failed to get solutions: {err}")
    buf = (C.c_uint64 * max(count.value, 1))()
    err = _lib.lib.petit_gemm_get_solutions(C.byref(ch), m, n, k, buf, C.byref(count))
    if err != 0:
        raise RuntimeError(f"Failed to get solutions: {err}")
    return [int(buf[i]) for i in range(count.value)]


get_nvfp4_solutions = get_fp4_solutions


# --- native-FP4 path (no counterpart in the reference; opt-in, see include/petit_amd.h) -------------

def enable_native_fp4(enable: bool = True) -> None:
    """Let get_fp4_solutions enumerate the native block-scaled-MFMA kernels (MXFP4 weights only;
    activations are quantised to MXFP8 on the fly, a different accuracy class)."""
    _lib.lib.petit_enable_native_fp4(int(bool(enable)))


def set_mxfp4_default_activations(fmt=None) -> None:
    """Process-wide opt-in for call sites that cannot name a sentinel: fmt 'mxfp8' / 'mxfp6' / 'mxfp4' makes solution_id = -1 on MXFP4
    weights (mul_mxfp4_a16 of an unchanged serving stack) run the default pick of that native class for size_m >= $PETIT_AMD_NATIVE_MIN_M
    (64); None switches it off.  The same as $PETIT_AMD_MXFP4_ACTIVATIONS; quantised activations are another accuracy class."""
    _check(fmt is None or fmt in _QFORMATS, "fmt must be None, 'mxfp8', 'mxfp6' or 'mxfp4'")
    _raise_on(_lib.lib.petit_set_mxfp4_default_class(_QFORMATS[fmt] if fmt else 0), "set_mxfp4_default_activations")
    _ws_need_cache.clear()      # (what AUTO needs as scratch has just changed)


def mxfp4_default_activations():
    v = int(_lib.lib.petit_get_mxfp4_default_class())
    return {8: "mxfp8", 6: "mxfp6", 4: "mxfp4"}.get(v)


def native_workspace_bytes(size_m: int, size_k: int) -> int:
    return int(_lib.lib.petit_native_workspace_bytes(size_m, size_k))


def workspace_bytes(hints: PetitSolutionHints, size_m: int, size_n: int, size_k: int, solution_id: int = -1) -> int:
    """Scratch bytes the call would use (split-K slabs, native-FP4 activations); mul_*_a16 allocates them itself."""
    sid = _c_solution_id(solution_id, native_ok=True)
    ch = _c_hints(hints)
    return int(_lib.lib.petit_gemm_workspace_bytes(C.byref(ch), size_m, size_n, size_k, C.c_uint64(sid)))


def resolve_solution(hints: PetitSolutionHints, size_m: int, size_n: int, size_k: int, solution_id: int = -1, activation=None,
                     workspace_bytes: int = 1 << 62) -> int:
    """The concrete kernel id a call with these arguments runs (petit_gemm_resolve_solution): solution_id may be -1, -2 / -3 / -4
    (default pick inside the native class) or an explicit id; 0 when the call would be refused."""
    ch = _c_hints(hints)
    act = _ACTIVATIONS[activation]
    epi = _lib.Epilogue(None, act, 0)
    return int(_lib.lib.petit_gemm_resolve_solution(C.byref(ch), size_m, size_n, size_k, C.c_uint64(_c_solution_id(solution_id, native_ok=True)),
                                                    C.byref(epi) if act else None, C.c_uint64(workspace_bytes)))


def auto_row_split(hints: PetitSolutionHints, size_m: int, size_n: int, size_k: int, activation=None, solution_id: int = -1) -> int:
    """Rows of the first of the TWO launches a default-pick call at this (ragged prefill) M runs as, 0 = one launch (petit_gemm_row_split).  solution_id -2 / -4 / -3:
    the rows a native-class call runs in the class -- the short tail goes through the exact default pick."""
    ch = _c_hints(hints)
    act = _ACTIVATIONS[activation]
    epi = _lib.Epilogue(None, act, 0)
    return int(_lib.lib.petit_gemm_row_split(C.byref(ch), size_m, size_n, size_k, C.c_uint64(_c_solution_id(solution_id, native_ok=True)), C.byref(epi) if act else None))


def dequant_packed(B: torch.Tensor, s: torch.Tensor, size_n: int, size_k: int, kind: str = "nvfp4", dtype=torch.float32,
                   global_scale: float = 1.0) -> torch.Tensor:
    """Dense [size_n, size_k] expansion of PACKED weights (from repack_nvfp4 / process_*_scales): a debug aid, the
    counterpart of the reference's test-only DequantPetitFp4 kernels (quantization_utils.cu:542-727)."""
    _check(kind in ("nvfp4", "mxfp4"), "kind must be 'nvfp4' or 'mxfp4'")
    _check(B.is_cuda and s.is_cuda and B.is_contiguous() and s.is_contiguous(), "packed tensors must be contiguous GPU tensors")
    _check(B.numel() * B.element_size() == size_n * size_k // 2, "B does not hold size_n * size_k packed 4-bit weights")
    out_type = {torch.float32: _lib.PETIT_DTYPE_FP32, torch.bfloat16: _lib.CXX_DTYPE_BF16, torch.float16: _lib.CXX_DTYPE_FP16}.get(dtype)
    _check(out_type is not None, "dtype must be float32, bfloat16 or float16")
    out = torch.empty((size_n, size_k), dtype=dtype, device=B.device)
    with torch.cuda.device(B.device):
        rc = _lib.lib.petit_dequant_packed_weights(_ptr(out), _ptr(B), _ptr(s), float(global_scale), size_n, size_k,
                                                   _lib.CXX_DTYPE_FP4_E2M1 if kind == "nvfp4" else _lib.CXX_DTYPE_MXFP4_E2M1,
                                                   out_type, _stream(B))
    _raise_on(rc, "dequant_packed")
    return out


_workspace_keepalive = {}


def set_workspace(buf) -> None:
    """Register (or with None, unregister) a device scratch tensor for kernels that need one
    (split-K slabs, quantised activations of the native path).  The tensor is kept alive here.
    Only raw C-ABI callers of the entry points WITHOUT a workspace argument use it (petit_gemm_*_grid, _ex):
    mul_*_a16 of both Python layers always allocates its scratch per call.  A registered workspace serves
    ONE stream (include/petit_amd.h "Scratch memory")."""
    if buf is None:
        with torch.cuda.device(torch.cuda.current_device()):
            _lib.lib.petit_set_workspace(None, 0)
        _workspace_keepalive.pop(torch.cuda.current_device(), None)
        return
    _check(buf.is_cuda and buf.is_contiguous(), "workspace must be a contiguous GPU tensor")
    with torch.cuda.device(buf.device):
        _lib.lib.petit_set_workspace(_ptr(buf), C.c_uint64(buf.numel() * buf.element_size()))
        _workspace_keepalive[buf.device.index] = buf


# --- the native class as a pipeline (include/petit_amd.h "The native class as a PIPELINE"; no counterpart in the reference) ---

_QFORMATS = {"mxfp8": 8, "mxfp6": 6, "mxfp4": 4}


class QuantizedActivations:
    """Activations [m, k] quantised to MXFP8 / MXFP4 in the layout the 32x32x64 native kernels read ("petit-qact/1": opaque
    bytes, k-tile major).  Produced by quantize_activations() or by mul_mxfp4_a16(..., activation="silu_mul",
    out_quantized=...); consumed by mul_mxfp4_a16(a=<this>, ...)."""

    def __init__(self, data: torch.Tensor, m: int, k: int, fmt: str, dtype: torch.dtype):
        self.data, self.m, self.k, self.fmt, self.dtype = data, m, k, fmt, dtype

    def __repr__(self) -> str:
        return f"QuantizedActivations(m={self.m}, k={self.k}, fmt={self.fmt!r}, dtype={self.dtype}, {self.data.numel()} bytes)"


def quantize_activations(A: torch.Tensor, fmt: str = "mxfp4") -> QuantizedActivations:
    """16-bit activations [m, k] -> QuantizedActivations (one launch; share the result among GEMMs with the same input)."""
    _check(fmt in _QFORMATS, "fmt must be 'mxfp8', 'mxfp6' or 'mxfp4'")
    _check(A.is_cuda and A.is_contiguous() and A.dim() == 2 and A.dtype in (torch.bfloat16, torch.float16),
           "A must be a contiguous 2-D bfloat16 / float16 GPU tensor")
    m, k = A.shape
    nbytes = int(_lib.lib.petit_quantized_activation_bytes(m, k, _QFORMATS[fmt]))
    qa = torch.empty(nbytes, dtype=torch.uint8, device=A.device)
    with torch.cuda.device(A.device):
        rc = _lib.lib.petit_quantize_activations(_ptr(qa), _ptr(A), m, k, _lib.CXX_DTYPE_BF16 if A.dtype == torch.bfloat16 else _lib.CXX_DTYPE_FP16,
                                                 _QFORMATS[fmt], _stream(A))
    if rc == _lib.PETIT_ERROR_PROBLEM_SHAPE:
        raise RuntimeError(f"Incompatible problem shape (m={m}, k={k})")
    _raise_on(rc, "quantize_activations")
    return QuantizedActivations(qa, m, k, fmt, A.dtype)


def mul_mxfp4_native(A, B, s, global_scale, size_m, size_n, size_k, solution_id=SOLUTION_AUTO_NATIVE_MXFP4, bias=None, activation=None,
                     out_quantized=None):
    """The native-FP4 class with its hand-over points (petit_gemm_mxfp4_native).  A: a 16-bit [size_m, size_k] tensor (quantised
    by the call: two launches) or QuantizedActivations (one launch).  out_quantized 'mxfp8' / 'mxfp6' / 'mxfp4' (with activation='silu_mul'):
    returns QuantizedActivations [size_m, size_n / 2] for the next GEMM instead of a 16-bit tensor.
    solution_id: SOLUTION_AUTO_NATIVE_MXFP8 / _MXFP4 / _MXFP6 (-2 / -3 / -4) or an explicit native kernel id."""
    pre = isinstance(A, QuantizedActivations)
    if pre:
        _check(A.m == size_m and A.k == size_k, f"quantised activations are [{A.m}, {A.k}], the call says [{size_m}, {size_k}]")
        a_t, dtype, a_fmt, dev = A.data, A.dtype, _QFORMATS[A.fmt], A.data.device
    else:
        _check(A.is_cuda and A.is_contiguous() and A.numel() == size_m * size_k and A.dtype in (torch.bfloat16, torch.float16),
               "A must be a contiguous [size_m, size_k] bfloat16 / float16 GPU tensor")
        a_t, dtype, a_fmt, dev = A, A.dtype, 0, A.device
    _check(B.is_cuda and s.is_cuda and global_scale.is_cuda, "all tensors must be on GPU")
    _check(B.is_contiguous() and B.numel() * B.element_size() == size_n * size_k // 2, "B does not hold size_n * size_k packed 4-bit weights")
    _check(s.is_contiguous() and s.numel() * s.element_size() == size_n * size_k // 32, "s does not hold size_n * size_k / 32 scales")
    _check(activation in _ACTIVATIONS, f"activation must be one of {sorted(k for k in _ACTIVATIONS if k)} or None")
    _check(out_quantized is None or out_quantized in _QFORMATS, "out_quantized must be None, 'mxfp8', 'mxfp6' or 'mxfp4'")
    act = _ACTIVATIONS[activation]
    out_fmt = _QFORMATS[out_quantized] if out_quantized else 0
    _check(not out_fmt or act, "out_quantized needs activation='silu_mul'")
    a_type = _lib.CXX_DTYPE_BF16 if dtype == torch.bfloat16 else _lib.CXX_DTYPE_FP16
    hints = _CHints(a_type, _lib.CXX_DTYPE_MXFP4_E2M1, a_type, 0)
    sid = _c_solution_id(solution_id, native_ok=True)
    epi = None
    if bias is not None or act:
        if bias is not None:
            _check(bias.is_cuda and bias.device == dev and bias.dtype == dtype and bias.is_contiguous() and bias.numel() == size_n,
                   "bias must be a contiguous [size_n] tensor of the activation dtype on the same device")
        epi = _lib.Epilogue(bias.data_ptr() if bias is not None else None, act, 0)
    na = _lib.NativeArgs(C.sizeof(_lib.NativeArgs), a_fmt, out_fmt, 0)
    epi_p = C.byref(epi) if epi is not None else None
    if out_fmt:
        c = torch.empty(int(_lib.lib.petit_quantized_activation_bytes(size_m, size_n // 2, out_fmt)), dtype=torch.uint8, device=dev)
    else:
        c = torch.empty((size_m, size_n // 2 if act else size_n), dtype=dtype, device=dev)
    ws_bytes = int(_lib.lib.petit_gemm_native_workspace_bytes(C.byref(hints), size_m, size_n, size_k, C.c_uint64(sid), epi_p, C.byref(na)))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev) if ws_bytes else None
    with torch.cuda.device(dev):
        err = _lib.lib.petit_gemm_mxfp4_native(_ptr(c), _ptr(a_t), _ptr(B), _ptr(s), _ptr(global_scale), size_m, size_n, size_k, C.byref(hints),
                                               C.c_uint64(sid), epi_p, C.byref(na), _ptr(ws) if ws is not None else None, C.c_uint64(ws_bytes),
                                               C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if err == _lib.PETIT_ERROR_PROBLEM_SHAPE:
        raise RuntimeError(f"Incompatible problem shape (m={size_m}, n={size_n}, k={size_k})")
    if err == _lib.PETIT_ERROR_KERNEL_SHAPE:
        raise RuntimeError(f"No kernel implementation for solution_id={solution_id}.")
    _raise_on(err, "mul_mxfp4_native")
    return QuantizedActivations(c, size_m, size_n // 2, out_quantized, dtype) if out_fmt else c


# --- NVFP4 weights on the native class (include/petit_amd.h "NVFP4 weights on the native class"; no counterpart in the reference) ----------

_attached_images = {}   # data_ptr of the packed weights -> the image tensor (kept alive for as long as it is attached)


def nvfp4_native_image(B: torch.Tensor, s: torch.Tensor, size_n: int, size_k: int) -> torch.Tensor:
    """The MFMA-native image ("petit-cdna4-nv6/1": FP6 e2m3 elements + one E8M0 scale per 32 k, 6.25 bits per weight) of NVFP4 weights,
    from the PACKED tensors of repack_nvfp4 / process_nvfp4_scales; one launch, once at load time.  An opaque uint8 tensor."""
    _check(B.is_cuda and s.is_cuda and B.device == s.device, "B and s must be GPU tensors on one device")
    _check(B.is_contiguous() and B.numel() * B.element_size() == size_n * size_k // 2, "B does not hold size_n * size_k packed 4-bit weights")
    _check(s.is_contiguous() and s.numel() * s.element_size() == size_n * size_k // 16, "s does not hold size_n * size_k / 16 scales")
    nbytes = int(_lib.lib.petit_nvfp4_native_image_bytes(size_k, size_n))
    if nbytes == 0:
        raise RuntimeError(f"Incompatible problem shape (n={size_n}, k={size_k})")
    image = torch.empty(nbytes, dtype=torch.uint8, device=B.device)
    with torch.cuda.device(B.device):
        rc = _lib.lib.petit_nvfp4_native_image(_ptr(image), _ptr(B), _ptr(s), size_k, size_n, _stream(B))
    if rc == _lib.PETIT_ERROR_PROBLEM_SHAPE:
        raise RuntimeError(f"Incompatible problem shape (n={size_n}, k={size_k})")
    _raise_on(rc, "nvfp4_native_image")
    return image


def attach_nvfp4_native(B: torch.Tensor, image) -> None:
    """Attach `image` (nvfp4_native_image) to the packed weights B: mul_nvfp4_a16(..., B, ..., solution_id = -2 / -3 / -4) then runs the native
    class on it (MXFP8 / MXFP4 / MXFP6 activations).  image = None detaches.  The image is kept alive while attached; detach before B is freed."""
    key = B.data_ptr()
    if image is None:
        _attached_images.pop(key, None)
        _raise_on(_lib.lib.petit_nvfp4_native_attach(C.c_void_p(key), None), "attach_nvfp4_native")
        return
    _check(image.is_cuda and image.device == B.device and image.dtype == torch.uint8 and image.is_contiguous(), "image must come from nvfp4_native_image")
    _raise_on(_lib.lib.petit_nvfp4_native_attach(C.c_void_p(key), _ptr(image)), "attach_nvfp4_native")
    _attached_images[key] = image


def mul_nvfp4_native(A, image: torch.Tensor, global_scale, size_m, size_n, size_k, solution_id=SOLUTION_AUTO_NATIVE_MXFP8, bias=None,
                     activation=None, out_quantized=None):
    """NVFP4 weights on the block-scaled MFMA (petit_gemm_nvfp4_native): `image` from nvfp4_native_image; everything else as mul_mxfp4_native
    (A a 16-bit tensor or QuantizedActivations; solution_id -2 / -3 / -4 = MXFP8 / MXFP4 / MXFP6 activations or an explicit native id of the
    NVFP4 family; out_quantized with activation='silu_mul')."""
    pre = isinstance(A, QuantizedActivations)
    if pre:
        _check(A.m == size_m and A.k == size_k, f"quantised activations are [{A.m}, {A.k}], the call says [{size_m}, {size_k}]")
        a_t, dtype, a_fmt, dev = A.data, A.dtype, _QFORMATS[A.fmt], A.data.device
    else:
        _check(A.is_cuda and A.is_contiguous() and A.numel() == size_m * size_k and A.dtype in (torch.bfloat16, torch.float16),
               "A must be a contiguous [size_m, size_k] bfloat16 / float16 GPU tensor")
        a_t, dtype, a_fmt, dev = A, A.dtype, 0, A.device
    _check(image.is_cuda and global_scale.is_cuda, "all tensors must be on GPU")
    _check(image.is_contiguous() and image.dtype == torch.uint8 and
           image.numel() == int(_lib.lib.petit_nvfp4_native_image_bytes(size_k, size_n)) and image.numel() > 0,
           "image does not hold the native image of size_n x size_k NVFP4 weights")
    _check(activation in _ACTIVATIONS, f"activation must be one of {sorted(k for k in _ACTIVATIONS if k)} or None")
    _check(out_quantized is None or out_quantized in _QFORMATS, "out_quantized must be None, 'mxfp8', 'mxfp6' or 'mxfp4'")
    act = _ACTIVATIONS[activation]
    out_fmt = _QFORMATS[out_quantized] if out_quantized else 0
    _check(not out_fmt or act, "out_quantized needs activation='silu_mul'")
    a_type = _lib.CXX_DTYPE_BF16 if dtype == torch.bfloat16 else _lib.CXX_DTYPE_FP16
    hints = _CHints(a_type, _lib.CXX_DTYPE_FP4_E2M1, a_type, 0)
    sid = _c_solution_id(solution_id, native_ok=True)
    epi = None
    if bias is not None or act:
        if bias is not None:
            _check(bias.is_cuda and bias.device == dev and bias.dtype == dtype and bias.is_contiguous() and bias.numel() == size_n,
                   "bias must be a contiguous [size_n] tensor of the activation dtype on the same device")
        epi = _lib.Epilogue(bias.data_ptr() if bias is not None else None, act, 0)
    na = _lib.NativeArgs(C.sizeof(_lib.NativeArgs), a_fmt, out_fmt, 0)
    epi_p = C.byref(epi) if epi is not None else None
    if out_fmt:
        c = torch.empty(int(_lib.lib.petit_quantized_activation_bytes(size_m, size_n // 2, out_fmt)), dtype=torch.uint8, device=dev)
    else:
        c = torch.empty((size_m, size_n // 2 if act else size_n), dtype=dtype, device=dev)
    ws_bytes = int(_lib.lib.petit_gemm_native_workspace_bytes(C.byref(hints), size_m, size_n, size_k, C.c_uint64(sid), epi_p, C.byref(na)))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev) if ws_bytes else None
    with torch.cuda.device(dev):
        err = _lib.lib.petit_gemm_nvfp4_native(_ptr(c), _ptr(a_t), _ptr(image), _ptr(global_scale), size_m, size_n, size_k, C.byref(hints),
                                               C.c_uint64(sid), epi_p, C.byref(na), _ptr(ws) if ws is not None else None, C.c_uint64(ws_bytes),
                                               C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if err == _lib.PETIT_ERROR_PROBLEM_SHAPE:
        raise RuntimeError(f"Incompatible problem shape (m={size_m}, n={size_n}, k={size_k})")
    if err == _lib.PETIT_ERROR_KERNEL_SHAPE:
        raise RuntimeError(f"No kernel implementation for solution_id={solution_id}.")
    _raise_on(err, "mul_nvfp4_native")
    return QuantizedActivations(c, size_m, size_n // 2, out_quantized, dtype) if out_fmt else c


# --- grouped launch (include/petit_amd.h "Grouped launch"; no counterpart in the reference) ---------------------------------

def mul_fp4_a16_grouped(kind: str, A: torch.Tensor, members, size_m: int, size_k: int, solution_id: int = -1) -> list:
    """Up to 8 GEMMs that share the activations A [size_m, size_k] in ONE launch (size_m <= 16): `members` is a list of
    (B, s, global_scale, size_n) or (B, s, global_scale, size_n, bias) with tensors from repack_* / process_*_scales of `kind`
    ('nvfp4' / 'mxfp4').  Returns the list of outputs [size_m, size_n_i]; bit-identical to separate calls with the same kernel."""
    _check(kind in ("nvfp4", "mxfp4"), "kind must be 'nvfp4' or 'mxfp4'")
    _check(A.dtype in (torch.bfloat16, torch.float16), "A must be bfloat16 or float16.")
    _check(A.is_cuda and A.is_contiguous() and A.numel() == size_m * size_k, "A must be a contiguous [size_m, size_k] tensor")
    _check(1 <= len(members) <= 8, "a group holds 1 to 8 members")
    group = 16 if kind == "nvfp4" else 32
    arr = (_lib.GroupMember * len(members))()
    outs = []
    for i, mem in enumerate(members):
        B, s, gs, n = mem[:4]
        bias = mem[4] if len(mem) > 4 else None
        _check(B.is_cuda and s.is_cuda and gs.is_cuda and B.device == A.device, "all tensors must be on A's GPU")
        _check(B.is_contiguous() and B.numel() * B.element_size() == n * size_k // 2, "B does not hold size_n * size_k packed 4-bit weights")
        _check(s.is_contiguous() and s.numel() * s.element_size() == n * size_k // group, f"s does not hold size_n * size_k / {group} scales")
        _check(gs.dtype == torch.float32 and gs.numel() >= 1, "global_scale must be float32")
        if bias is not None:
            _check(bias.is_cuda and bias.dtype == A.dtype and bias.is_contiguous() and bias.numel() == n, "bias must be a contiguous [size_n] tensor of A's dtype")
        c = torch.empty((size_m, n), dtype=A.dtype, device=A.device)
        outs.append(c)
        arr[i] = _lib.GroupMember(c.data_ptr(), B.data_ptr(), s.data_ptr(), gs.data_ptr(), bias.data_ptr() if bias is not None else None, n, 0)
    a_type = _lib.CXX_DTYPE_BF16 if A.dtype == torch.bfloat16 else _lib.CXX_DTYPE_FP16
    b_type = _lib.CXX_DTYPE_FP4_E2M1 if kind == "nvfp4" else _lib.CXX_DTYPE_MXFP4_E2M1
    hints = _CHints(a_type, b_type, a_type, 0)
    with torch.cuda.device(A.device):
        err = _lib.lib.petit_gemm_fp4_fp16_grouped(arr, len(members), _ptr(A), size_m, size_k, C.byref(hints), C.c_uint64(_c_solution_id(solution_id)),
                                                   _stream(A))
    if err == _lib.PETIT_ERROR_PROBLEM_SHAPE:
        raise RuntimeError(f"Incompatible problem shape (m={size_m}, k={size_k}, n={[mem[3] for mem in members]})")
    if err == _lib.PETIT_ERROR_KERNEL_SHAPE:
        raise RuntimeError(f"No kernel implementation for solution_id={solution_id}.")
    _raise_on(err, "mul_fp4_a16_grouped")
    return outs
