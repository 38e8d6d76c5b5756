"""petit_kernel.tuning -- tune-and-persist on the user's device (include/petit_amd.h, petit_gemm_tune).

The reference ships a benchmark binary (`bench_matmul -algo tune`, tools/benchmarks/matmul/main.cc:269-325, driven by
tools/benchmarks/matmul.py) that times every solution for a shape and prints the best id; here the same loop lives inside
the library and its result feeds `solution_id = -1` directly:

    import petit_kernel
    petit_kernel.tune([(8192, 8192), (10240, 8192)], ms=(1, 4, 16, 64), path="petit_tune.txt")   # once, e.g. at install time
    # later processes:  PETIT_AMD_TUNE_FILE=petit_tune.txt  ->  the rows are loaded on the first call

Every candidate's output is checked against the class's reference kernel before it is timed; launches rotate over enough
distinct weight copies that the 256 MB Infinity Cache cannot serve them.  `PETIT_AMD_AUTOTUNE=1` does the same implicitly
on the first `solution_id = -1` call of a shape no table knows; `reserve()` hands the tuner a memory pool so that such a
call allocates nothing (optional).  A tuning run leaves a graph capture in progress on another stream intact.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib

_KLASS = {"exact": 0, "native_mxfp8": 8, "native_mxfp6": 6, "native_mxfp4": 4}

_reserved = {}   # device index -> the tensor the library's tuner draws from (kept alive here)


def reserve(megabytes: int = None, device=None) -> None:
    """Give the library's tuner a memory pool on `device` (petit_tune_reserve): what a tuning run ($PETIT_AMD_AUTOTUNE=1, tune_tensors)
    takes its reference output and its rotation of weight clones from instead of hipMalloc.  Optional; call it at start-up, outside any
    graph capture; default size $PETIT_AMD_AUTOTUNE_RESERVE_MB or 640 MB.  reserve(0) releases the pool."""
    import os
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if megabytes is None:
        megabytes = int(os.environ.get("PETIT_AMD_AUTOTUNE_RESERVE_MB", "640"))
    with torch.cuda.device(dev):
        if megabytes <= 0:
            _lib.lib.petit_tune_reserve(None, 0)
            _reserved.pop(dev.index, None)
            return
        buf = torch.empty(megabytes << 20, dtype=torch.uint8, device=dev)
        rc = _lib.lib.petit_tune_reserve(C.c_void_p(buf.data_ptr()), C.c_uint64(buf.numel()))
        if rc != _lib.PETIT_OK:
            raise RuntimeError(f"petit_tune_reserve: {_lib.error_string(rc)}")
        _reserved[dev.index] = buf


def _hints(dtype, kind: str) -> _lib.SolutionHints:
    a = _lib.CXX_DTYPE_BF16 if dtype == torch.bfloat16 else _lib.CXX_DTYPE_FP16
    b = _lib.CXX_DTYPE_FP4_E2M1 if kind == "nvfp4" else _lib.CXX_DTYPE_MXFP4_E2M1
    return _lib.SolutionHints(a, b, a, 0)


def tune_tensors(A: torch.Tensor, packed, global_scale: torch.Tensor, size_m: int, size_n: int, size_k: int, kind: str = "nvfp4",
                 klass: str = "exact", persist: bool = True, rotate_mb: int = 384, samples: int = 5, workspace_mb: int = 512):
    """Tune one problem on existing tensors.  `packed` is one (B, s) pair from repack_* / process_*_scales, or a list of such
    pairs to rotate over; with a single pair the library clones it on the device up to `rotate_mb`.
    Returns (solution_id, microseconds per launch); with persist=True `solution_id = -1` picks it from now on."""
    if kind not in ("nvfp4", "mxfp4", "mxfp4_f16range") or klass not in _KLASS:
        raise RuntimeError("kind must be 'nvfp4' or 'mxfp4' ('mxfp4_f16range', round 3's name for fp16-safe scales, still reads as 'mxfp4'), klass one of " + ", ".join(_KLASS))
    if A.dtype not in (torch.bfloat16, torch.float16) or not A.is_cuda or not A.is_contiguous() or A.numel() != size_m * size_k:
        raise RuntimeError("A must be a contiguous [size_m, size_k] bfloat16 / float16 GPU tensor")
    pairs = [packed] if isinstance(packed[0], torch.Tensor) else list(packed)
    for b, s in pairs:
        if not (b.is_cuda and s.is_cuda and b.is_contiguous() and s.is_contiguous()) or b.numel() * b.element_size() != size_n * size_k // 2:
            raise RuntimeError("packed tensors must be contiguous GPU tensors from repack_* / process_*_scales")
    dev = A.device
    c = torch.empty((size_m, size_n), dtype=A.dtype, device=dev)
    native_ws = int(_lib.lib.petit_native_workspace_bytes(size_m, size_k))
    ws_bytes = min(workspace_mb << 20, ((native_ws + 255) & ~255) + 8 * size_m * size_n * 4)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    bp = (C.c_void_p * len(pairs))(*[b.data_ptr() for b, _ in pairs])
    sp = (C.c_void_p * len(pairs))(*[s.data_ptr() for _, s in pairs])
    params = _lib.TuneParams(C.sizeof(_lib.TuneParams), _KLASS[klass], len(pairs), 0, bp, sp, (rotate_mb << 20) if len(pairs) == 1 else 0,
                             samples, 0.0, int(persist), 0, 0, 0)
    best, us = C.c_uint64(0), C.c_float(0.0)
    hints = _hints(A.dtype, kind)
    with torch.cuda.device(dev):
        rc = _lib.lib.petit_gemm_tune(c.data_ptr(), A.data_ptr(), global_scale.data_ptr(), size_m, size_n, size_k, C.byref(hints),
                                      C.byref(params), ws.data_ptr(), C.c_uint64(ws_bytes), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream),
                                      C.byref(best), C.byref(us))
    if rc != _lib.PETIT_OK:
        raise RuntimeError(f"petit_gemm_tune(m={size_m}, n={size_n}, k={size_k}): {_lib.error_string(rc)}")
    return int(best.value), float(us.value)


def tune(shapes, ms=(1, 2, 4, 8, 16, 32, 64, 128, 256, 512), kind: str = "nvfp4", dtype=torch.bfloat16, path: str = None,
         klass: str = "exact", rotate_mb: int = 1280, device=None, verbose: bool = False) -> list:
    """Tune every (N, K) of `shapes` at every M of `ms` on synthetic weights (random packed bytes: any bytes are a valid weight
    matrix; scales drawn valid) and record the winners for `solution_id = -1` (klass 'exact') or -2 / -4 / -3 (the native classes: 'native_mxfp8' / 'native_mxfp6' / 'native_mxfp4').
    `path`: also write the rows in the $PETIT_AMD_TUNE_FILE format.  Returns one dict per (shape, M)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    out = []
    for (n, k) in shapes:
        group = 16 if kind == "nvfp4" else 32
        wbytes = n * k // 2 + n * k // group
        copies = int(max(2, min(64, (rotate_mb << 20) // wbytes + 2)))
        gen = torch.Generator(device=dev).manual_seed(1234)
        pairs = []
        for _ in range(copies):
            b = torch.randint(-2 ** 31, 2 ** 31 - 1, (n // 16, 2 * k), generator=gen, dtype=torch.int32, device=dev)
            if kind == "nvfp4":
                s = (torch.rand((n, k // 16), generator=gen, device=dev) * 3.5 + 0.25).to(torch.float8_e4m3fn)
            else:
                s = torch.randint(119, 136, (n // 32, k), generator=gen, dtype=torch.uint8, device=dev)
            pairs.append((b, s))
        gs = torch.ones(1, dtype=torch.float32, device=dev)
        for m in ms:
            a = torch.randn((m, k), generator=gen, device=dev, dtype=torch.float32).to(dtype)
            sid, us = tune_tensors(a, pairs, gs, m, n, k, kind, klass)
            row = {"n": n, "k": k, "m": m, "solution": sid, "us": us, "desc": _lib.describe_solution(sid)}
            out.append(row)
            if verbose:
                print(f"[petit_kernel.tune] {n}x{k} M={m}: {us:.2f} us  0x{sid:x} {row['desc']}", flush=True)
        del pairs
        torch.cuda.empty_cache()
    if path:
        save(path)
    return out


def save(path: str) -> None:
    """Write every run-time row (and the rows loaded from $PETIT_AMD_TUNE_FILE) to `path`."""
    rc = _lib.lib.petit_tune_save(str(path).encode())
    if rc != _lib.PETIT_OK:
        raise RuntimeError(f"petit_tune_save({path}): {_lib.error_string(rc)}")
