"""ctypes/numpy front-end of the CPU oracle (oracle/petit_oracle.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never by the product package.  See the header of
petit_oracle.c for what each function restates (reference file:line).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "_build" / "libpetit_oracle.so"
_REF_PATH = _HERE / "_ref" / "libref_floats.so"


def build(force: bool = False) -> None:
    """Compile the oracle (and oracle/_ref when /root/reference exists)."""
    src = _HERE / "petit_oracle.c"
    stale = (not _LIB_PATH.exists()) or _LIB_PATH.stat().st_mtime < src.stat().st_mtime
    if force or stale:
        subprocess.run(["make", "-C", str(_HERE), "oracle"], check=True,
                       stdout=subprocess.DEVNULL)
    if Path("/root/reference/lib/tests/floating_points.h").exists():
        shim = _HERE / "ref_shim.cc"
        if force or not _REF_PATH.exists() or _REF_PATH.stat().st_mtime < shim.stat().st_mtime:
            subprocess.run(["make", "-C", str(_HERE), "ref"], check=True,
                           stdout=subprocess.DEVNULL)


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(str(_LIB_PATH))
        u8p, u16p, u32p, f32p = (C.POINTER(t) for t in (C.c_uint8, C.c_uint16, C.c_uint32, C.c_float))
        L.po_fp4_to_f32.restype = C.c_float
        L.po_fp4_to_f32.argtypes = [C.c_uint]
        for name in ("po_e4m3_to_f32", "po_e5m2_to_f32", "po_e8m0_to_f32", "po_e5m3_to_f32"):
            getattr(L, name).restype = C.c_float
            getattr(L, name).argtypes = [C.c_uint8]
        L.po_e4m3_to_e5m3.restype = C.c_uint8
        L.po_e4m3_to_e5m3.argtypes = [C.c_uint8]
        for name in ("po_f32_to_bf16", "po_f32_to_f16"):
            getattr(L, name).restype = C.c_uint16
            getattr(L, name).argtypes = [C.c_float]
        for name in ("po_bf16_to_f32", "po_f16_to_f32"):
            getattr(L, name).restype = C.c_float
            getattr(L, name).argtypes = [C.c_uint16]
        L.po_dequant_nvfp4.restype = None
        L.po_dequant_nvfp4.argtypes = [u8p, u8p, C.c_int, C.c_int, f32p]
        L.po_dequant_mxfp4.restype = None
        L.po_dequant_mxfp4.argtypes = [u8p, u8p, C.c_int, C.c_int, f32p]
        L.po_gemm_ref.restype = None
        L.po_gemm_ref.argtypes = [u16p, C.c_int, f32p, C.c_float, C.c_int, C.c_int, C.c_int, u16p, f32p]
        L.po_fp4_gemm_cpu.restype = C.c_int
        L.po_fp4_gemm_cpu.argtypes = [u16p, C.c_int, u8p, u8p, C.c_float, C.c_int,
                                      C.c_int, C.c_int, C.c_int, u16p]
        L.po_num_threads.restype = C.c_int
        L.po_petit_format.restype = C.c_uint32
        L.po_petit_format.argtypes = [C.c_uint32]
        L.po_petit_word_decode.restype = None
        L.po_petit_word_decode.argtypes = [C.c_uint32, f32p]
        L.po_petit_repack_weights.restype = C.c_int
        L.po_petit_repack_weights.argtypes = [u32p, C.c_int, C.c_int, u32p]
        L.po_petit_repack_nvscales.restype = C.c_int
        L.po_petit_repack_nvscales.argtypes = [u8p, C.c_int, C.c_int, u8p]
        L.po_petit_repack_mxscales.restype = C.c_int
        L.po_petit_repack_mxscales.argtypes = [u8p, C.c_int, C.c_int, u8p]
        L.po_petit_dequant.restype = C.c_int
        L.po_petit_dequant.argtypes = [u32p, u8p, C.c_int, C.c_int, C.c_int, f32p]
        for name in ("po_petit_weight_word_index", "po_petit_nvscale_byte_index",
                     "po_petit_mxscale_byte_index"):
            getattr(L, name).restype = C.c_size_t
            getattr(L, name).argtypes = [C.c_int, C.c_int, C.c_int]
        _lib = L
    return _lib


def ref_lib():
    """The reference's own software floats (oracle/_ref); None if not built."""
    if not _REF_PATH.exists():
        return None
    L = C.CDLL(str(_REF_PATH))
    for name in ("ref_e4m3_to_f32", "ref_e5m2_to_f32"):
        getattr(L, name).restype = C.c_float
        getattr(L, name).argtypes = [C.c_uint8]
    for name in ("ref_f32_to_bf16", "ref_f32_to_f16"):
        getattr(L, name).restype = C.c_uint16
        getattr(L, name).argtypes = [C.c_float]
    L.ref_e4m3_to_e5m3.restype = C.c_uint8
    L.ref_e4m3_to_e5m3.argtypes = [C.c_uint8]
    return L


def _p(a: np.ndarray, ctype):
    return a.ctypes.data_as(C.POINTER(ctype))


def _c(a, dtype) -> np.ndarray:
    a = np.ascontiguousarray(a)
    assert a.dtype == dtype, (a.dtype, dtype)
    return a


FP4_VALUES = np.array([0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0,
                       -0.0, -0.5, -1.0, -1.5, -2.0, -3.0, -4.0, -6.0], dtype=np.float32)


# --- dense dequant of the native checkpoint formats -------------------------

def dequant_nvfp4(q_u8: np.ndarray, s_e4m3_bits: np.ndarray) -> np.ndarray:
    """q uint8 [N, K/2], scales e4m3 bit patterns uint8 [N, K/16] -> f32 [N, K]."""
    q = _c(q_u8, np.uint8)
    s = _c(s_e4m3_bits, np.uint8)
    n, k = q.shape[0], q.shape[1] * 2
    assert s.shape == (n, k // 16)
    out = np.empty((n, k), dtype=np.float32)
    lib().po_dequant_nvfp4(_p(q, C.c_uint8), _p(s, C.c_uint8), n, k, _p(out, C.c_float))
    return out


def dequant_mxfp4(q_u8: np.ndarray, s_e8m0: np.ndarray) -> np.ndarray:
    q = _c(q_u8, np.uint8)
    s = _c(s_e8m0, np.uint8)
    n, k = q.shape[0], q.shape[1] * 2
    assert s.shape == (n, k // 32)
    out = np.empty((n, k), dtype=np.float32)
    lib().po_dequant_mxfp4(_p(q, C.c_uint8), _p(s, C.c_uint8), n, k, _p(out, C.c_float))
    return out


def gemm_ref(a_bits: np.ndarray, a_is_bf16: bool, b_dq: np.ndarray, global_scale: float):
    """a: uint16 bit patterns [M, K]; b_dq f32 [N, K].  Returns (c bits u16 [M,N], c f32)."""
    a = _c(a_bits, np.uint16)
    b = _c(b_dq, np.float32)
    m, k = a.shape
    n = b.shape[0]
    assert b.shape[1] == k
    c = np.empty((m, n), dtype=np.uint16)
    cf = np.empty((m, n), dtype=np.float32)
    lib().po_gemm_ref(_p(a, C.c_uint16), int(a_is_bf16), _p(b, C.c_float),
                      float(global_scale), m, n, k, _p(c, C.c_uint16), _p(cf, C.c_float))
    return c, cf


def fp4_gemm_cpu(a_bits, a_is_bf16, q_u8, s_u8, global_scale, fmt: str):
    """Whole CPU path (dequant + matmul) as one call; fmt in {'nvfp4', 'mxfp4'}."""
    a = _c(a_bits, np.uint16)
    q = _c(q_u8, np.uint8)
    s = _c(s_u8, np.uint8)
    m, k = a.shape
    n = q.shape[0]
    c = np.empty((m, n), dtype=np.uint16)
    rc = lib().po_fp4_gemm_cpu(_p(a, C.c_uint16), int(a_is_bf16), _p(q, C.c_uint8), _p(s, C.c_uint8),
                               float(global_scale), 0 if fmt == "nvfp4" else 1, m, n, k,
                               _p(c, C.c_uint16))
    if rc != 0:
        raise MemoryError("oracle scratch allocation failed")
    return c


def num_threads() -> int:
    return lib().po_num_threads()


# --- scalar helpers, vectorised through numpy -----------------------------------

def e4m3_to_f32(bits: np.ndarray) -> np.ndarray:
    tab = np.array([lib().po_e4m3_to_f32(i) for i in range(256)], dtype=np.float32)
    return tab[np.asarray(bits, dtype=np.uint8)]


def e8m0_to_f32(bits: np.ndarray) -> np.ndarray:
    return (np.asarray(bits, dtype=np.uint32) << 23).view(np.float32)


def f32_to_bf16_bits(x: np.ndarray) -> np.ndarray:
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    nan = (u & 0x7F800000) == 0x7F800000
    r = np.where(nan, np.where((u & 0xFFFF) != 0, u | 0x10000, u), u + 0x7FFF + ((u >> 16) & 1))
    return (r >> 16).astype(np.uint16)


def bf16_bits_to_f32(b: np.ndarray) -> np.ndarray:
    return (np.asarray(b, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)


def f16_bits_to_f32(b: np.ndarray) -> np.ndarray:
    return np.asarray(b, dtype=np.uint16).view(np.float16).astype(np.float32)


# --- the reference's packed ("Petit") formats ------------------------------------

def petit_format(words: np.ndarray) -> np.ndarray:
    w = _c(words, np.uint32).ravel()
    out = np.array([lib().po_petit_format(int(x)) for x in w], dtype=np.uint32)
    return out.reshape(np.shape(words))


def petit_repack_weights(qw_u32: np.ndarray) -> np.ndarray:
    """qw uint32 [N, K/8] -> packed uint32 (flat, N*K/8 words), reference layout."""
    qw = _c(qw_u32, np.uint32)
    n, k = qw.shape[0], qw.shape[1] * 8
    out = np.zeros(qw.size, dtype=np.uint32)
    if lib().po_petit_repack_weights(_p(qw, C.c_uint32), n, k, _p(out, C.c_uint32)) != 0:
        raise ValueError(f"reference repack needs n%32==0 and k%64==0, got n={n} k={k}")
    return out


def petit_repack_nvscales(s: np.ndarray, k: int) -> np.ndarray:
    s = _c(s, np.uint8)
    n = s.shape[0]
    out = np.zeros(s.size, dtype=np.uint8)
    if lib().po_petit_repack_nvscales(_p(s, C.c_uint8), n, k, _p(out, C.c_uint8)) != 0:
        raise ValueError(f"reference NV scale repack needs n%64==0 and k%64==0, got n={n} k={k}")
    return out


def petit_repack_mxscales(s: np.ndarray, k: int) -> np.ndarray:
    s = _c(s, np.uint8)
    n = s.shape[0]
    out = np.zeros(s.size, dtype=np.uint8)
    if lib().po_petit_repack_mxscales(_p(s, C.c_uint8), n, k, _p(out, C.c_uint8)) != 0:
        raise ValueError(f"reference MX scale repack needs n%32==0 and k%256==0, got n={n} k={k}")
    return out


def petit_dequant(w_packed: np.ndarray, s_packed: np.ndarray, fmt: str, n: int, k: int) -> np.ndarray:
    w = _c(w_packed, np.uint32)
    s = _c(s_packed, np.uint8)
    out = np.empty((n, k), dtype=np.float32)
    rc = lib().po_petit_dequant(_p(w, C.c_uint32), _p(s, C.c_uint8), 0 if fmt == "nvfp4" else 1, n, k,
                                _p(out, C.c_float))
    if rc != 0:
        raise ValueError("bad shape for petit_dequant")
    return out


# --- the MFMA-native image of NVFP4 weights ("petit-cdna4-nv6/1"): a statement of THIS build's load-time re-encoding (petit-kernel_amd/csrc/nvnative.hip,
# include/petit_amd.h "NVFP4 weights on the native class"), not of anything in the reference: it is what the native-class tests run the oracle GEMM on.

E2M3_GRID = np.array([i / 8.0 for i in range(8)] + [(1.0 + i / 8.0) * 2.0 ** e for e in range(3) for i in range(8)])


def e2m3_rne(x: np.ndarray) -> np.ndarray:
    """nearest e2m3 value (grid 0, 1/8 .. 7/8, 1 .. 1.875, 2 .. 3.75, 4 .. 7.5), ties to the even code, saturating at 7.5; sign kept."""
    ax = np.abs(np.asarray(x, dtype=np.float64))
    idx = np.searchsorted(E2M3_GRID, ax, side="left").clip(1, 31)
    lo, hi = E2M3_GRID[idx - 1], E2M3_GRID[idx]
    pick_hi = (ax - lo > hi - ax) | ((ax - lo == hi - ax) & (idx % 2 == 0))       # grid index = code: ties go to the even code
    return np.sign(x) * np.where(ax >= 7.5, 7.5, np.where(pick_hi, hi, lo))


def nv6_reencode(q_u8: np.ndarray, s_e4m3_bits: np.ndarray):
    """NVFP4 (q uint8 [N, K/2], e4m3 scale bytes [N, K/16]) -> (w [N, K] f32: the image's values WITHOUT the global scale, scale bytes [N, K/32]).
    Per 32-k block: v = fp4 x e4m3 (exact), E = floor(log2 max|v|) - 2, element = RNE_e2m3(v / 2^E), byte = E + 127 (127 for a zero block)."""
    v = dequant_nvfp4(q_u8, s_e4m3_bits).astype(np.float64)
    n, k = v.shape
    blk = v.reshape(n, k // 32, 32)
    amax = np.abs(blk).max(axis=2)
    ebits = (amax.astype(np.float32).view(np.uint32) >> 23) & 0xFF
    sbyte = np.where(amax == 0, 127, ebits.astype(np.int64) - 2)
    scale = np.ldexp(1.0, sbyte - 127)[:, :, None]
    w = e2m3_rne(blk / scale) * scale
    return w.reshape(n, k).astype(np.float32), sbyte.astype(np.uint8)
