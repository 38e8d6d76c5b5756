"""numpy model of THIS repo's packed layout (petit-kernel_amd/csrc/layout.h).

TEST INFRASTRUCTURE ONLY.  Unlike oracle/petit_oracle.c this is not a
restatement of the reference -- the reference's own packed format is restated
there (po_petit_*).  This file is the independent, index-by-index statement of
the gfx950 layout that the HIP repack kernels are checked against bit for bit,
and the CPU "unpack" that proves the reference's own repack invariant for it:
    dequant(unpack(repack(x))) == dequant(x)
(lib/gemm/rocm/quantization/fp4/quantization_utils_fp4_test.cc:103-133).
"""
from __future__ import annotations

import numpy as np


def span_tiles_for_k(k: int) -> int:
    return 8 if k % 1024 == 0 else 4 if k % 512 == 0 else 2


def pack_weights(qw_u32: np.ndarray) -> np.ndarray:
    """u32 [N, K/8] -> u32 flat [N*K/8]: uint4 pw[N/16][K/128][lane = 16g + r][j]."""
    n, k8 = qw_u32.shape
    k = k8 * 8
    assert n % 16 == 0 and k % 128 == 0
    # qw[nt, r, kt, g, j] -> out[nt, kt, g, r, j]
    v = qw_u32.reshape(n // 16, 16, k // 128, 4, 4)
    return np.ascontiguousarray(v.transpose(0, 2, 3, 1, 4)).reshape(-1)


def unpack_weights(pw_u32: np.ndarray, n: int, k: int) -> np.ndarray:
    v = np.asarray(pw_u32, dtype=np.uint32).reshape(n // 16, k // 128, 4, 16, 4)
    return np.ascontiguousarray(v.transpose(0, 3, 1, 2, 4)).reshape(n, k // 8)


def pack_nvscales(s_u8: np.ndarray, k: int) -> np.ndarray:
    """u8 [N, K/16] -> u8 flat: ps[N/16][K/(128*KS)][lane = 16g + r][t][h]."""
    n = s_u8.shape[0]
    ks = span_tiles_for_k(k)
    assert n % 16 == 0 and k % 256 == 0 and s_u8.shape[1] == k // 16
    # s[nt, r, sp, t, g, h] -> out[nt, sp, g, r, t, h]
    v = s_u8.reshape(n // 16, 16, k // (128 * ks), ks, 4, 2)
    return np.ascontiguousarray(v.transpose(0, 2, 4, 1, 3, 5)).reshape(-1)


def unpack_nvscales(ps_u8: np.ndarray, n: int, k: int) -> np.ndarray:
    ks = span_tiles_for_k(k)
    v = np.asarray(ps_u8, dtype=np.uint8).reshape(n // 16, k // (128 * ks), 4, 16, ks, 2)
    return np.ascontiguousarray(v.transpose(0, 3, 1, 4, 2, 5)).reshape(n, k // 16)


def pack_mxscales(s_u8: np.ndarray, k: int) -> np.ndarray:
    """u8 [N, K/32] -> u8 flat: ps[N/16][K/(128*KS)][lane = 16g + r][t]."""
    n = s_u8.shape[0]
    ks = span_tiles_for_k(k)
    assert n % 16 == 0 and k % 256 == 0 and s_u8.shape[1] == k // 32
    # s[nt, r, sp, t, g] -> out[nt, sp, g, r, t]
    v = s_u8.reshape(n // 16, 16, k // (128 * ks), ks, 4)
    return np.ascontiguousarray(v.transpose(0, 2, 4, 1, 3)).reshape(-1)


def unpack_mxscales(ps_u8: np.ndarray, n: int, k: int) -> np.ndarray:
    ks = span_tiles_for_k(k)
    v = np.asarray(ps_u8, dtype=np.uint8).reshape(n // 16, k // (128 * ks), 4, 16, ks)
    return np.ascontiguousarray(v.transpose(0, 3, 1, 4, 2)).reshape(n, k // 32)


# --- literal per-element index functions (mirror of layout.h, used to cross-check
# --- the reshape/transpose statements above) ---------------------------------------

def weight_word_index(k_total: int, n: int, k8: int) -> int:
    nt, r = divmod(n, 16)
    kt, rem = divmod(k8, 16)
    g, j = divmod(rem, 4)
    tile = nt * (k_total // 128) + kt
    return (tile * 64 + g * 16 + r) * 4 + j


def nvscale_byte_index(k_total: int, n: int, grp: int) -> int:
    ks = span_tiles_for_k(k_total)
    nt, r = divmod(n, 16)
    kt, rem = divmod(grp, 8)
    g, h = divmod(rem, 2)
    sp, t = divmod(kt, ks)
    rec = (nt * (k_total // (128 * ks)) + sp) * 64 + g * 16 + r
    return rec * (ks * 2) + t * 2 + h


def mxscale_byte_index(k_total: int, n: int, blk: int) -> int:
    ks = span_tiles_for_k(k_total)
    nt, r = divmod(n, 16)
    kt, g = divmod(blk, 4)
    sp, t = divmod(kt, ks)
    rec = (nt * (k_total // (128 * ks)) + sp) * 64 + g * 16 + r
    return rec * ks + t
