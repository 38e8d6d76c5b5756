#!/usr/bin/env python3
"""tools/quantize_weights.py -- CPU weight quantiser: 16-bit weights -> NVFP4 / MXFP4 in the source formats the library ingests.

    python tools/quantize_weights.py --n 4096 --k 4096 --fmt nvfp4 [--out w.npz] [--seed 0]

Why it exists: the accuracy budgets of rounds 3-4 drew the packed nibbles and the scale bytes UNIFORMLY at random.  A real FP4
checkpoint is quantised from bell-shaped weights: the code histogram is peaked at the small codes, every block's scale follows the block's own
largest weight, and the quantisation error of a weight is correlated with its size -- none of which a uniform draw has.  The reference's accuracy
claim (README.md:3: MMLU 82.15 -> 80.79) is about such a checkpoint.  This module makes checkpoint-LIKE weights without one (there is no network):
  synthetic_weights   bf16 weights ~ N(0, sigma^2), sigma = 1 / sqrt(K), a fraction of heavy-tailed output rows (x 3..8) and a sprinkle of
                      Student-t outliers, the shape LLM linear layers have;
  quantize_nvfp4      the `nvidia/*-FP4` (TensorRT Model Optimizer) recipe SURVEY.md section 8 f4 names: weight_scale_2 = amax / (6 * 448) (f32, one
                      per tensor), weight_scale[n][k / 16] = e4m3(block_amax / 6 / weight_scale_2), q = e2m1(w / (weight_scale * weight_scale_2)),
                      round to nearest even; packed uint8 [N, K / 2], low nibble = even k (tests/ops/test_fp4_gemm_quark.py:15-19);
  quantize_mxfp4      OCP MX: one e8m0 scale per 32 weights, 2^ceil(log2(block_amax / 6)) (no element clips), q = e2m1(w / scale).
Outputs feed petit_kernel.repack_nvfp4 / process_nvfp4_scales (repack_mxfp4 / process_mxfp4_scales) unchanged: the `global_scale` argument of
mul_nvfp4_a16 is weight_scale_2.  Used by tests/test_gpu_parity.py::test_*_accuracy_budget_checkpoint_like_weights.
"""
from __future__ import annotations

import argparse
import json

import numpy as np

FP4_VALUES = np.array([0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0], dtype=np.float64)
# midpoints between neighbouring magnitudes; a tie goes to the EVEN code (0 / 1.0 / 2.0 / 4.0): round-to-nearest-even on the e2m1 grid
_MID = (FP4_VALUES[1:] + FP4_VALUES[:-1]) / 2


def e2m1_rne(x: np.ndarray) -> np.ndarray:
    """nearest e2m1 code (sign in bit 3) of every element, ties to the even mantissa, saturating at 6; -0 -> code 8 is avoided (0 -> code 0)."""
    ax = np.abs(x).astype(np.float64)
    idx = np.searchsorted(_MID, ax, side="left")            # ax == mid -> lower index
    tie = np.isin(ax, _MID)
    # at a tie `idx` is the lower code; take the upper one when the lower is odd
    idx = np.where(tie & (idx % 2 == 1), idx + 1, idx)
    idx = np.minimum(idx, 7)
    sign = (np.signbit(x) & (idx != 0)).astype(np.uint8)
    return (idx.astype(np.uint8) | (sign << 3)).astype(np.uint8)


def pack_nibbles(codes: np.ndarray) -> np.ndarray:
    """[N, K] codes -> uint8 [N, K / 2], low nibble = even k."""
    return (codes[:, 0::2] | (codes[:, 1::2] << 4)).astype(np.uint8)


def bf16_round(x: np.ndarray) -> np.ndarray:
    """f32 -> the nearest bf16 value (RNE), as f32."""
    u = x.astype(np.float32).view(np.uint32)
    r = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return r.view(np.float32)


def synthetic_weights(n: int, k: int, seed: int = 0, heavy_rows: float = 0.01, outliers: float = 2e-4) -> np.ndarray:
    """bf16-valued f32 [N, K]: N(0, 1/K) body, `heavy_rows` of the output rows scaled by 3..8, `outliers` of the elements from a Student-t (3 dof)."""
    rng = np.random.default_rng(seed)
    w = rng.standard_normal((n, k), dtype=np.float32) / np.float32(np.sqrt(k))
    rows = rng.choice(n, max(1, int(round(heavy_rows * n))), replace=False)
    w[rows] *= rng.uniform(3.0, 8.0, (len(rows), 1)).astype(np.float32)
    cnt = int(outliers * n * k)
    if cnt:
        pos = rng.choice(n * k, cnt, replace=False)
        w.reshape(-1)[pos] = (rng.standard_t(3, cnt) * 4.0 / np.sqrt(k)).astype(np.float32)
    return bf16_round(w)


def _e4m3_rne(x: np.ndarray) -> np.ndarray:
    """f32 -> e4m3fn bytes (RNE, |x| <= 448 assumed) through torch's cast -- the cast the checkpoint recipe itself uses."""
    import torch
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(torch.float8_e4m3fn).view(torch.uint8).numpy()


def _e4m3_to_f32(b: np.ndarray) -> np.ndarray:
    import torch
    return torch.from_numpy(np.ascontiguousarray(b)).view(torch.float8_e4m3fn).float().numpy()


def quantize_nvfp4(w: np.ndarray):
    """-> (q uint8 [N, K/2], weight_scale e4m3 bytes [N, K/16], weight_scale_2 float)."""
    n, k = w.shape
    assert k % 16 == 0
    w = w.astype(np.float32)
    amax = float(np.abs(w).max())
    ws2 = amax / (6.0 * 448.0) if amax > 0 else 1.0
    blk = np.abs(w).reshape(n, k // 16, 16).max(axis=2)
    s_bytes = _e4m3_rne(blk / 6.0 / ws2)
    s = _e4m3_to_f32(s_bytes).astype(np.float64) * ws2
    s_safe = np.where(s > 0, s, 1.0)
    codes = e2m1_rne(w.reshape(n, k // 16, 16) / s_safe[:, :, None]).reshape(n, k)
    return pack_nibbles(codes), s_bytes, ws2


def quantize_mxfp4(w: np.ndarray):
    """-> (q uint8 [N, K/2], e8m0 bytes [N, K/32], 1.0)."""
    n, k = w.shape
    assert k % 32 == 0
    w = w.astype(np.float32)
    blk = np.abs(w).reshape(n, k // 32, 32).max(axis=2).astype(np.float64)
    e = np.where(blk > 0, np.ceil(np.log2(np.maximum(blk, 1e-300) / 6.0)), -127.0)
    e = np.clip(e, -126, 127)
    s_bytes = (e + 127).astype(np.uint8)
    codes = e2m1_rne(w.reshape(n, k // 32, 32) / np.exp2(e)[:, :, None]).reshape(n, k)
    return pack_nibbles(codes), s_bytes, 1.0


def dequantize(fmt: str, q: np.ndarray, s_bytes: np.ndarray, gs: float) -> np.ndarray:
    """the f64 weights the packed tensors stand for (the oracle's dequant, restated here so that the tool is self-contained)."""
    n = q.shape[0]
    codes = np.empty((n, q.shape[1] * 2), dtype=np.uint8)
    codes[:, 0::2], codes[:, 1::2] = q & 15, q >> 4
    v = FP4_VALUES[codes & 7] * np.where(codes & 8, -1.0, 1.0)
    if fmt == "nvfp4":
        s = _e4m3_to_f32(s_bytes).astype(np.float64) * gs
        return (v.reshape(n, -1, 16) * s[:, :, None]).reshape(n, -1)
    s = np.exp2(s_bytes.astype(np.float64) - 127.0) * gs
    return (v.reshape(n, -1, 32) * s[:, :, None]).reshape(n, -1)


def stats(fmt: str, w: np.ndarray, q: np.ndarray, s_bytes: np.ndarray, gs: float) -> dict:
    dq = dequantize(fmt, q, s_bytes, gs)
    err = dq - w
    hist = np.bincount(np.concatenate([q & 15, q >> 4], axis=None) & 7, minlength=8)
    return {"fmt": fmt, "shape": list(w.shape), "weight_rms": float(np.sqrt(np.mean(w.astype(np.float64) ** 2))),
            "quantisation_rms_err_over_weight_rms": float(np.sqrt(np.mean(err ** 2)) / np.sqrt(np.mean(w.astype(np.float64) ** 2))),
            "magnitude_code_histogram": (hist / hist.sum()).round(4).tolist(), "global_scale": gs,
            "scale_byte_min_max": [int(s_bytes.min()), int(s_bytes.max())]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--k", type=int, default=4096)
    ap.add_argument("--fmt", default="nvfp4", choices=["nvfp4", "mxfp4"])
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    w = synthetic_weights(args.n, args.k, args.seed)
    q, s, gs = (quantize_nvfp4 if args.fmt == "nvfp4" else quantize_mxfp4)(w)
    print(json.dumps(stats(args.fmt, w, q, s, gs)))
    if args.out:
        np.savez_compressed(args.out, q=q, s=s, gs=np.float32(gs), w=w)


if __name__ == "__main__":
    main()
