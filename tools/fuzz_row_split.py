#!/usr/bin/env python3
"""tools/fuzz_row_split.py [seed] [seconds] [native] -- random prefill problems on which a default-pick call runs as bulk + tail (csrc/pick.hip plan_row_split), against the oracle.

tools/fuzz_parity.py checks every output of its problems and therefore keeps M N K below 4e9, where a tile grid never reaches a whole round of the chip; here the shapes are
prefill-sized (N 2048-16384, K 2048-8192, M 513-4500, all four dtype families, plain / bias / SiLU-mul epilogues) and the oracle checks a SAMPLE: the rows around the split,
the tail, the first rows and seeded others x 64 columns (tests/test_gpu_parity.py FullSizeProblem).  Prints one line per problem and a summary.
`native` (round 6): the same for the native class's plan (petit_gemm_row_split with a sentinel: bulk in the class, the short tail through the EXACT default pick) -- bf16
activations, NVFP4 weights on their attached image or MXFP4 weights, the three activation formats; bulk rows are held to the class's bounds (check_native_sampled), tail rows to
the exact class's."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "petit-kernel_amd")); sys.path.insert(0, str(ROOT))
import conftest  # noqa
import test_gpu_parity as T
import petit_kernel as pk

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
native = len(sys.argv) > 3 and sys.argv[3] == "native"
t0, n_ok, fails, tried = time.time(), 0, [], 0
while time.time() - t0 < budget:
    kind = str(rng.choice(["nv", "mx"])); is_bf16 = True if native else bool(rng.integers(0, 2))
    sid, code = [(pk.SOLUTION_AUTO_NATIVE_MXFP8, 2), (pk.SOLUTION_AUTO_NATIVE_MXFP6, 4), (pk.SOLUTION_AUTO_NATIVE_MXFP4, 6)][int(rng.integers(0, 3))] if native else (-1, 0)
    n = int(rng.choice([2048, 3072, 4096, 5120, 6144, 8192, 10240, 12288, 16384])) + int(rng.choice([0, 0, 0, 32, 224]))
    k = int(rng.choice([2048, 3072, 4096, 5120, 7168, 8192]))
    P = None
    for _ in range(40):                     # an M on which the plan fires for this shape (host arithmetic: cheap)
        m = int(rng.integers(513, 4500))
        h = pk.PetitSolutionHints(); h.a_type = h.c_type = torch.bfloat16 if is_bf16 else torch.float16
        h.b_type = pk.DataType.float4_e2m1 if kind == "nv" else pk.DataType.mxfloat4_e2m1
        mode = "plain" if native else str(rng.choice(["plain", "bias", "silu"]))
        m1 = pk.ops.auto_row_split(h, m, n, k, activation="silu_mul" if mode == "silu" else None, solution_id=sid)
        if m1:
            break
    tried += 1
    if not m1:
        continue
    tag = f"{'bf16' if is_bf16 else 'fp16'}x{kind} {n}x{k} M={m} -> {m1} + {m - m1} {mode}" + (f" native class {code}" if native else "")
    try:
        P = T.FullSizeProblem(pk, kind, n, k, int(rng.integers(1 << 30)))
        a = P.activations(m, is_bf16, int(rng.integers(1 << 30)))
        rows = np.unique(np.concatenate([np.arange(max(0, m1 - 24), min(m, m1 + 72)), np.arange(m - 24, m), np.arange(16), rng.integers(0, m, 40)]))
        sel = torch.from_numpy(rows).to("cuda")
        dtype = torch.bfloat16 if is_bf16 else torch.float16
        x = T.from_bits(a, dtype).to("cuda")
        if mode == "silu":   # gate x up of N(0, 1) activations overflows fp16 (NVFP4: |g|, |u| ~ 300; MXFP4 with block scales up to 2^9: ~ 4e4, and an output NEAR 65504 is
            x = x * (0.0625 if kind == "nv" else 2.0 ** -12)   # inf under one rounding and finite under another): powers of two are exact in both dtypes
            a = T.bits(x).copy()
        if native:
            c = P.run(a, True, sid)
            bulk, tail = rows[rows < m1], rows[rows >= m1]
            T.check_native_sampled(P, c[torch.from_numpy(bulk).to("cuda")], a[bulk], code, tag + " (bulk)")
            P.check_sampled(c[torch.from_numpy(tail).to("cuda")], a[tail], True, tag + " (tail, exact)")
        elif mode == "plain":
            c = P.mul(x, P.b, P.sp, P.gsd, m, n, k, -1)
            P.check_sampled(c[sel], a[rows], is_bf16, tag)
        else:
            # reference for the epilogues: the plain product of the SAME library call form without a split (an explicit id never splits), checked against the oracle itself
            sid = pk.ops.resolve_solution(h, m, n, k, -1)
            c0 = P.mul(x, P.b, P.sp, P.gsd, m, n, k, sid)
            P.check_sampled(c0[sel], a[rows], is_bf16, tag + " (explicit id, plain)")
            if mode == "bias":
                bias = torch.randn(n, dtype=dtype, device="cuda")
                c = P.mul(x, P.b, P.sp, P.gsd, m, n, k, -1, bias=bias)
                want = c0[sel].float() + bias.float()
            else:
                c = P.mul(x, P.b, P.sp, P.gsd, m, n, k, -1, activation="silu_mul")
                g, u = c0[sel, : n // 2].float(), c0[sel, n // 2:].float()
                want = g * torch.sigmoid(g) * u
            assert c.shape[0] == m
            want = want.to(dtype).float()      # (fp16: what overflows the output type is inf in both)
            assert torch.allclose(c[sel].float(), want, rtol=3e-2, atol=3e-2 * float(want.abs().mean()) + 2e-2), tag
        n_ok += 1
        print("ok  ", tag, flush=True)
    except Exception as exc:  # noqa: BLE001
        fails.append(tag)
        print("FAIL", tag, str(exc)[:300], flush=True)
    del P
    torch.cuda.empty_cache()
print(f"ok {n_ok} fails {len(fails)} (shapes tried {tried})")
