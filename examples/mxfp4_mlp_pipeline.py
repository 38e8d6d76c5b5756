#!/usr/bin/env python3
"""examples/mxfp4_mlp_pipeline.py -- the round-3 extensions on one gated-MLP block with MXFP4 weights (needs an MI355X):

  1. exact:     bf16 activations end to end, SiLU-mul fused into gate_up's epilogue            (solution_id = -1)
  2. pipeline:  the opt-in native FP4 class -- quantise x once, gate_up emits the quantised h, down reads it (solution_id = -4 / -2 / -3: MXFP6 / MXFP8 / MXFP4 activations)
  3. grouped:   gate and up kept as two tensors, one launch for both (decode batch)              (mul_fp4_a16_grouped)
  4. tuning:    petit_kernel.tune_tensors on the layer's own tensors; solution_id = -1 uses the winner from then on

    python examples/mxfp4_mlp_pipeline.py
"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "petit-kernel_amd"))
import petit_kernel as pk  # noqa: E402


def mx_weights(n, k, g, dev):
    q = torch.randint(0, 256, (n, k // 2), generator=g, dtype=torch.uint8).to(dev)
    s = torch.randint(122, 127, (n, k // 32), generator=g, dtype=torch.uint8).to(dev)
    return pk.repack_mxfp4(q.view(torch.int32), n, k), pk.process_mxfp4_scales(s, n, k)


def main() -> None:
    dev = torch.device("cuda")
    hidden, inter, m = 4096, 14336, 256
    g = torch.Generator().manual_seed(0)
    b1, s1 = mx_weights(2 * inter, hidden, g, dev)        # gate_up: rows [0, inter) = gate, [inter, 2 inter) = up
    b2, s2 = mx_weights(hidden, inter, g, dev)            # down
    gs = torch.tensor([0.03], device=dev)
    x = torch.randn((m, hidden), generator=g).bfloat16().to(dev)

    h = pk.mul_mxfp4_a16(x, b1, s1, gs, m, 2 * inter, hidden, -1, activation="silu_mul")
    y_exact = pk.mul_mxfp4_a16(h, b2, s2, gs, m, hidden, inter, -1)

    # the native class, three activation formats: MXFP6 (e2m3: e4m3's mantissa at the instruction's FP4 rate) is the one to serve with
    for fmt, sentinel in (("mxfp6", pk.SOLUTION_AUTO_NATIVE_MXFP6), ("mxfp8", pk.SOLUTION_AUTO_NATIVE_MXFP8), ("mxfp4", pk.SOLUTION_AUTO_NATIVE_MXFP4)):
        xq = pk.quantize_activations(x, fmt)
        hq = pk.mul_mxfp4_native(xq, b1, s1, gs, m, 2 * inter, hidden, sentinel, activation="silu_mul", out_quantized=fmt)
        y_q = pk.mul_mxfp4_native(hq, b2, s2, gs, m, hidden, inter, sentinel)
        rel = ((y_q.float() - y_exact.float()).pow(2).mean().sqrt() / y_exact.float().pow(2).mean().sqrt()).item()
        print(f"native pipeline, activations -> {fmt}: rms difference to the exact kernels {100 * rel:.1f} % of the output rms (an accuracy class of its own: DESIGN.md 3.3)")

    # decode batch, gate and up as two tensors: two launches vs one
    bg, sg = mx_weights(inter, hidden, g, dev)
    bu, su = mx_weights(inter, hidden, g, dev)
    xd = x[:4].contiguous()
    sep = [pk.mul_mxfp4_a16(xd, bg, sg, gs, 4, inter, hidden, -1), pk.mul_mxfp4_a16(xd, bu, su, gs, 4, inter, hidden, -1)]
    grp = pk.mul_fp4_a16_grouped("mxfp4", xd, [(bg, sg, gs, inter), (bu, su, gs, inter)], 4, hidden)
    print("grouped launch of gate and up:", "matches the separate calls" if all(torch.allclose(a.float(), b.float(), rtol=1e-2, atol=1e-2) for a, b in zip(sep, grp)) else "MISMATCH")

    # tune this layer's `down` at the decode batch on its own tensors; -1 picks the winner from now on
    hints = pk.PetitSolutionHints()
    hints.a_type = hints.c_type = torch.bfloat16
    hints.b_type = pk.DataType.mxfloat4_e2m1
    before = pk.ops.resolve_solution(hints, 4, hidden, inter, -1)
    t0 = time.time()
    sid, us = pk.tune_tensors(h[:4].contiguous(), (b2, s2), gs, 4, hidden, inter, kind="mxfp4")
    print(f"tune_tensors: {time.time() - t0:.2f} s, winner 0x{sid:x} at {us:.2f} us (heuristic pick was 0x{before:x}); "
          f"solution_id = -1 now resolves to 0x{pk.ops.resolve_solution(hints, 4, hidden, inter, -1):x}")


if __name__ == "__main__":
    main()
