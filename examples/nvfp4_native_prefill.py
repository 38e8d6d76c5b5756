#!/usr/bin/env python3
"""examples/nvfp4_native_prefill.py -- an NVFP4 linear layer that decodes on the exact kernels and prefills on the block-scaled MFMA (round 6; needs an MI355X).

Load time: the reference's two repack calls, then ONE more launch builds the weights' MFMA-native image (FP6 e2m3 + one E8M0 scale per 32 k) and attaches it to the packed
weights.  Forward: `solution_id = -1` (exact) for small batches, `-2` (MXFP8 activations on the image) from `native_min_m` tokens on -- the same `mul_nvfp4_a16` call.

    python examples/nvfp4_native_prefill.py
"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "petit-kernel_amd"))
import petit_kernel  # noqa: E402


class PetitNvFp4Linear(torch.nn.Module):
    def __init__(self, qweight, weight_scale, weight_scale_2, native_min_m: int = 256):
        super().__init__()
        self.size_n, self.size_k, self.native_min_m = qweight.shape[0], qweight.shape[1] * 2, native_min_m
        self.register_buffer("b", petit_kernel.repack_nvfp4(qweight.view(torch.int32), self.size_n, self.size_k))
        self.register_buffer("s", petit_kernel.process_nvfp4_scales(weight_scale, self.size_n, self.size_k))
        self.register_buffer("global_scale", weight_scale_2.reshape(1).float())
        # opt-in, per weight: the image stays this module's memory; attaching it lets -2 / -4 / -3 name the native class on this weight
        self.register_buffer("image", petit_kernel.nvfp4_native_image(self.b, self.s, self.size_n, self.size_k))
        petit_kernel.attach_nvfp4_native(self.b, self.image)

    def forward(self, x):
        x2 = x.reshape(-1, self.size_k)
        sid = -2 if x2.shape[0] >= self.native_min_m else -1
        y = petit_kernel.mul_nvfp4_a16(x2, self.b, self.s, self.global_scale, x2.shape[0], self.size_n, self.size_k, sid)
        return y.reshape(*x.shape[:-1], self.size_n)


def main() -> None:
    dev = torch.device("cuda")
    n, k = 8192, 8192
    g = torch.Generator().manual_seed(0)
    # checkpoint-like NVFP4: bell-shaped weights quantised by the nvidia/*-FP4 recipe (tools/quantize_weights.py)
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tools"))
    import quantize_weights as QW
    w = QW.synthetic_weights(n, k, seed=0)
    q, ws, ws2 = QW.quantize_nvfp4(w)
    layer = PetitNvFp4Linear(torch.from_numpy(q).to(dev), torch.from_numpy(ws).to(dev).view(torch.float8_e4m3fn), torch.tensor(float(ws2)).to(dev))
    wq = torch.from_numpy(QW.dequantize("nvfp4", q, ws, float(ws2))).float()
    for tokens in (4, 64, 2048):
        x = torch.randn((tokens, k), generator=g).bfloat16()
        y = layer(x.to(dev)).float().cpu()
        ref = x.float() @ wq.T
        rel = ((y - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
        print(f"{tokens:5d} tokens ({'native class on the image' if tokens >= layer.native_min_m else 'exact kernels'}): rms error vs the dequantised-weight reference {rel:.4f}")
        assert rel < (0.06 if tokens >= layer.native_min_m else 0.01), "MISMATCH"
    x = torch.randn((4096, k), generator=g).bfloat16().to(dev)
    for sid, name in ((-1, "exact NVFP4"), (-2, "image x MXFP8"), (-4, "image x MXFP6")):
        for _ in range(3):
            petit_kernel.mul_nvfp4_a16(x, layer.b, layer.s, layer.global_scale, 4096, n, k, sid)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            petit_kernel.mul_nvfp4_a16(x, layer.b, layer.s, layer.global_scale, 4096, n, k, sid)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / 20 * 1e6
        print(f"M = 4096, {name:14s}: {us:7.1f} us  = {2.0 * 4096 * n * k / us / 1e6:6.0f} TFLOP/s")
    petit_kernel.attach_nvfp4_native(layer.b, None)


if __name__ == "__main__":
    main()
