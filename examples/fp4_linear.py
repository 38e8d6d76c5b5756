#!/usr/bin/env python3
"""examples/fp4_linear.py -- how a serving stack uses the drop-in surface (the shape of SGLang's / vLLM's
"petit" NVFP4 linear method, SURVEY.md section 8b "What calls it"): repack once at load time, one
mul_nvfp4_a16 per forward.  Needs an MI355X.

    python examples/fp4_linear.py            # random NVFP4 layer, checks against a dequantised torch matmul
"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "petit-kernel_amd"))
import petit_kernel  # noqa: E402


class PetitNvFp4Linear(torch.nn.Module):
    """y = x @ dequant(W)^T * weight_scale_2 (+ bias); W in NVFP4 (uint8 [N, K/2] + e4m3 [N, K/16] scales)."""

    def __init__(self, qweight: torch.Tensor, weight_scale: torch.Tensor, weight_scale_2: torch.Tensor, bias=None):
        super().__init__()
        self.size_n, self.size_k = qweight.shape[0], qweight.shape[1] * 2
        # load time: the two repack entry points of the reference API, outputs are opaque to the caller
        self.register_buffer("b", petit_kernel.repack_nvfp4(qweight.view(torch.int32), size_n=self.size_n, size_k=self.size_k))
        self.register_buffer("s", petit_kernel.process_nvfp4_scales(scales=weight_scale, size_n=self.size_n, size_k=self.size_k))
        self.register_buffer("global_scale", weight_scale_2.reshape(1).float())
        self.bias = bias

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x2 = x.reshape(-1, self.size_k)
        y = petit_kernel.mul_nvfp4_a16(a=x2, b=self.b, s=self.s, global_scale=self.global_scale, size_m=x2.shape[0],
                                       size_n=self.size_n, size_k=self.size_k, solution_id=-1,
                                       bias=self.bias)          # fused; the reference needs a separate `y + bias`
        return y.reshape(*x.shape[:-1], self.size_n)


def main() -> None:
    dev = torch.device("cuda")
    n, k = 4096, 4096
    g = torch.Generator().manual_seed(0)
    q = torch.randint(0, 256, (n, k // 2), generator=g, dtype=torch.uint8)
    ws = (torch.rand((n, k // 16), generator=g) * 3.5 + 0.25).to(torch.float8_e4m3fn)
    ws2 = torch.tensor(0.75)
    bias = torch.randn(n, generator=g).bfloat16()
    layer = PetitNvFp4Linear(q.to(dev), ws.to(dev), ws2.to(dev), bias.to(dev))
    # dense reference of the same layer (tests/ops/test_fp4_gemm_quark.py:9-24 of the reference)
    lut = torch.tensor([0, .5, 1, 1.5, 2, 3, 4, 6, -0., -.5, -1, -1.5, -2, -3, -4, -6])
    w = torch.stack((lut[(q & 15).long()], lut[(q >> 4).long()]), dim=-1).reshape(n, k)
    w = (w.reshape(n, k // 16, 16) * ws.float()[..., None]).reshape(n, k) * ws2
    for batch in (1, 7, 64):
        x = torch.randn((batch, k), generator=g).bfloat16()
        y = layer(x.to(dev)).float().cpu()
        ref = x.float() @ w.T + bias.float()
        err = ((y - ref).abs() / ref.abs().clamp_min(1.0)).max().item()
        print(f"batch {batch:3d}: max rel err vs dense f32 reference {err:.2e}")
        assert err < 2e-2


if __name__ == "__main__":
    main()
