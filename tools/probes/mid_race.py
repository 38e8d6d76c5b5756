"""tools/probes/mid_race.py m n k -- run every shared-activation-tile kernel (gemm_mid.hpp) many times on one problem and
count launches whose output differs grossly from the staged streaming kernel's."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "petit-kernel_amd")); sys.path.insert(0, str(ROOT / "tools"))
import torch
import petit_kernel as pk
from petit_kernel import _lib

m, n, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda", 0)
torch.manual_seed(0)
a = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
q = torch.randint(0, 256, (n, k // 2), device=dev, dtype=torch.uint8)
s = (torch.rand(n, k // 16, device=dev) * 3 + 0.25).to(torch.float8_e4m3fn)
gs = torch.tensor([1.0], device=dev)
b = pk.repack_nvfp4(q.view(torch.int32), size_n=n, size_k=k)
ps = pk.process_nvfp4_scales(scales=s, size_n=n, size_k=k)
h = pk.PetitSolutionHints(); h.a_type = h.c_type = torch.bfloat16; h.b_type = pk.DataType.float4_e2m1
sols = pk.ops.get_fp4_solutions(h, m, n, k)
ref_sid = next(x for x in sols if (x >> 48) & 0xF in (10, 11) and (x >> 36) & 0xF == 1)
ref = pk.mul_nvfp4_a16(a, b, ps, gs, m, n, k, ref_sid).float()
for sid in [x for x in sols if (x >> 36) & 0xF == 2]:
    bad, worst = 0, 0.0
    for it in range(50):
        c = pk.mul_nvfp4_a16(a, b, ps, gs, m, n, k, sid).float()
        d = (c - ref).abs().max().item()
        if d > 0.05 * ref.abs().max().item():
            bad += 1
            worst = max(worst, d)
    print(f"{sid:#x} {_lib.describe_solution(sid)[:80]} bad {bad}/50 worst {worst:.3g}", flush=True)
