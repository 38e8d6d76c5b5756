"""tools/probes/native_race.py m n k [reps] -- run every native-FP4 kernel (gemm_native.hpp / gemm_native32.hpp: counted vmcnt waits,
raw barriers, LDS stages reused every PF + 1 stages) `reps` times on one problem and count launches whose output is not
bit-identical to the kernel's own first launch (the kernels are deterministic: any difference is a race)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "petit-kernel_amd"))
import torch
import petit_kernel as pk
from petit_kernel import _lib

m, n, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 30
dev = torch.device("cuda", 0)
torch.manual_seed(0)
a = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
q = torch.randint(0, 256, (n, k // 2), device=dev, dtype=torch.uint8)
s = torch.randint(119, 136, (n, k // 32), device=dev, dtype=torch.uint8)
gs = torch.tensor([1.0], device=dev)
b = pk.repack_mxfp4(q.view(torch.int32), size_n=n, size_k=k)
ps = pk.process_mxfp4_scales(scales=s, size_n=n, size_k=k)
pk.ops.enable_native_fp4(True)
h = pk.PetitSolutionHints(); h.a_type = h.c_type = torch.bfloat16; h.b_type = pk.DataType.mxfloat4_e2m1
sols = [x for x in pk.ops.get_fp4_solutions(h, m, n, k) if (x >> 48) & 0xF in (9, 13)]
# something else streams through L2 between launches so that timing varies
junk = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
total_bad = 0
for sid in sols:
    ref = pk.mul_mxfp4_a16(a, b, ps, gs, m, n, k, sid).clone()
    bad = 0
    for it in range(reps):
        if it % 3 == 0:
            junk.add_(1)
        c = pk.mul_mxfp4_a16(a, b, ps, gs, m, n, k, sid)
        bad += int(not torch.equal(c.view(torch.int16), ref.view(torch.int16)))
    total_bad += bad
    print(f"{sid:#x} {_lib.describe_solution(sid)[:95]} bad {bad}/{reps}", flush=True)
print("TOTAL bad", total_bad)
