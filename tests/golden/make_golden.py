"""Generate the golden vectors under tests/golden/ from the REFERENCE's own
Python oracle (tests/ops/test_fp4_gemm_quark.py:9-24), imported from
/root/reference in the build container.  The reference source never travels:
only the data this script emits is committed.

Run:  python tests/golden/make_golden.py        (needs /root/reference)

What is pinned
  * nv_<m>_<n>_<k>_<seed>.npz  -- the reference pytest's NVFP4 cases
    (tests/ops/test_fp4_gemm_quark.py:27-30) plus an fp16-activation twin.
    Inputs are drawn with the reference's distributions (:41-46) from the
    torch *CPU* generator (the reference draws on the GPU generator, whose
    stream is unrecoverable without the GPU).  `b_ref` and `c_ref` are the
    outputs of the reference's `_dequant_nvfp4` and `_gemm_ref`.
  * mx_<m>_<n>_<k>_<seed>.npz  -- the MXFP4 cases (:32-35).  The reference
    delegates MX dequant to amd-quark's `dq_mxfp4` (:69,83), which is not in
    /root/reference and not installed: PARITY UNPINNED at that boundary.  The
    expected values here come from torch's own float8_e8m0fnu decode and the
    reference's `_gemm_ref`, i.e. an independent implementation of the
    semantics stated in lib/gemm/rocm/quantization/dequant.cuh:198-203.
  * config1_nv_1_4096_4096.npz -- BASELINE.json configs[0]; inputs are
    regenerated from numpy seeds in the test (8 MiB of weights is not a
    fixture), only c_ref and input checksums are stored.
  * dequant_tables.npz -- 16 fp4 codes x 126 positive e4m3 scales and
    16 codes x e8m0 1..237 (quantization_utils_fp4_test.cc:246-278,344-365),
    values from the reference's LUT * torch's float8 decodes.
"""
import hashlib
import importlib.util
import sys
import types
from pathlib import Path

import numpy as np
import torch

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent


def load_reference_oracle():
    # `import petit_kernel` in the reference test needs the unbuilt extension;
    # the two functions we want do not use it.
    sys.modules.setdefault("petit_kernel", types.ModuleType("petit_kernel"))
    spec = importlib.util.spec_from_file_location(
        "ref_test_fp4", REF / "tests/ops/test_fp4_gemm_quark.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def bits16(t: torch.Tensor) -> np.ndarray:
    return t.contiguous().view(torch.int16).numpy().view(np.uint16).copy()


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def nv_case(ref, m, n, k, seed, dtype):
    torch.manual_seed(seed)
    a = torch.randn((m, k), dtype=torch.bfloat16).to(dtype)
    q = torch.randint(0, 256, (n, k // 2), dtype=torch.uint8)
    s = (torch.rand((n, k // 16)) * 3.5 + 0.25).to(torch.float8_e4m3fn)
    gs = torch.rand((1,), dtype=torch.float32) * 1.5 + 0.5
    b_ref = ref._dequant_nvfp4(q, s) * gs.item()
    c_ref = ref._gemm_ref(a, b_ref)
    return dict(a=bits16(a), q=q.numpy(), s=s.view(torch.uint8).numpy(),
                gs=gs.numpy(), b_dq=ref._dequant_nvfp4(q, s).numpy(),
                c_ref=bits16(c_ref), a_is_bf16=np.array(dtype == torch.bfloat16))


def mx_case(ref, m, n, k, seed, dtype):
    torch.manual_seed(seed)
    a = torch.randn((m, k), dtype=torch.bfloat16).to(dtype)
    q = torch.randint(0, 256, (n, k // 2), dtype=torch.uint8)
    s = torch.randint(1, 238, (n, k // 32), dtype=torch.uint8)
    gs = torch.rand((1,), dtype=torch.float32) * 1.5 + 0.5
    lut = torch.tensor([0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0,
                        -0.0, -0.5, -1.0, -1.5, -2.0, -3.0, -4.0, -6.0])
    dq = torch.empty((n, k), dtype=torch.float32)
    dq[:, 0::2] = lut[(q & 0xF).long()]
    dq[:, 1::2] = lut[(q >> 4).long()]
    sc = s.view(torch.float8_e8m0fnu).float()
    dq = (dq.view(n, -1, 32) * sc.unsqueeze(-1)).view(n, -1)
    # reference MX test: ((a.float() @ b.T.float()) * gs).to(dtype)  (:87)
    c_ref = ((a.float() @ dq.t().float()) * gs.item()).to(a.dtype)
    # and the imported _gemm_ref itself on the gs-scaled weights, the way the reference's NV test calls it (:49-50)
    c_ref_gemm = ref._gemm_ref(a, dq * gs.item())
    return dict(a=bits16(a), q=q.numpy(), s=s.numpy(), gs=gs.numpy(),
                b_dq=dq.numpy(), c_ref=bits16(c_ref), c_ref_gemm=bits16(c_ref_gemm),
                a_is_bf16=np.array(dtype == torch.bfloat16))


def config1(ref):
    # numpy-seeded so the test can regenerate the 8 MiB inputs bit-for-bit.
    m, n, k = 1, 4096, 4096
    rng = np.random.default_rng(1234)
    a = torch.from_numpy(rng.standard_normal((m, k), dtype=np.float32)).bfloat16()
    q = rng.integers(0, 256, (n, k // 2), dtype=np.uint8)
    s_f = (rng.random((n, k // 16), dtype=np.float32) * 3.5 + 0.25)
    s = torch.from_numpy(s_f).to(torch.float8_e4m3fn)
    gs = np.float32(rng.random() * 1.5 + 0.5)
    b_ref = ref._dequant_nvfp4(torch.from_numpy(q), s) * float(gs)
    c_ref = ref._gemm_ref(a, b_ref)
    s_bits = s.view(torch.uint8).numpy()
    return dict(c_ref=bits16(c_ref), gs=np.array([gs], dtype=np.float32),
                sha_a=np.array(sha(bits16(a))), sha_q=np.array(sha(q)),
                sha_s=np.array(sha(s_bits)))


def tables():
    lut = np.array([0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0,
                    -0.0, -0.5, -1.0, -1.5, -2.0, -3.0, -4.0, -6.0], dtype=np.float32)
    e4 = torch.arange(1, 0x7F, dtype=torch.uint8).view(torch.float8_e4m3fn).float().numpy()
    e8 = torch.arange(1, 238, dtype=torch.uint8).view(torch.float8_e8m0fnu).float().numpy()
    return dict(nv=np.outer(lut, e4).astype(np.float32),    # [16, 126]
                mx=np.outer(lut, e8).astype(np.float32))    # [16, 237]


def main():
    ref = load_reference_oracle()
    for (m, n, k, seed) in ref.NVFP4_CASES:
        np.savez_compressed(OUT / f"nv_{m}_{n}_{k}_{seed}.npz", **nv_case(ref, m, n, k, seed, torch.bfloat16))
        np.savez_compressed(OUT / f"nv_{m}_{n}_{k}_{seed}_f16.npz", **nv_case(ref, m, n, k, seed, torch.float16))
    for (m, n, k, seed) in ref.MXFP4_CASES:
        np.savez_compressed(OUT / f"mx_{m}_{n}_{k}_{seed}.npz", **mx_case(ref, m, n, k, seed, torch.bfloat16))
        np.savez_compressed(OUT / f"mx_{m}_{n}_{k}_{seed}_f16.npz", **mx_case(ref, m, n, k, seed, torch.float16))
    np.savez_compressed(OUT / "config1_nv_1_4096_4096.npz", **config1(ref))
    np.savez_compressed(OUT / "dequant_tables.npz", **tables())
    for p in sorted(OUT.glob("*.npz")):
        print(p.name, p.stat().st_size)


if __name__ == "__main__":
    main()
