#!/usr/bin/env python3
"""Launch the default (or a given) solution a few times, eagerly, for profiling under rocprofv3.

    rocprofv3 --kernel-trace --stats ... -- python3 tools/profile_one.py --n 8192 --k 8192 --m 1
    rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU ... -- python3 tools/profile_one.py ...
"""
import argparse
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "petit-kernel_amd"))
import torch

from petit_kernel import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=8192)
ap.add_argument("--k", type=int, default=8192)
ap.add_argument("--m", type=int, default=1)
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--solution", default="auto")
ap.add_argument("--fmt", default="nv")
ap.add_argument("--native", action="store_true", help="mx only: run the fastest-looking native-FP4 kernel (largest tile)")
ap.add_argument("--sentinel", default="", help="mxfp8 / mxfp6 / mxfp4: the native class's default pick (NVFP4 weights: on their MFMA-native image, built and attached here)")
a = ap.parse_args()
dev = torch.device("cuda", 0)
n, k, m = a.n, a.k, a.m
group = 16 if a.fmt == "nv" else 32
copies = max(2, (320 << 20) // (n * k // 2) + 2)
gen = torch.Generator(device=dev).manual_seed(1)
packed = []
for _ in range(copies):
    b = torch.randint(-2 ** 31, 2 ** 31 - 1, (n // 16, 2 * k), generator=gen, dtype=torch.int32, device=dev)
    if a.fmt == "nv":
        sp = (torch.rand((n, k // 16), generator=gen, device=dev) * 3.5 + 0.25).to(torch.float8_e4m3fn)
    else:
        sp = torch.randint(119, 136, (n // 32, k), generator=gen, dtype=torch.uint8, device=dev)
    packed.append((b, sp))
x = torch.randn((m, k), device=dev).bfloat16()
c = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
gs = torch.ones(1, device=dev)
hints = _lib.SolutionHints(_lib.CXX_DTYPE_BF16, _lib.CXX_DTYPE_FP4_E2M1 if a.fmt == "nv" else _lib.CXX_DTYPE_MXFP4_E2M1,
                           _lib.CXX_DTYPE_BF16, 0)
sid = _lib.PETIT_SOLUTION_AUTO if a.solution == "auto" else int(a.solution, 16)
images = []
if a.sentinel:
    sid = {"mxfp8": _lib.PETIT_SOLUTION_AUTO_NATIVE_MXFP8, "mxfp6": _lib.PETIT_SOLUTION_AUTO_NATIVE_MXFP6, "mxfp4": _lib.PETIT_SOLUTION_AUTO_NATIVE_MXFP4}[a.sentinel]
    if a.fmt == "nv":
        nbytes = int(_lib.lib.petit_nvfp4_native_image_bytes(k, n))
        for b, sp in packed:
            img = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            assert _lib.lib.petit_nvfp4_native_image(C.c_void_p(img.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(sp.data_ptr()), k, n, None) == 0
            assert _lib.lib.petit_nvfp4_native_attach(C.c_void_p(b.data_ptr()), C.c_void_p(img.data_ptr())) == 0
            images.append(img)
        torch.cuda.synchronize()
    print("native class pick", hex(int(_lib.lib.petit_gemm_resolve_solution(C.byref(hints), m, n, k, C.c_uint64(sid), None, C.c_uint64(1 << 62)))))
ws = None
if a.native:
    _lib.lib.petit_enable_native_fp4(1)
    cnt = C.c_uint(0)
    _lib.lib.petit_gemm_get_solutions(C.byref(hints), m, n, k, None, C.byref(cnt))
    ids = (C.c_uint64 * cnt.value)()
    _lib.lib.petit_gemm_get_solutions(C.byref(hints), m, n, k, ids, C.byref(cnt))
    native = [i for i in ids if (i >> 48) & 0xF in (9, 13)]
    if a.solution == "auto":   # tile_m * n-tiles per wave: the largest tile
        sid = max(native, key=lambda i: (i & 0xFF) * ((i >> 52) & 0xF))
    print("native solution", hex(sid), _lib.describe_solution(sid))
need = int(_lib.lib.petit_gemm_workspace_bytes(C.byref(hints), m, n, k, C.c_uint64(sid)))
if need:
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    _lib.lib.petit_set_workspace(C.c_void_p(ws.data_ptr()), C.c_uint64(need))
fn = _lib.lib.petit_gemm_fp4_fp16_grid if a.fmt == "nv" else _lib.lib.petit_gemm_mxfp4_fp16_grid
torch.cuda.synchronize()
for i in range(a.iters):
    b, sp = packed[i % copies]
    rc = fn(C.c_void_p(c.data_ptr()), C.c_void_p(x.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(sp.data_ptr()),
            C.c_void_p(gs.data_ptr()), m, n, k, C.byref(hints), C.c_uint64(sid), None)
    assert rc == 0, rc
torch.cuda.synchronize()
print("done", c.float().abs().mean().item())
