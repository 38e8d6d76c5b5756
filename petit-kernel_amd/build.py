"""Build libpetit_amd.so for gfx950 with hipcc (cross-compiles without a GPU).

    python petit-kernel_amd/build.py [--force] [-j N]

One object per translation unit, compiled in parallel, linked into
petit-kernel_amd/lib/libpetit_amd.so.  No cmake, no torch: the library's only
dependency is the HIP runtime.
"""
from __future__ import annotations

import argparse
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent
CSRC = ROOT / "csrc"
OBJ = ROOT / "build"
LIB = ROOT / "lib" / "libpetit_amd.so"
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++20", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function", "-Wno-unused-variable",
         "-fno-gpu-rdc", "-DNDEBUG"]
# kernel arguments preloaded into SGPRs by the command processor (gemm_stream.hpp: scalar arguments, most urgent first);
# $PETIT_AMD_NO_KERNARG_PRELOAD=1 builds without (A/B measurements)
if not os.environ.get("PETIT_AMD_NO_KERNARG_PRELOAD"):
    FLAGS += ["-mllvm", "-amdgpu-kernarg-preload-count=16"]


def sources():
    # the slowest translation units first (by the size of their previous object file), so that the tail of a parallel build is short
    def weight(src: Path) -> int:
        obj = OBJ / (src.stem + ".o")
        return -(obj.stat().st_size if obj.exists() else 1 << 30)
    return sorted(CSRC.glob("*.hip"), key=lambda s: (weight(s), s.name))


def deps_mtime() -> float:
    files = list(CSRC.glob("*.h")) + list(CSRC.glob("*.hpp")) + list(CSRC.glob("*.inc"))
    files.append(ROOT.parent / "include" / "petit_amd.h")
    return max(f.stat().st_mtime for f in files)


def compile_one(src: Path, force: bool, hdr_mtime: float) -> Path:
    obj = OBJ / (src.stem + ".o")
    if not force and obj.exists() and obj.stat().st_mtime > max(src.stat().st_mtime, hdr_mtime):
        return obj
    cmd = ["hipcc", *FLAGS, "-c", str(src), "-o", str(obj)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError(f"hipcc failed on {src.name}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build(force: bool = False, jobs: int | None = None) -> Path:
    OBJ.mkdir(exist_ok=True)
    LIB.parent.mkdir(exist_ok=True)
    hdr = deps_mtime()
    srcs = sources()
    jobs = jobs or min(len(srcs), os.cpu_count() or 4)
    for stale in OBJ.glob("*.o"):  # objects of translation units that no longer exist
        if not (CSRC / (stale.stem + ".hip")).exists():
            stale.unlink()
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(lambda s: compile_one(s, force, hdr), srcs))
    if force or not LIB.exists() or any(o.stat().st_mtime > LIB.stat().st_mtime for o in objs):
        cmd = ["hipcc", "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", str(LIB), *map(str, objs)]
        subprocess.run(cmd, check=True)
    return LIB


TORCH_LIB = ROOT / "lib" / "libpetit_torch.so"


def build_torch_binding(force: bool = False) -> Path | None:
    """csrc/torch_binding.cpp -> lib/libpetit_torch.so with the host compiler (no device code); needs torch's headers.
    Returns None (with a note on stderr) when torch is not importable: the ctypes layer does not need it."""
    src = CSRC / "torch_binding.cpp"
    hdr = ROOT.parent / "include" / "petit_amd.h"
    if not force and TORCH_LIB.exists() and TORCH_LIB.stat().st_mtime > max(src.stat().st_mtime, hdr.stat().st_mtime, LIB.stat().st_mtime):
        return TORCH_LIB
    try:
        import torch
        from torch.utils import cpp_extension as ce
    except Exception as exc:  # noqa: BLE001
        sys.stderr.write(f"[build] torch binding skipped: {exc}\n")
        return None
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", "-w"]
    cmd += [f"-I{p}" for p in ce.include_paths()] + ["-I/opt/rocm/include", str(src), "-o", str(TORCH_LIB),
                                                      f"-L{LIB.parent}", "-lpetit_amd"]
    cmd += [f"-L{p}" for p in ce.library_paths()] + ["-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip", "-ltorch_hip",
                                                      "-Wl,-rpath,$ORIGIN"]
    subprocess.run(cmd, check=True)
    return TORCH_LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("-j", type=int, default=None)
    a = ap.parse_args()
    print(build(a.force, a.j))
    print(build_torch_binding(a.force))
