"""tools/probes/run_ksplit_combine.py [out.json] -- see ksplit_combine.hip.  us per launch (HIP-graph replay, weights rotated over 1.28 GB, median of
7) for `o` (8192 x 8192) and `down` (8192 x 28672) at M = 8 / 16, bf16 x NVFP4: unsplit, split + second launch, split combined in the launch;
every split output is checked against the unsplit kernel's (same summation inside a slice, slices added in order: equal up to f32 rounding)."""
import ctypes as C
import json
import subprocess
import sys
from pathlib import Path

import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT / "petit-kernel_amd"))
sys.path.insert(0, str(ROOT / "tools"))
import benchlib as BL

so = HERE / "libksplitcombine.so"
src = HERE / "ksplit_combine.hip"
if not so.exists() or so.stat().st_mtime < max(src.stat().st_mtime, (ROOT / "petit-kernel_amd/csrc/gemm_stream.hpp").stat().st_mtime):
    subprocess.run(["hipcc", "-O3", "-std=c++20", "-shared", "-fPIC", "--offload-arch=gfx950", "-mllvm", "-amdgpu-kernarg-preload-count=16",
                    f"-I{ROOT / 'petit-kernel_amd/csrc'}", str(src), "-o", str(so)], check=True)
lib = C.CDLL(str(so))
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(dev)
out = []
SHAPES = {0: "16x32 (NT 2)", 1: "16x64 (NT 4)", 2: "16x128 (WN 2 x NT 4)"}
for name, (n, k) in (("o", (8192, 8192)), ("down", (8192, 28672))):
    w = BL.Weights("nv", n, k, 1280, dev)
    for m in (16, 8):
        g = BL.Gemm(w, m, torch.bfloat16, dev)
        ws = torch.empty(8 * m * n, dtype=torch.float32, device=dev)
        tickets = torch.zeros(4096, dtype=torch.int32, device=dev)

        def launcher(shape, variant, splitk, c):
            def launch(i):
                b, sp = w[i]
                rc = lib.ksc_launch(shape, variant, C.c_void_p(b.data_ptr()), C.c_void_p(sp.data_ptr()), C.c_void_p(g.a.data_ptr()), C.c_void_p(c.data_ptr()),
                                    C.c_void_p(g.gs.data_ptr()), C.c_void_p(ws.data_ptr()), C.c_void_p(tickets.data_ptr()), m, n, k, splitk,
                                    C.c_void_p(torch.cuda.current_stream().cuda_stream))
                assert rc == 0, rc
            return launch
        # the library's own default, for reference
        dflt = g.time(0xFFFFFFFFFFFFFFFF, stream)["us"]
        out.append({"shape": name, "M": m, "kernel": "library default (solution_id = -1)", "us": round(dflt, 2)})
        print(json.dumps(out[-1]), flush=True)
        for shape in SHAPES:
            ref = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
            with torch.cuda.stream(stream):
                launcher(shape, 0, 1, ref)(0)
            stream.synchronize()
            for variant, splitk in ((0, 1), (1, 2), (2, 2), (1, 4), (2, 4)):
                c = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
                with torch.cuda.stream(stream):
                    for rep in range(3):   # (repeat: the tickets must come back to zero by themselves)
                        launcher(shape, variant, splitk, c)(0)
                stream.synchronize()
                err = (c.float() - ref.float()).abs().max().item() / max(ref.float().abs().max().item(), 1e-9)
                assert err < 1e-2, (name, m, shape, variant, splitk, err)
                assert int(tickets.abs().sum().item()) == 0
                us = BL.median(BL.time_graph(launcher(shape, variant, splitk, c), 100, 7, stream))
                rec = {"shape": name, "M": m, "wg_tile": SHAPES[shape], "variant": ["unsplit", "split + second launch", "split, combined in the launch"][variant],
                       "splitk": splitk, "us": round(us, 2), "max_rel_diff_vs_unsplit": err}
                out.append(rec)
                print(json.dumps(rec), flush=True)
    del w
    torch.cuda.empty_cache()
dst = Path(sys.argv[1]) if len(sys.argv) > 1 else ROOT / "gpurun_out" / "ksplit_combine.json"
dst.parent.mkdir(parents=True, exist_ok=True)
dst.write_text(json.dumps(out, indent=1))
