"""Time the ablation variants of the streaming kernel (see ablate.hip)."""
import ctypes as C
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent.parent
lib = C.CDLL(str(Path(__file__).resolve().parent / "libablate.so"))
dev = torch.device("cuda", 0)
n = k = 8192
ms = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,16").split(",")]
copies = 37
gen = torch.Generator(device=dev).manual_seed(1)
packed = [(torch.randint(-2 ** 31, 2 ** 31 - 1, (n // 16, 2 * k), generator=gen, dtype=torch.int32, device=dev),
           (torch.rand((n, k // 16), generator=gen, device=dev) * 3.5 + 0.25).to(torch.float8_e4m3fn)) for _ in range(copies)]
gs = torch.ones(1, device=dev)
stream = torch.cuda.Stream(dev)
names = {0: "full", 1: "no A loads", 2: "no unpack", 3: "no A, no unpack", 4: "no mfma", 5: "no A, no mfma",
         7: "loads of W/scales only", 8: "empty kernel", 32: "full, private copy of A per workgroup"}
out = {}
for m in ms:
    a = torch.randn((64 * m, k), device=dev).bfloat16()   # (abl 32 reads copy blockIdx % 64; the others the first m rows)
    c = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
    variants = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else ((0, 1) if m == 1 else (2, 3))
    for variant in variants:
        for abl in (0, 1, 2, 3, 4, 5, 7, 8, 32):
            def launch(i):
                b, sp = packed[i % copies]
                rc = lib.ablate_launch(variant, abl, C.c_void_p(c.data_ptr()), C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()),
                                       C.c_void_p(sp.data_ptr()), C.c_void_p(gs.data_ptr()), m, n, k,
                                       C.c_void_p(torch.cuda.current_stream().cuda_stream))
                assert rc == 0
            with torch.cuda.stream(stream):
                launch(0)
                stream.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=stream):
                    for i in range(200):
                        launch(i)
                for _ in range(12):
                    g.replay()
                stream.synchronize()
                ts = []
                for _ in range(5):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(stream)
                    g.replay()
                    e1.record(stream)
                    stream.synchronize()
                    ts.append(e0.elapsed_time(e1) * 1e3 / 200)
            us = sorted(ts)[2]
            out[f"m{m}_v{variant}_abl{abl}"] = us
            print(f"M={m:<2d} variant {variant} abl {abl} ({names[abl]:24s}) {us:7.2f} us", flush=True)
print(json.dumps(out))
