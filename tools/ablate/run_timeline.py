"""Per-wave timeline of the M = 1 bench kernel (ablate.hip variant 0, ABL = 16): s_memrealtime (100 MHz) stamps at
kernel entry, after the wave's first tile, after its last tile and at exit; printed relative to the earliest entry."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import torch

lib = C.CDLL(str(Path(__file__).resolve().parent / "libablate.so"))
dev = torch.device("cuda", 0)
n = k = 8192
m = int(sys.argv[1]) if len(sys.argv) > 1 else 1
variant = int(sys.argv[2]) if len(sys.argv) > 2 else (0 if m == 1 else 2)
copies = 37
gen = torch.Generator(device=dev).manual_seed(1)
packed = [(torch.randint(-2 ** 31, 2 ** 31 - 1, (n // 16, 2 * k), generator=gen, dtype=torch.int32, device=dev),
           (torch.rand((n, k // 16), generator=gen, device=dev) * 3.5 + 0.25).to(torch.float8_e4m3fn)) for _ in range(copies)]
gs = torch.ones(1, device=dev)
a = torch.randn((m, k), device=dev).bfloat16()
c = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
stamps = torch.zeros(8192 * 4, dtype=torch.int64, device=dev)
stream = torch.cuda.Stream(dev)


def launch(i, abl):
    b, sp = packed[i % copies]
    rc = lib.ablate_launch(variant, abl, C.c_void_p(c.data_ptr()), C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()),
                           C.c_void_p(sp.data_ptr()), C.c_void_p(gs.data_ptr()), m, n, k,
                           C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_void_p(stamps.data_ptr()))
    assert rc == 0


with torch.cuda.stream(stream):
    launch(0, 16)
    stream.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        for i in range(200):
            launch(i, 16)
    for _ in range(12):
        g.replay()
    stream.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    g.replay()
    e1.record(stream)
    stream.synchronize()
    print(f"M={m} variant {variant}: {e0.elapsed_time(e1) * 1e3 / 200:.2f} us per launch (instrumented)")
s = stamps.cpu().numpy().reshape(-1, 4)
s = s[s[:, 0] != 0].astype(np.float64)
t0 = s[:, 0].min()
rel = (s - t0) * 10.0 / 1000.0          # 100 MHz ticks -> us
names = ["wave entry", "first tile consumed", "last tile consumed", "wave exit"]
print(f"{len(s)} waves of the LAST launch; microseconds after the earliest wave entry")
for i, nm in enumerate(names):
    col = rel[:, i]
    print(f"  {nm:20s} min {col.min():5.2f}  p10 {np.percentile(col, 10):5.2f}  median {np.median(col):5.2f}  p90 {np.percentile(col, 90):5.2f}  max {col.max():5.2f}")
d = rel[:, 3] - rel[:, 2]
print(f"  exit - last tile     median {np.median(d):5.2f}  p90 {np.percentile(d, 90):5.2f}  max {d.max():5.2f}")
d = rel[:, 2] - rel[:, 1]
print(f"  last - first tile    median {np.median(d):5.2f}  p90 {np.percentile(d, 90):5.2f}  max {d.max():5.2f}")
