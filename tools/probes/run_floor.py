"""Launch-floor and raw-streaming probes: what does ONE back-to-back kernel launch cost, and how
fast can 37.7 MB (the 8192^2 NVFP4 problem) be pulled from HBM by a kernel that does nothing else?"""
import ctypes as C
import json
from pathlib import Path

import torch

lib = C.CDLL(str(Path(__file__).resolve().parent / "libprobes.so"))
lib.run_probe_stream.argtypes = [C.c_int, C.c_int, C.c_uint, C.c_uint, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
dev = torch.device("cuda", 0)
sink = torch.zeros(1024, dtype=torch.int32, device=dev)
stream = torch.cuda.Stream(dev)


def timeit(fn, launches=200, reps=5):
    with torch.cuda.stream(stream):
        fn(0)
        stream.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            for i in range(launches):
                fn(i)
        g.replay()
        stream.synchronize()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            g.replay()
            e1.record(stream)
            stream.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / launches)
    return sorted(ts)[len(ts) // 2]


out = {"empty": {}, "stream": {}}
for grid, block in [(1, 64), (512, 512)]:
    us = timeit(lambda i: lib.run_probe_empty(grid, block, C.c_void_p(sink.data_ptr()),
                                              C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    out["empty"][f"{grid}x{block}"] = us
    print(f"empty {grid:5d} x {block:4d}: {us:6.2f} us", flush=True)

nbytes = 37781504 // 1024 * 1024
copies = 10
bufs = [torch.randint(-2 ** 31, 2 ** 31 - 1, (nbytes // 4,), dtype=torch.int32, device=dev) for _ in range(copies)]
for kind, per, grid, block in [(0, 16, 512, 256), (0, 2, 4096, 256), (0, 4, 2048, 256), (0, 8, 576, 512), (0, 8, 1152, 256),
                               (1, 8, 0, 512), (1, 8, 0, 256), (2, 8, 0, 512), (2, 8, 0, 256), (2, 8, 0, 1024),
                               (2, 4, 0, 512), (2, 16, 0, 256), (2, 16, 0, 512), (1, 8, 0, 512), (2, 8, 0, 512)]:
    if kind >= 1:  # grid must cover the buffer exactly: waves * per KiB
        waves = nbytes // (per * 1024)
        grid = (waves * 64 + block - 1) // block
    us = timeit(lambda i: lib.run_probe_stream(kind, per, grid, block, bufs[i % copies].data_ptr(), nbytes,
                                               sink.data_ptr(), torch.cuda.current_stream().cuda_stream))
    out["stream"][f"kind{kind}_per{per}_{grid}x{block}"] = us
    print(f"stream kind {kind} per {per:2d} grid {grid:5d} x {block:4d}: {us:6.2f} us  {nbytes / us / 1e3:7.0f} GB/s", flush=True)
print(json.dumps(out))
