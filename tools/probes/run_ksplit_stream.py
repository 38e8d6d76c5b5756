"""tools/probes/run_ksplit_stream.py -- see ksplit_stream.hip.  us per launch for the `o` (K = 8192) and `down` (K = 28672) walks at M = 512, N = 8192
(256 workgroups of one 128 x 128 tile), with and without the MFMAs, by register ring depth."""
import ctypes as C
import json
import subprocess
import sys
from pathlib import Path

import torch

HERE = Path(__file__).resolve().parent
so = HERE / "libksplit.so"
if not so.exists() or so.stat().st_mtime < (HERE / "ksplit_stream.hip").stat().st_mtime:
    subprocess.run(["hipcc", "-O3", "-shared", "-fPIC", "--offload-arch=gfx950", str(HERE / "ksplit_stream.hip"), "-o", str(so)], check=True)
lib = C.CDLL(str(so))
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(dev)
sink = torch.zeros(4, dtype=torch.float32, device=dev)
out = []
with torch.cuda.stream(stream):
    for k, nsets in ((8192, 1), (8192, 12), (28672, 12)):
        ktiles = k // 128
        # rotate over several weight sets so that the panels come from HBM / Infinity Cache as in the benchmark
        w = torch.randint(0, 2 ** 31 - 1, (nsets, 64 * ktiles * 2048), dtype=torch.int32, device=dev)
        a = torch.randint(0, 2 ** 31 - 1, (4 * ktiles * 2048,), dtype=torch.int32, device=dev)
        for r, mfma, stag in [(r, m, s) for r in (2, 3) for s in (0, 1) for m in (1, 0)]:
            if True:
                def launch(i):
                    return lib.ksplit_launch(r, mfma, stag, C.c_void_p(w[i % nsets].data_ptr()), C.c_void_p(a.data_ptr()), ktiles, C.c_void_p(sink.data_ptr()),
                                             C.c_void_p(stream.cuda_stream))
                for i in range(3):
                    assert launch(i) == 0
                stream.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                n = 24
                for i in range(n):
                    launch(i)
                e1.record(stream)
                stream.synchronize()
                us = e0.elapsed_time(e1) / n * 1e3
                flops = 2.0 * 512 * 8192 * k
                rec = {"k": k, "weight_sets": nsets, "ring_tiles": r, "mfma": mfma, "interleaved_refills": stag, "us": round(us, 2), "PFLOPs": round(flops / us / 1e9, 2) if mfma else None,
                       "GBps_per_cu": round((ktiles * 16384) / us / 1e3, 1)}
                out.append(rec)
                print(json.dumps(rec), flush=True)
dst = Path(sys.argv[1]) if len(sys.argv) > 1 else HERE.parent.parent / "gpurun_out" / "ksplit_stream.json"
dst.parent.mkdir(parents=True, exist_ok=True)
dst.write_text(json.dumps(out, indent=1))
