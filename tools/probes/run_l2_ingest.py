"""tools/probes/run_l2_ingest.py -- aggregate rate at which the CUs pull SHARED data (L2 / Infinity Cache / HBM resident windows) with the kernels'
access shape (1 KiB per wave-load), by workgroups per CU and loads in flight per wave.  Build: hipcc -O3 -shared -fPIC --offload-arch=gfx950
tools/probes/l2_ingest.hip -o tools/probes/libl2ingest.so"""
import ctypes as C
import json
import subprocess
import sys
from pathlib import Path

import torch

HERE = Path(__file__).resolve().parent
so = HERE / "libl2ingest.so"
if not so.exists() or so.stat().st_mtime < (HERE / "l2_ingest.hip").stat().st_mtime:
    subprocess.run(["hipcc", "-O3", "-shared", "-fPIC", "--offload-arch=gfx950", str(HERE / "l2_ingest.hip"), "-o", str(so)], check=True)
lib = C.CDLL(str(so))
dev = torch.device("cuda", 0)
buf = torch.randint(0, 2 ** 31 - 1, (1 << 28,), dtype=torch.int32, device=dev)      # 1 GiB
sink = torch.zeros(4, dtype=torch.int32, device=dev)
stream = torch.cuda.Stream(dev)
out = []
with torch.cuda.stream(stream):
    for window_mb in (1, 16, 128, 1024):
        for wg_per_cu in (1, 2, 4, 8):
            for u in (2, 4, 8, 16):
                blocks, iters = 256 * wg_per_cu, 4096 // wg_per_cu
                args = (u, C.c_void_p(buf.data_ptr()), window_mb << 20, iters, blocks, C.c_void_p(sink.data_ptr()), C.c_void_p(stream.cuda_stream))
                for _ in range(2):
                    assert lib.ingest_launch(*args) == 0
                stream.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(5):
                    lib.ingest_launch(*args)
                e1.record(stream)
                stream.synchronize()
                ms = e0.elapsed_time(e1) / 5
                nbytes = blocks * 4 * iters * 1024
                rec = {"window_MB": window_mb, "wg_per_cu": wg_per_cu, "waves_per_cu": 4 * wg_per_cu, "loads_in_flight_per_wave": u,
                       "KiB_in_flight_per_cu": 4 * wg_per_cu * u, "TBps": round(nbytes / ms / 1e9, 2), "GBps_per_cu": round(nbytes / ms / 1e6 / 256, 1)}
                out.append(rec)
                print(json.dumps(rec), flush=True)
dst = Path(sys.argv[1]) if len(sys.argv) > 1 else HERE.parent.parent / "gpurun_out" / "l2_ingest.json"
dst.parent.mkdir(parents=True, exist_ok=True)
dst.write_text(json.dumps(out, indent=1))
