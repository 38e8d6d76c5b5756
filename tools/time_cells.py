#!/usr/bin/env python3
"""tools/time_cells.py -- time a filtered subset of bench.py's cell plan (same method: tools/benchlib.py) and print one JSON line per cell.

    python tools/time_cells.py --w nv --mode native --m 512,1024 [--shape o,down] [--out gpurun_out/x.jsonl] [--sid 0x...]

Used between full bench runs while a kernel family is being worked on."""
import argparse
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
sys.path.insert(0, str(ROOT / "petit-kernel_amd"))
import benchlib as BL  # noqa: E402
from petit_kernel import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--w", default="nv")
    ap.add_argument("--a", default="bf16")
    ap.add_argument("--mode", default="native", help="substring of the cell's mode (auto, native, native_mxfp8, ...)")
    ap.add_argument("--m", default="")
    ap.add_argument("--shape", default="")
    ap.add_argument("--out", default="")
    ap.add_argument("--all-kernels", action="store_true", help="time every enumerated kernel of the class instead of the default pick")
    args = ap.parse_args()
    ms = {int(x) for x in args.m.split(",") if x}
    shapes = {x for x in args.shape.split(",") if x}
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream()
    mode_sid = {"auto": _lib.PETIT_SOLUTION_AUTO, "native_mxfp8": _lib.PETIT_SOLUTION_AUTO_NATIVE_MXFP8,
                "native_mxfp6": _lib.PETIT_SOLUTION_AUTO_NATIVE_MXFP6, "native_mxfp4": _lib.PETIT_SOLUTION_AUTO_NATIVE_MXFP4}
    sink = open(args.out, "a") if args.out else None
    weights = {}
    for cell in BL.bench_cell_plan():
        shape, m, a, w, mode = cell["shape"], cell["M"], cell["a"], cell["w"], cell["mode"]
        if shape not in BL.LLAMA70B or w != args.w or a != args.a or args.mode not in mode or (ms and m not in ms) or (shapes and shape not in shapes):
            continue
        n, k = BL.LLAMA70B[shape]
        if (shape, w) not in weights:
            weights.clear()
            torch.cuda.empty_cache()
            weights[(shape, w)] = BL.Weights(w, n, k, 1280, dev)
        g = BL.Gemm(weights[(shape, w)], m, torch.bfloat16 if a == "bf16" else torch.float16, dev)
        sids = [mode_sid[mode]]
        if args.all_kernels and mode != "auto":
            _lib.lib.petit_enable_native_fp4(1)
            code = {"native_mxfp8": 2, "native_mxfp6": 4, "native_mxfp4": 6}[mode]
            sids = [s for s in g.solutions() if (s >> 48) & 0xF == 13 and (s >> 32) & 7 == code]
            g.w.attach_native()
        for sid in sids:
            try:
                picked = g.resolve(sid)
                if sid in mode_sid.values() and w == "nv" and mode != "auto":
                    g.w.attach_native()
                r = g.time(sid, stream, reps=5)
                out = {"shape": shape, "M": m, "dt": f"{a}x{w} {mode}", "us": round(r["us"], 2), "us_min": round(r["us_min"], 2), "TF": round(r["tflops"], 1),
                       "sid": f"{picked:x}", "kernel": _lib.describe_solution(picked)}
            except Exception as exc:  # noqa: BLE001
                out = {"shape": shape, "M": m, "dt": f"{a}x{w} {mode}", "error": str(exc), "sid": f"{sid:x}"}
            line = json.dumps(out)
            print(line, flush=True)
            if sink:
                sink.write(line + "\n")
                sink.flush()


if __name__ == "__main__":
    main()
