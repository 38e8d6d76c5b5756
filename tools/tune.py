#!/usr/bin/env python3
"""tools/tune.py -- autotune sweep + roofline table on a real MI355X.

Replaces the reference's `bench_matmul -algo tune` / tools/benchmarks/matmul.py
(tools/benchmarks/matmul/main.cc:269-325, matmul.py:92-165): for every (shape, M) it times
every enumerated solution (optionally with split-K variants), prints the ranking, and writes
  * a JSON report (all timings, achieved GB/s, TFLOPS, fraction of the HBM roofline),
  * the arch-table rows for petit-kernel_amd/csrc/tuned_gfx950.inc and the
    $PETIT_AMD_TUNE_FILE text format.

Method: launches are replayed from a HIP graph (so the host is out of
the loop), rotate over enough distinct (W, scales) copies that no launch re-reads weights
resident in the 256 MB Infinity Cache (the reference reuses ONE buffer,
matmul_petit.cc:116-132), and are timed with events on the launch stream.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "petit-kernel_amd"))
sys.path.insert(0, str(ROOT))

import torch

import petit_kernel
from petit_kernel import _lib

LLAMA70B = {"qkv": (10240, 8192), "o": (8192, 8192), "gate_up": (57344, 8192), "down": (8192, 28672),
            "sq4096": (4096, 4096), "sq8192": (8192, 8192)}
HBM_PEAK = 8000.0


def alg_bytes(m, n, k, g):
    return n * k // 2 + n * k // g + 2 * m * k + 2 * m * n + 4


def time_graph(fn_launch, launches: int, reps: int, stream) -> list:
    """Capture `launches` calls, replay `reps` times; returns us per launch for each replay."""
    with torch.cuda.stream(stream):
        fn_launch(0)
        stream.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            for i in range(launches):
                fn_launch(i)
        # warm-up: replay for >= 20 ms so clocks are in steady state (DVFS ramp, see bench.py)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.02:
            g.replay()
            stream.synchronize()
        out = []
        for _ in range(reps):
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            g.replay()
            e1.record(stream)
            stream.synchronize()
            out.append(e0.elapsed_time(e1) * 1e3 / launches)
        del g
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="sq8192,sq4096,qkv,gate_up,down")
    ap.add_argument("--ms", default="1,4,8,16")
    ap.add_argument("--fmt", default="nv", choices=["nv", "mx"])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16"])
    ap.add_argument("--splitk", default="1", help="comma list of split-K factors to try on top of each shape")
    ap.add_argument("--launches", type=int, default=0, help="launches per graph (0 = auto)")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--out", default=str(ROOT / "gpurun_out" / "tune.json"))
    ap.add_argument("--only-default", action="store_true", help="time only the solution_id=-1 choice")
    ap.add_argument("--compare-dense", action="store_true",
                    help="also time torch.matmul (hipBLASLt/rocBLAS) on a dense 16-bit weight of the same shape")
    ap.add_argument("--rotate-mb", type=int, default=1280, help="rotate over at least this many MB of distinct weights")
    ap.add_argument("--native", action="store_true",
                    help="also enumerate the opt-in native-FP4 kernels (MXFP4 only; activations quantised to MXFP8)")
    args = ap.parse_args()

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    group = 16 if args.fmt == "nv" else 32
    a_type = _lib.CXX_DTYPE_BF16 if args.dtype == "bf16" else _lib.CXX_DTYPE_FP16
    b_type = _lib.CXX_DTYPE_FP4_E2M1 if args.fmt == "nv" else _lib.CXX_DTYPE_MXFP4_E2M1
    stream = torch.cuda.Stream(dev)
    ws = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device=dev)  # 256 MB of split-K scratch
    _lib.lib.petit_set_workspace(C.c_void_p(ws.data_ptr()), C.c_uint64(ws.numel() * 4))
    if args.native:
        _lib.lib.petit_enable_native_fp4(1)
    splitks = [int(x) for x in args.splitk.split(",")]
    report = {"device": torch.cuda.get_device_properties(0).gcnArchName, "fmt": args.fmt, "dtype": args.dtype,
              "hbm_peak_gbs": HBM_PEAK, "results": []}
    rows = []
    t_start = time.time()
    for name in args.shapes.split(","):
        n, k = LLAMA70B[name] if name in LLAMA70B else tuple(int(x) for x in name.split("x"))
        wbytes = n * k // 2 + n * k // group
        copies = max(2, (args.rotate_mb << 20) // wbytes + 2)
        gen = torch.Generator(device=dev).manual_seed(1234)
        packed = []
        for _ in range(copies):
            b = torch.randint(-2 ** 31, 2 ** 31 - 1, (n // 16, 2 * k), generator=gen, dtype=torch.int32, device=dev)
            if args.fmt == "nv":
                sp = (torch.rand((n, k // 16), generator=gen, device=dev) * 3.5 + 0.25).to(torch.float8_e4m3fn)
            else:
                sp = torch.randint(119, 136, (n // 32, k), generator=gen, dtype=torch.uint8, device=dev)
            packed.append((b, sp))
        gs = torch.tensor([1.0], dtype=torch.float32, device=dev)
        for m in [int(x) for x in args.ms.split(",")]:
            a = torch.randn((m, k), generator=gen, device=dev, dtype=torch.float32).to(dtype)
            c = torch.empty((m, n), dtype=dtype, device=dev)
            hints = _lib.SolutionHints(a_type, b_type, a_type, 0)
            default_sid = _lib.lib.petit_gemm_default_solution(C.byref(hints), m, n, k)
            if args.only_default:
                cands = [default_sid]
            else:
                h = petit_kernel.PetitSolutionHints()
                h.a_type = dtype
                h.c_type = dtype
                h.b_type = b_type
                base = petit_kernel.ops.get_fp4_solutions(h, m, n, k)
                cands = []
                for sid in base:
                    for sk in splitks:
                        cands.append((sid & ~(0xF << 60)) | (sk << 60))
            fn = _lib.lib.petit_gemm_fp4_fp16_grid if args.fmt == "nv" else _lib.lib.petit_gemm_mxfp4_fp16_grid
            nbytes = alg_bytes(m, n, k, group)
            ideal_us = nbytes / (HBM_PEAK * 1e3)
            launches = args.launches or int(max(20, min(400, 3000.0 / max(ideal_us, 1.0))))
            results = []
            for sid in cands:
                def launch(i, sid=sid):
                    b, sp = packed[i % copies]
                    rc = fn(C.c_void_p(c.data_ptr()), C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()),
                            C.c_void_p(sp.data_ptr()), C.c_void_p(gs.data_ptr()), m, n, k, C.byref(hints),
                            C.c_uint64(sid), C.c_void_p(torch.cuda.current_stream().cuda_stream))
                    if rc != 0:
                        raise RuntimeError(f"rc={rc}")
                try:
                    us = time_graph(launch, launches, args.reps, stream)
                except Exception as exc:  # noqa: BLE001
                    results.append({"solution": f"0x{sid:x}", "error": str(exc)})
                    continue
                us_med = sorted(us)[len(us) // 2]
                results.append({"solution": f"0x{sid:x}", "desc": _lib.describe_solution(sid), "us_median": us_med,
                                "us_min": min(us), "gbs": nbytes / us_med / 1e3, "frac_hbm": nbytes / us_med / 1e3 / HBM_PEAK,
                                "tflops": 2.0 * m * n * k / us_med / 1e6, "is_default": sid == default_sid})
            dense = None
            if args.compare_dense:
                # the reference's comparator (tools/benchmarks/matmul/rocm/matmul_hipblaslt.cc): a plain
                # 16-bit GEMM C = A . Wd^T through the vendor library, weights rotated the same way
                dcopies = max(2, min(copies, (args.rotate_mb << 20) // (n * k * 2) + 2))
                wd = [torch.randn((n, k), device=dev, dtype=torch.float32).to(dtype) for _ in range(dcopies)]
                def dlaunch(i):
                    torch.matmul(a, wd[i % dcopies].t(), out=c)
                us = time_graph(dlaunch, max(10, launches // 4), args.reps, stream)
                us_med = sorted(us)[len(us) // 2]
                dense = {"us_median": us_med, "tflops": 2.0 * m * n * k / us_med / 1e6,
                         "gbs_dense_weights": (2.0 * n * k + 2 * m * k + 2 * m * n) / us_med / 1e3}
                del wd
            ok = sorted([r for r in results if "us_median" in r], key=lambda r: r["us_median"])
            entry = {"shape": name, "n": n, "k": k, "m": m, "bytes": nbytes, "ideal_us_at_8TBs": ideal_us,
                     "copies": copies, "launches": launches, "dense_16bit_gemm": dense,
                     "results": ok + [r for r in results if "error" in r]}
            report["results"].append(entry)
            if ok:
                best = ok[0]
                dflt = next((r for r in ok if r["is_default"]), None)
                print(f"{name:8s} M={m:<3d} best {best['us_median']:8.2f} us {best['gbs']:7.0f} GB/s "
                      f"({100 * best['frac_hbm']:.1f}% of 8 TB/s) {best['desc']}"
                      + (f" | default {dflt['us_median']:.2f} us" if dflt else "")
                      + (f" | {best['tflops']:.0f} TF vs dense 16-bit GEMM {dense['us_median']:.1f} us {dense['tflops']:.0f} TF" if dense else ""),
                      flush=True)
                rows.append((a_type, b_type, n, k, m, int(best["solution"], 16)))
        del packed
        torch.cuda.empty_cache()
    report["elapsed_s"] = time.time() - t_start
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    Path(args.out).write_text(json.dumps(report, indent=1))
    tune_txt = Path(args.out).with_suffix(".tune.txt")
    with open(tune_txt, "w") as f:
        f.write("# a_type b_type n k m_lo m_hi solution   (tools/tune.py; $PETIT_AMD_TUNE_FILE format)\n")
        for (at, bt, n, k, m, sid) in rows:
            f.write(f"{at} {bt} {n} {k} {m} {m} {sid:x}\n")
    print(f"wrote {args.out} and {tune_txt}")


if __name__ == "__main__":
    main()
