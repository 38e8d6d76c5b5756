#!/usr/bin/env python3
"""tools/tune.py -- autotune sweep + roofline table on a real MI355X.

Replaces the reference's `bench_matmul -algo tune` / tools/benchmarks/matmul.py
(tools/benchmarks/matmul/main.cc:269-325, matmul.py:92-165): for every (shape, M) it times
every enumerated solution (optionally with split-K variants), prints the ranking, and writes
  * a compact CSV (one row per timed candidate) and a small JSON summary (best / default per cell),
  * the arch-table rows for petit-kernel_amd/csrc/tuned_gfx950.inc in the
    $PETIT_AMD_TUNE_FILE text format.
Every candidate's OUTPUT is checked before it is timed (against the direct-path kernel of the same family, full
matrix, the library's 1e-2 bound): a kernel that miscomputes at this shape is dropped and reported, never ranked.
Native-FP4 kernels (--native) are timed and reported but never written to the arch table (own accuracy class).

Method: tools/benchlib.py (HIP-graph replay, rotating weights, warm-up, median).
"""
from __future__ import annotations

import argparse
import csv
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "petit-kernel_amd"))
sys.path.insert(0, str(ROOT / "tools"))
sys.path.insert(0, str(ROOT))

import torch

import benchlib as BL
from petit_kernel import _lib


def is_native(sid: int) -> bool:
    return (sid >> 48) & 0xF in (9, 13)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="sq8192,sq4096,qkv,gate_up,down")
    ap.add_argument("--ms", default="1,4,8,16")
    ap.add_argument("--fmt", default="nv", choices=["nv", "mx"])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16"])
    ap.add_argument("--splitk", default="1", help="comma list of split-K factors to try on top of each shape")
    ap.add_argument("--splitk-kinds", default="all", choices=["all", "tiled", "stream"],
                    help="which kernel kinds get the split-K variants (> 1)")
    ap.add_argument("--launches", type=int, default=0, help="launches per graph (0 = auto)")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--out", default=str(ROOT / "gpurun_out" / "tune.json"))
    ap.add_argument("--only-default", action="store_true", help="time only the solution_id=-1 choice")
    ap.add_argument("--kinds", default="", help="comma list of kernel-kind codes (bits 48-51 of the id) to keep, e.g. 8,9")
    ap.add_argument("--compare-dense", action="store_true",
                    help="also time hipBLASLt (explicitly; tools/comparators/hipblaslt_gemm.cc) on a dense 16-bit weight")
    ap.add_argument("--rotate-mb", type=int, default=1280, help="rotate over at least this many MB of distinct weights")
    ap.add_argument("--native", action="store_true",
                    help="also enumerate the opt-in native-FP4 kernels (MXFP4 only; activations quantised on the fly)")
    ap.add_argument("--no-check", action="store_true", help="skip the per-candidate output check")
    args = ap.parse_args()

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    stream = torch.cuda.Stream(dev)
    if args.native:
        _lib.lib.petit_enable_native_fp4(1)
    splitks = [int(x) for x in args.splitk.split(",")]
    kinds = {int(x) for x in args.kinds.split(",")} if args.kinds else None
    summary = {"device": torch.cuda.get_device_properties(0).gcnArchName, "fmt": args.fmt, "dtype": args.dtype,
               "hbm_peak_gbs": BL.HBM_PEAK_GBS, "cells": []}
    out = Path(args.out)
    out.parent.mkdir(parents=True, exist_ok=True)
    csv_f = open(out.with_suffix(".csv"), "w", newline="")
    cw = csv.writer(csv_f)
    cw.writerow(["dtype", "fmt", "shape", "n", "k", "m", "solution", "desc", "us_median", "us_min", "gbs", "frac_hbm", "tflops",
                 "is_default", "checked"])
    rows = []
    t_start = time.time()
    for name in args.shapes.split(","):
        n, k = BL.LLAMA70B[name] if name in BL.LLAMA70B else tuple(int(x) for x in name.split("x"))
        w = BL.Weights(args.fmt, n, k, args.rotate_mb, dev)
        for m in [int(x) for x in args.ms.split(",")]:
            g = BL.Gemm(w, m, dtype, dev)
            default_sid = g.default_solution()
            if args.only_default:
                cands = [default_sid]
            else:
                cands = []
                for sid in g.solutions():
                    kind = (sid >> 48) & 0xF
                    if kinds is not None and kind not in kinds:
                        continue
                    for sk in splitks:
                        if sk > 1 and ((args.splitk_kinds == "tiled" and kind not in (8, 9)) or
                                       (args.splitk_kinds == "stream" and kind in (8, 9))):
                            continue
                        cands.append((sid & ~(0xF << 60)) | (sk << 60))
                if default_sid not in cands:
                    cands.append(default_sid)
            # reference output for the check: the direct-path streaming kernel (am = 0), else the first exact candidate
            c_ref = None
            native_refs = {}     # activation format (mfma nibble 2 / 6) -> output of the first native kernel of that class
            if not args.no_check:
                exact = [s for s in g.solutions() if not is_native(s)]
                ref_sid = next((s for s in exact if (s >> 48) & 0xF == 0), exact[0])
                g.launcher(ref_sid)(0)
                torch.cuda.synchronize()
                c_ref = g.c.float().clone()
            results, dropped = [], []
            for sid in cands:
                try:
                    launch = g.launcher(sid)
                    checked = ""
                    if c_ref is not None:
                        g.c.zero_()
                        launch(0)
                        torch.cuda.synchronize()
                        # two exact kernels differ by f32 summation order and one 16-bit rounding: 1 % of the value, or 2 % of
                        # the output rms where the value itself cancels; a wrong tile / layout is off by ~ the rms itself.
                        # Native kernels quantise the activations (own accuracy class): each is compared with the FIRST native
                        # kernel of its activation format (same quantised inputs: only the summation order differs), as the
                        # in-library tuner does (csrc/tune.hip); that first one is covered by tests/test_gpu_parity.py.
                        ref = c_ref
                        if is_native(sid):
                            ref = native_refs.setdefault((sid >> 32) & 7, g.c.float().clone())
                        err = (g.c.float() - ref).abs()
                        bad = err > torch.clamp(ref.abs() * 1e-2, min=2e-2 * ref.pow(2).mean().sqrt().item())
                        if bad.any():
                            dropped.append({"solution": f"0x{sid:x}", "desc": _lib.describe_solution(sid),
                                            "mismatches": int(bad.sum()), "max_err": float(err.max())})
                            continue
                        checked = "ok"
                    r = g.time(sid, stream, reps=args.reps, launches=args.launches)
                except Exception as exc:  # noqa: BLE001
                    dropped.append({"solution": f"0x{sid:x}", "error": str(exc)})
                    continue
                rec = {"solution": f"0x{sid:x}", "desc": _lib.describe_solution(sid), "us_median": r["us"], "us_min": r["us_min"],
                       "gbs": r["gbs"], "frac_hbm": r["gbs"] / BL.HBM_PEAK_GBS, "tflops": r["tflops"],
                       "is_default": sid == default_sid, "native": is_native(sid)}
                results.append(rec)
                cw.writerow([args.dtype, args.fmt, name, n, k, m, rec["solution"], rec["desc"], f"{r['us']:.3f}", f"{r['us_min']:.3f}",
                             f"{r['gbs']:.1f}", f"{rec['frac_hbm']:.4f}", f"{r['tflops']:.2f}", int(rec["is_default"]), checked])
            dense = None
            if args.compare_dense:
                hb = BL.HipblasLtGemm(m, n, k, dtype, dev, args.rotate_mb)
                hb.check()
                d = hb.time(stream, reps=args.reps)
                dense = {"us_median": d["us"], "tflops": d["tflops"], "library": "hipBLASLt (HIPBLAS_COMPUTE_32F, TRANSA=T)"}
                hb.close()
                del hb
            ok = sorted(results, key=lambda r: r["us_median"])
            exact_ok = [r for r in ok if not r["native"]]
            cell = {"shape": name, "n": n, "k": k, "m": m, "candidates": len(cands), "dropped": dropped,
                    "best": exact_ok[0] if exact_ok else None, "best_native": next((r for r in ok if r["native"]), None),
                    "default": next((r for r in ok if r["is_default"]), None), "top5": ok[:5], "dense_16bit_gemm": dense}
            summary["cells"].append(cell)
            if dropped:
                print(f"{name:8s} M={m:<3d} DROPPED {len(dropped)}: " + "; ".join(str(d) for d in dropped[:3]), flush=True)
            if exact_ok:
                best, dflt = exact_ok[0], cell["default"]
                print(f"{name:8s} M={m:<3d} best {best['us_median']:8.2f} us {best['gbs']:7.0f} GB/s "
                      f"({100 * best['frac_hbm']:.1f}% of 8 TB/s) {best['tflops']:.0f} TF {best['desc']}"
                      + (f" | default {dflt['us_median']:.2f} us" if dflt else "")
                      + (f" | native {cell['best_native']['us_median']:.2f} us {cell['best_native']['tflops']:.0f} TF" if cell["best_native"] else "")
                      + (f" | hipBLASLt {dense['us_median']:.1f} us {dense['tflops']:.0f} TF" if dense else ""),
                      flush=True)
                rows.append((g.a_type, g.b_type, n, k, m, int(best["solution"], 16)))   # never a native kernel
        del w
        torch.cuda.empty_cache()
    summary["elapsed_s"] = time.time() - t_start
    csv_f.close()
    out.write_text(json.dumps(summary, indent=1))
    tune_txt = out.with_suffix(".tune.txt")
    with open(tune_txt, "w") as f:
        f.write("# a_type b_type n k m_lo m_hi solution   (tools/tune.py; $PETIT_AMD_TUNE_FILE format)\n")
        for (at, bt, n, k, m, sid) in rows:
            f.write(f"{at} {bt} {n} {k} {m} {m} {sid:x}\n")
    print(f"wrote {out}, {out.with_suffix('.csv')} and {tune_txt}")


if __name__ == "__main__":
    main()
