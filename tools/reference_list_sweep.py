#!/usr/bin/env python3
"""tools/reference_list_sweep.py -- the reference's own benchmark problem list, every entry, timed here.

The reference's tools/benchmarks/matmul.py:8-117 holds two lists of (m, n, k): 80 problems at the M values a serving trace of Llama-3 8B / 70B / 70B at TP = 8
produced (15 ... 16375, ragged) and 24 at M = 16 / 256 / 512; its defaults are fp16 x nvfp4 -> fp16, `-algo tune`, 5 warm-up + 20 timed launches
(matmul.py:119-127).  This tool restates that list by shape family, runs every problem through the C ABI with solution_id = -1 (what a caller gets without tuning) and,
next to it, hipBLASLt's 16-bit dense GEMM of the same size (first heuristic result and the best of the results, the reference's `-backend hipblaslt -algo tune`).
One JSON line per problem; `--md` turns a log into the table of profiles/rNN_reference_list.md.

    python tools/reference_list_sweep.py --out gpurun_out/r06_reference_list.jsonl [--atype fp16] [--btype nv] [--native]
    python tools/reference_list_sweep.py --md a.jsonl[,b.jsonl] > profiles/r06_reference_list.md
"""
import argparse
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent

# (n, k) families of matmul.py's list
L8B = [(4096, 4096), (4096, 14336), (6144, 4096), (28672, 4096)]            # Llama-3 8B: o, down, qkv, gate_up
L70B = [(8192, 8192), (8192, 28672), (10240, 8192), (57344, 8192)]          # Llama-3 70B: o, down, qkv, gate_up
L70B_TP8 = [(1280, 8192), (7168, 8192), (8192, 1024), (8192, 3584)]         # 70B at TP = 8: qkv, gate_up, o, down
# M -> families (matmul.py:9-90 trace list, then :92-117)
TRACE = [(15, [L8B, L70B]), (44, [L8B, L70B, L70B_TP8]), (566, [L70B_TP8]), (582, [L8B]), (611, [L8B]), (874, [L70B]), (932, [L70B]), (1003, [L70B]),
         (1324, [L8B]), (1340, [L70B_TP8]), (1466, [L70B_TP8]), (1906, [L70B_TP8]), (2084, [L70B]), (4314, [L8B]), (14437, [L8B]), (15961, [L70B_TP8]),
         (16375, [L70B])]
ROUND = [(16, [L8B, L70B]), (256, [L8B, L70B]), (512, [L8B, L70B])]


def problems():
    out = []
    for m, fams in TRACE + ROUND:
        for fam in fams:
            for n, k in fam:
                out.append((m, n, k))
    return out


def family(n, k):
    return "8B" if (n, k) in L8B else "70B" if (n, k) in L70B else "70B/TP8"


def sweep(args):
    import torch
    sys.path.insert(0, str(ROOT / "tools"))
    sys.path.insert(0, str(ROOT / "petit-kernel_amd"))
    import benchlib as BL
    from petit_kernel import _lib

    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream()
    dtype = torch.float16 if args.atype == "fp16" else torch.bfloat16
    probs = problems()
    if args.ms:
        keep = {int(x) for x in args.ms.split(",")}
        probs = [p for p in probs if p[0] in keep]
    assert len(problems()) == 104
    sink = open(args.out, "a") if args.out else None
    by_shape = {}
    for m, n, k in probs:
        by_shape.setdefault((n, k), []).append(m)
    for (n, k), ms in by_shape.items():
        torch.cuda.empty_cache()
        w = BL.Weights(args.btype, n, k, 1280, dev)
        for m in ms:
            g = BL.Gemm(w, m, dtype, dev)
            rec = {"m": m, "n": n, "k": k, "family": family(n, k), "dt": f"{args.atype}x{args.btype}"}
            picked = g.resolve(_lib.PETIT_SOLUTION_AUTO)
            r = g.time(_lib.PETIT_SOLUTION_AUTO, stream, reps=5)
            rec.update(us=round(r["us"], 2), TF=round(r["tflops"], 1), GBs=round(r["gbs"], 0), kernel=_lib.describe_solution(picked))
            split = int(_lib.lib.petit_gemm_row_split(BL.C.byref(g.hints), m, n, k, BL.C.c_uint64(_lib.PETIT_SOLUTION_AUTO), None))
            if split:
                rec["bulk_rows"] = split
            if args.native and m >= 256:
                for name, sid in (("native_mxfp8", _lib.PETIT_SOLUTION_AUTO_NATIVE_MXFP8), ("native_mxfp6", _lib.PETIT_SOLUTION_AUTO_NATIVE_MXFP6)):
                    try:
                        rn = g.time(sid, stream, reps=5)
                        rec[name + "_us"] = round(rn["us"], 2)
                        rec[name + "_TF"] = round(rn["tflops"], 1)
                    except RuntimeError as exc:
                        rec[name + "_error"] = str(exc)
            if not args.no_vendor:
                try:
                    h = BL.HipblasLtGemm(m, n, k, dtype, dev, rotate_mb=1280)
                    first = h.time(stream, reps=5)
                    best = h.time_best(stream, first, max_algos=args.algos)
                    rec.update(hbl_us=round(first["us"], 2), hbl_best_us=round(best["us"], 2), hbl_best_algo=best["algo_index"], hbl_algos=best["algos_timed"])
                    h.close()
                except Exception as exc:  # noqa: BLE001 -- the comparator must not stop the sweep
                    rec["hbl_error"] = str(exc)[:120]
            line = json.dumps(rec)
            print(line, flush=True)
            if sink:
                sink.write(line + "\n")
                sink.flush()
            del g
        w.detach_native()
        del w


def table(path):
    recs = [json.loads(x) for one in path.split(",") for x in open(one) if x.strip().startswith("{")]
    order = {(m, n, k): i for i, (m, n, k) in enumerate(sorted(problems(), key=lambda p: (p[0], family(p[1], p[2]), p[1], p[2])))}
    recs.sort(key=lambda r: order[(r["m"], r["n"], r["k"])])
    dts = sorted({r["dt"] for r in recs})
    print("# The reference's benchmark list (tools/benchmarks/matmul.py:8-117), every entry, on one MI355X\n")
    print("`python tools/reference_list_sweep.py`: solution_id = -1 through the C ABI, graph-replayed launches over rotating weight copies (tools/benchlib.py); hipBLASLt = its 16-bit dense GEMM of the same "
          "(m, n, k), first heuristic result / best of the results timed (`-algo tune`).  ratio = hipBLASLt best time / ours (> 1: the 4-bit path is faster than the vendor's dense one).\n")
    for dt in dts:
        rows = [r for r in recs if r["dt"] == dt]
        has_native = any("native_mxfp8_us" in r for r in rows)
        print(f"## {dt} ({len(rows)} problems)\n")
        head = "| M | family | n x k | us | TFLOP/s | GB/s | kernel | hipBLASLt us (first / best) | ratio |"
        sep = "|---|---|---|---|---|---|---|---|---|"
        if has_native:
            head += " native MXFP8 us (TF) | native MXFP6 us (TF) |"
            sep += "---|---|"
        print(head + "\n" + sep)
        ratios = {}
        for r in rows:
            kern = r["kernel"].split(":")[0][:40] + (f" (bulk {r['bulk_rows']} + tail)" if r.get("bulk_rows") else "")
            hb = f"{r['hbl_us']:.1f} / {r['hbl_best_us']:.1f}" if "hbl_us" in r else "-"
            ratio = r["hbl_best_us"] / r["us"] if "hbl_best_us" in r else None
            if ratio:
                ratios.setdefault(r["m"], []).append(ratio)
            line = f"| {r['m']} | {r['family']} | {r['n']} x {r['k']} | {r['us']:.1f} | {r['TF']:.0f} | {r['GBs']:.0f} | {kern} | {hb} | {ratio:.2f} |" if ratio else \
                   f"| {r['m']} | {r['family']} | {r['n']} x {r['k']} | {r['us']:.1f} | {r['TF']:.0f} | {r['GBs']:.0f} | {kern} | {hb} | - |"
            if has_native:
                for nm in ("native_mxfp8", "native_mxfp6"):
                    line += f" {r[nm + '_us']:.1f} ({r[nm + '_TF']:.0f}) |" if nm + "_us" in r else " - |"
            print(line)
        if ratios:
            print("\nGeometric mean of the ratio per M: " + ", ".join(f"M={m}: {_gm(v):.2f}" for m, v in sorted(ratios.items())) + ".\n")


def _gm(v):
    p = 1.0
    for x in v:
        p *= x
    return p ** (1.0 / len(v))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--atype", default="fp16", choices=["fp16", "bf16"])
    ap.add_argument("--btype", default="nv", choices=["nv", "mx"])
    ap.add_argument("--ms", default="")
    ap.add_argument("--native", action="store_true", help="also time the native class (MXFP8 / MXFP6 activations) at M >= 256")
    ap.add_argument("--no-vendor", action="store_true")
    ap.add_argument("--algos", type=int, default=12)
    ap.add_argument("--md", default="")
    args = ap.parse_args()
    if args.md:
        table(args.md)
    else:
        sweep(args)


if __name__ == "__main__":
    main()
