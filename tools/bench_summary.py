#!/usr/bin/env python3
"""tools/bench_summary.py <bench.json> [<bench_steps20.json>] -- the cell tables of profiles/rNN_summary.md from a bench.py line (stdout: markdown)."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
d20 = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]) if len(sys.argv) > 2 else None
PEAK = {"hipblaslt_fp8": 5000.0, "hipblaslt_fp8_best": 5000.0, "native_mxfp8": 5000.0, "native_mxfp6": 10000.0, "native_mxfp4": 10000.0, "mlp_native_mxfp8_pipeline": 5000.0}


def expand(d):
    """round 5: the line carries one row per (shape, dtypes mode): [shape, dt, [us per M], [rate per M]] over d["cells_m"][dt];
    round 4: [shape, M, dt, us, rate, frac]; earlier / the side file: dicts."""
    out = []
    for c in d["cells"]:
        if isinstance(c, dict):
            out.append(c)
        elif "cells_m" in d:
            shape, dt, us, rates = c
            for m, u, r in zip(d["cells_m"][dt], us, rates):
                if u is None:
                    continue
                hbm = (m <= 64 and shape != "mlp") or shape == "tp8_layer"
                mode = dt.split()[-1] if " " in dt else ""
                peak = 8000.0 if hbm else PEAK.get(mode, 10000.0 if mode.startswith("mlp_native") else 2500.0)
                out.append({"shape": shape, "M": m, "dt": dt, "us": u, "frac": r / peak, "GBs" if hbm else "TF": r})
        else:
            cell = {"shape": c[0], "M": c[1], "dt": c[2], "us": c[3], "frac": c[5]}
            cell["GBs" if (c[1] <= 16 and c[0] != "mlp") else "TF"] = c[4]
            out.append(cell)
    return out


cells = {(c["shape"], c["M"], c["dt"]): c for c in expand(d)}
COPY_CEILING = 6290.0   # GB/s: the guide's measured HBM copy ceiling (what a pure stream reaches of the 8 TB/s spec)
shapes = ["qkv", "o", "gate_up", "down"]
print(f"Headline (BASELINE configs[1], M = 1, N = K = 8192, bf16 x NVFP4, solution_id = -1): **{d['ms_per_step'] * 1e3:.2f} us/step = {d['value']:.0f} GB/s = "
      f"{d['roofline']['frac']:.3f} of 8 TB/s** ({d['steps']} graph-replayed steps x {d['config'].get('timed_regions', 1)} regions, median)"
      + (f"; as the driver runs it ({d20['steps']} steps): {d20['ms_per_step'] * 1e3:.2f} us = {d20['roofline']['frac']:.3f}." if d20 else "."))
print("\nHBM-bound cells (us per call, fraction of the 8 TB/s spec / of the 6.29 TB/s copy ceiling a pure stream reaches):\n")
cols = [("bf16xnv", 1), ("bf16xnv", 4), ("bf16xnv", 8), ("bf16xnv", 16), ("fp16xnv", 16), ("fp16xmx", 1), ("fp16xmx", 16), ("bf16xmx", 1), ("bf16xmx", 16)]
print("| shape | " + " | ".join(f"{dt} M={m}" for dt, m in cols) + " |")
print("|---|" + "---|" * len(cols))
for s in shapes:
    print(f"| {s} | " + " | ".join(f"{cells[(s, m, dt)]['us']:.2f} us, {cells[(s, m, dt)]['frac']:.2f} / {cells[(s, m, dt)]['GBs'] / COPY_CEILING:.2f}" if (s, m, dt) in cells else "-"
                                   for dt, m in cols) + " |")
print("\nM = 512 (TFLOP/s; fraction of 2.5 PF bf16 peak, native: of the 5 / 10 PF FP8 / FP4 peaks; native cells include the activation-quantiser launch):\n")
cols = [("bf16xnv", "bf16 x NVFP4"), ("fp16xnv", "fp16 x NVFP4"), ("bf16xmx", "bf16 x MXFP4"), ("fp16xmx", "fp16 x MXFP4"), ("bf16xmx native_mxfp8", "native, act -> MXFP8 (-2)"), ("bf16xmx native_mxfp6", "native, act -> MXFP6 (-4)"),
        ("bf16xmx native_mxfp4", "native, act -> MXFP4 (-3)"), ("bf16xnv native_mxfp8", "NVFP4 image x MXFP8 (-2)"), ("bf16xnv native_mxfp6", "NVFP4 image x MXFP6 (-4)"), ("bf16xnv native_mxfp4", "NVFP4 image x MXFP4 (-3)"),
        ("bf16xdense hipblaslt", "hipBLASLt bf16 dense"), ("fp8xdense hipblaslt_fp8", "hipBLASLt FP8 dense")]
print("| shape | " + " | ".join(name for _, name in cols) + " |")
print("|---|" + "---|" * len(cols))
for s in shapes:
    print(f"| {s} | " + " | ".join(f"{cells[(s, 512, dt)]['us']:.1f} us, {cells[(s, 512, dt)]['TF']:.0f} TF, {cells[(s, 512, dt)]['frac']:.2f}" if (s, 512, dt) in cells else "-"
                                   for dt, _ in cols) + " |")
if any(k[1] == 32 for k in cells):
    print("\nMid M (17 <= M <= 128; bf16 activations; M <= 64: us, fraction of 8 TB/s / of the copy ceiling; M = 128: us, TFLOP/s):\n")
    cols = [(dt, m) for dt in ("bf16xnv", "bf16xmx") for m in (32, 44, 64, 128)]
    print("| shape | " + " | ".join(f"{dt} M={m}" for dt, m in cols) + " |")
    print("|---|" + "---|" * len(cols))
    for s in shapes:
        def fmt(c):
            return f"{c['us']:.2f} us, {c['frac']:.2f} / {c['GBs'] / COPY_CEILING:.2f}" if "GBs" in c else f"{c['us']:.1f} us, {c['TF']:.0f} TF"
        print(f"| {s} | " + " | ".join(fmt(cells[(s, m, dt)]) if (s, m, dt) in cells else "-" for dt, m in cols) + " |")
pre = sorted({k[1] for k in cells if k[1] > 512})
if pre:
    print("\nPrefill (M > 512; TFLOP/s, native cells include the activation-quantiser launch):\n")
    cols = [("bf16xnv", "bf16 x NVFP4"), ("bf16xmx", "bf16 x MXFP4"), ("bf16xmx native_mxfp8", "native MXFP8"), ("bf16xmx native_mxfp6", "native MXFP6"),
            ("bf16xmx native_mxfp4", "native MXFP4"), ("bf16xnv native_mxfp8", "NVFP4 image x MXFP8"), ("bf16xnv native_mxfp6", "NVFP4 image x MXFP6"), ("bf16xnv native_mxfp4", "NVFP4 image x MXFP4"),
            ("bf16xdense hipblaslt", "hipBLASLt bf16"), ("bf16xdense hipblaslt_best", "hipBLASLt bf16, best of the heuristic's results"), ("fp8xdense hipblaslt_fp8", "hipBLASLt FP8"), ("fp8xdense hipblaslt_fp8_best", "hipBLASLt FP8, best")]
    print("| shape | M | " + " | ".join(name for _, name in cols) + " |")
    print("|---|---|" + "---|" * len(cols))
    for s in shapes:
        for m in pre:
            print(f"| {s} | {m} | " + " | ".join(f"{cells[(s, m, dt)]['us']:.0f} us, {cells[(s, m, dt)]['TF']:.0f} TF" if (s, m, dt) in cells else "-" for dt, _ in cols) + " |")
if any(k[1] == 256 for k in cells):
    print("\nfp16 x NVFP4 at M = 256 (the reference benchmark's middle column): " +
          ", ".join(f"{s} {cells[(s, 256, 'fp16xnv')]['us']:.1f} us = {cells[(s, 256, 'fp16xnv')]['TF']:.0f} TFLOP/s" for s in shapes if (s, 256, "fp16xnv") in cells) + ".")
big = [k for k in cells if k[2] == "fp16xmx" and k[1] > 512]
if big:
    m_big = max(k[1] for k in big)
    print(f"\nfp16 x MXFP4 at M = {m_big} (the family x regime of round 5's tuner-check bug): " +
          ", ".join(f"{s} {cells[(s, m_big, 'fp16xmx')]['us']:.0f} us = {cells[(s, m_big, 'fp16xmx')]['TF']:.0f} TFLOP/s" for s in shapes if (s, m_big, "fp16xmx") in cells) + ".")
tp8 = ["tp8_qkv", "tp8_o", "tp8_gate_up", "tp8_down"]
if any(k[0] in tp8 for k in cells):
    print("\nTP = 8 shard shapes (bf16 x NVFP4, solution_id = -1; M <= 64: us, fraction of 8 TB/s; M = 512: us, TFLOP/s) -- qkv 1280 x 8192, o 8192 x 1024, gate_up 7168 x 8192, down 8192 x 3584:\n")
    ms = sorted({k[1] for k in cells if k[0] in tp8})
    print("| shape | " + " | ".join(f"M={m}" for m in ms) + " |")
    print("|---|" + "---|" * len(ms))
    for s in tp8:
        def fmt(c):
            return f"{c['us']:.2f} us, {c['frac']:.2f}" if "GBs" in c else f"{c['us']:.1f} us, {c['TF']:.0f} TF"
        print(f"| {s} | " + " | ".join(fmt(cells[(s, m, 'bf16xnv')]) if (s, m, "bf16xnv") in cells else "-" for m in ms) + " |")
    for (s, m, dt), c in cells.items():
        if s == "tp8_layer":
            print(f"\nOne decode layer's four launches at TP = 8 (grouped q / k / v -> o -> gate_up + SiLU-mul -> down), M = {m}: {c['us']:.2f} us = {c['GBs']:.0f} GB/s of algorithmic bytes = {c['frac']:.2f} of 8 TB/s.")
print("\nLaunch-gap-bound shapes and the MLP block:\n\n| cell | us | rate |\n|---|---|---|")
for (s, m, dt), c in cells.items():
    if s == "tp8_qkv_3x1280":
        print(f"| three 1280 x 8192 shards, M = {m}, {dt.split()[-1]} | {c['us']:.2f} | {c['GBs']} GB/s |")
    elif s == "mlp":
        print(f"| Llama-70B MLP block M = 512, {dt.split()[-1].replace('mlp_', '')} | {c['us']:.1f} | {c['TF']:.0f} TFLOP/s |")
