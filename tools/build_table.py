#!/usr/bin/env python3
"""tools/build_table.py -- the built-in arch table from the in-library tuner, over the linear shapes of several model families.

    python tools/build_table.py [--part table|heldout|all] [--families ...] [--ms ...] [--out-dir gpurun_out/r04_table]

For every (family, shape, M): petit_kernel.tune_tensors(persist=False) -- every kernel that fits is checked against the class's reference
kernel and timed on rotating weights (csrc/tune.hip) -- and one row "a_type b_type n k m m solution" for tools/make_tuned_inc.py.  With
$PETIT_AMD_TUNE_LOG set (this script sets it) the library also logs every candidate's timing: the data tools/check_heuristic.py replays
the heuristic against (shapes of the `heldout` part get no table row: they measure what an unseen shape gets).
The reference tells its users to tune per shape with its benchmark binary (tools/benchmarks/matmul.py:14-117, README.md:35); this is that
sweep, kept in the library's table."""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "petit-kernel_amd"))
sys.path.insert(0, str(ROOT / "tools"))


def model_shapes():
    """(N, K) of every linear of the listed models at TP in {1, 2, 4, 8}: column-parallel layers (qkv, gate_up) shard N, row-parallel
    ones (o, down) shard K.  Kept: N % 32 == 0 (the MX scale tensor contract) and K % 256 == 0 (the packed layout's smallest span)."""
    models = {
        # name: (hidden, q_heads, kv_heads, head_dim, intermediate)
        "llama3-8b": (4096, 32, 8, 128, 14336),
        "llama3-70b": (8192, 64, 8, 128, 28672),
        "llama3.1-405b": (16384, 128, 8, 128, 53248),
        "qwen2.5-72b": (8192, 64, 8, 128, 29568),
        "qwen3-32b": (5120, 64, 8, 128, 25600),
        "mixtral-8x22b": (6144, 48, 8, 128, 16384),
        "deepseek-v3-dense": (7168, 128, 128, 128, 18432),   # (MLA: its own projections below; the three dense MLP layers)
    }
    out = {}
    for name, (hid, qh, kvh, hd, inter) in models.items():
        for tp in (1, 2, 4, 8):
            if name != "deepseek-v3-dense":
                kv = max(kvh // tp, 1)
                out.setdefault(((qh // tp + 2 * kv) * hd, hid), f"{name} qkv tp{tp}")
                out.setdefault((hid, qh * hd // tp), f"{name} o tp{tp}")
            out.setdefault((2 * inter // tp, hid), f"{name} gate_up tp{tp}")
            out.setdefault((hid, inter // tp), f"{name} down tp{tp}")
    # DeepSeek-V3 MLA projections and expert MLPs (hidden 7168, q_lora 1536, kv_lora 512 + 64 rope, 128 heads x (128 + 64) / 128, moe inter 2048)
    for nk, what in (((1536, 7168), "q_a"), ((24576, 1536), "q_b"), ((576, 7168), "kv_a"), ((32768, 512), "kv_b"), ((7168, 16384), "o"),
                     ((4096, 7168), "expert gate_up"), ((7168, 2048), "expert down")):
        out.setdefault(nk, f"deepseek-v3 {what}")
    for tp in (2, 4, 8):
        out.setdefault((24576 // tp, 1536), f"deepseek-v3 q_b tp{tp}")
        out.setdefault((7168, 16384 // tp), f"deepseek-v3 o tp{tp}")
    # the shapes the table already had (BASELINE's square shapes, Llama TP-8 shards)
    for nk in ((8192, 8192), (4096, 4096), (1280, 8192), (8192, 1024), (7168, 8192), (8192, 3584), (6144, 4096), (28672, 4096), (4096, 14336)):
        out.setdefault(nk, "r01-r03 table")
    return {nk: what for nk, what in out.items() if nk[0] % 32 == 0 and nk[1] % 256 == 0 and nk[0] * nk[1] // 2 < (1 << 31) and nk[1] * 128 < (1 << 31)}


# shapes that get NO table row: what an unseen shape gets from the heuristic (tools/check_heuristic.py --heldout)
HELDOUT = {(12288, 4096), (5120, 13824), (16384, 5120), (5120, 8192), (9216, 6144), (6144, 2048), (13824, 5120), (3072, 8192), (8192, 7168), (20480, 4096)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--part", default="table", choices=["table", "heldout", "all"])
    ap.add_argument("--families", default="nv:bf16,nv:f16,mx:bf16,mx:f16")
    ap.add_argument("--ms", default="1,2,4,8,16,32,64,128,256,512")
    ap.add_argument("--out-dir", default=str(ROOT / "gpurun_out" / "r04_table"))
    ap.add_argument("--budget-s", type=float, default=1e9)
    ap.add_argument("--klass", default="exact", choices=["exact", "native_mxfp8", "native_mxfp6", "native_mxfp4"],
                    help="the native classes (MXFP4 weights only: --families mx:bf16,mx:f16) fill csrc/tuned_native_gfx950.inc: tools/make_tuned_inc.py --native <out>.tune.txt")
    ap.add_argument("--only", default="", help="comma-separated substrings: keep the shapes whose description (model, layer, TP) contains one, e.g. llama3-70b,r01-r03")
    ap.add_argument("--n-multiple", type=int, default=0, help="keep only shapes whose N is a multiple of this (e.g. 320: the shapes a 320-column tile divides)")
    ap.add_argument("--k-not-multiple", type=int, default=0, help="keep only shapes whose K is NOT a multiple of this (1024: the shapes that need KS = 4 / 2 kernels)")
    ap.add_argument("--n-max", type=int, default=0, help="keep only shapes with N <= this (the narrow shapes whose rows name batched-decode kernels at prefill M)")
    ap.add_argument("--samples", type=int, default=5)
    ap.add_argument("--list", action="store_true", help="print the shape list and exit (no GPU needed)")
    args = ap.parse_args()
    shapes = model_shapes()
    table = sorted(nk for nk in shapes if nk not in HELDOUT)
    held = sorted(HELDOUT)
    todo = table if args.part == "table" else held if args.part == "heldout" else table + held
    if args.only:
        keys = [x for x in args.only.split(",") if x]
        todo = [nk for nk in todo if any(x in shapes.get(nk, "held-out") for x in keys)]
    if args.k_not_multiple:
        todo = [nk for nk in todo if nk[1] % args.k_not_multiple != 0]
    if args.n_max:
        todo = [nk for nk in todo if nk[0] <= args.n_max]
    if args.n_multiple:
        todo = [nk for nk in todo if nk[0] % args.n_multiple == 0]
    if args.list:
        for nk in todo:
            print(f"{nk[0]}x{nk[1]}  {shapes.get(nk, 'held-out')}")
        print(f"{len(table)} table shapes, {len(held)} held-out shapes, x {len(args.ms.split(','))} M x {len(args.families.split(','))} families")
        return
    out_dir = Path(args.out_dir)
    out_dir.mkdir(parents=True, exist_ok=True)
    tag = args.part if args.klass == "exact" else f"{args.part}_{args.klass}"
    os.environ["PETIT_AMD_TUNE_LOG"] = str(out_dir / f"candidates_{tag}.csv")
    import torch

    import benchlib as BL
    import petit_kernel as pk
    dev = torch.device("cuda", 0)
    ms = [int(x) for x in args.ms.split(",")]
    rows, t0 = [], time.time()
    for fam in args.families.split(","):
        fmt, dt = fam.split(":")
        dtype = torch.bfloat16 if dt == "bf16" else torch.float16
        for (n, k) in todo:
            if time.time() - t0 > args.budget_s:
                break
            w = BL.Weights(fmt, n, k, 1280, dev, max_copies=48)
            for m in ms:
                g = BL.Gemm(w, m, dtype, dev)
                try:
                    sid, us = pk.tune_tensors(g.a, w.packed, g.gs, m, n, k, {"nv": "nvfp4", "mx": "mxfp4"}[fmt], klass=args.klass, persist=False, samples=args.samples)
                except RuntimeError as exc:
                    print(f"{fam} {n}x{k} M={m}: {exc}", flush=True)
                    continue
                rows.append((g.a_type, g.b_type, n, k, m, sid, us))
            del w
            torch.cuda.empty_cache()
            print(f"[{time.time() - t0:6.0f} s] {fam} {n}x{k} done ({len(rows)} rows)", flush=True)
    with open(out_dir / f"{tag}.tune.txt", "w") as f:
        f.write("# a_type b_type n k m_lo m_hi solution   (tools/build_table.py; $PETIT_AMD_TUNE_FILE format; us per launch in the json beside it)\n")
        for (at, bt, n, k, m, sid, us) in rows:
            f.write(f"{at} {bt} {n} {k} {m} {m} {sid:x}\n")
    (out_dir / f"{tag}.json").write_text(json.dumps({"elapsed_s": time.time() - t0, "rows": [list(r[:5]) + [f"{r[5]:x}", r[6]] for r in rows]}))
    print(f"{len(rows)} rows in {time.time() - t0:.0f} s -> {out_dir}")


if __name__ == "__main__":
    main()
