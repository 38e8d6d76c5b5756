#!/usr/bin/env python3
"""tools/refresh_table.py [--shapes NxK,...] [--ms ...] [--out gpurun_out/refresh.tune.txt] -- re-validate the built-in arch table with the in-library tuner.

For every (family, shape, M bucket): time what `solution_id = -1` runs today (tools/benchlib.py: graph replay, rotating weights), ask the in-library tuner
(petit_kernel.tune_tensors, persist = False: ~0.1 s per problem instead of the exhaustive sweep's ~20 s) for its pick, time that the same way, time the default
again; where the tuner's pick beats BOTH default timings by more than --gain (default 5 %; 3 % when the same challenger won an earlier session too: --confirm), write a row in the $PETIT_AMD_TUNE_FILE format for
tools/make_tuned_inc.py.  Rows of rounds that predate newer kernel families go stale silently; this finds them in minutes."""
import argparse
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "petit-kernel_amd"))
sys.path.insert(0, str(ROOT / "tools"))
import torch

import benchlib as BL
import petit_kernel as pk
from petit_kernel import _lib

ALL = "8192x8192,10240x8192,57344x8192,8192x28672,6144x4096,4096x4096,28672x4096,4096x14336,1280x8192,8192x1024,7168x8192,8192x3584"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default=ALL)
    ap.add_argument("--ms", default="1,2,4,8,16,32,64,128,256,512")
    ap.add_argument("--families", default="nv:bf16,nv:f16,mx:bf16,mx:f16")
    ap.add_argument("--gain", type=float, default=0.05, help="a challenger seen in THIS session only must beat both default timings by this much")
    ap.add_argument("--confirm", default="", help="the .json log of an EARLIER refresh session: a challenger that also won there (same id, >= --confirm-gain in both "
                                                    "sessions) is adopted at the lower margin -- single-session timings sit inside the noise the picks are made on")
    ap.add_argument("--confirm-gain", type=float, default=0.03)
    ap.add_argument("--out", default=str(ROOT / "gpurun_out" / "refresh.tune.txt"))
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(dev)
    rows, log = [], []
    earlier = {}
    if args.confirm:
        for r in json.loads(Path(args.confirm).read_text())["log"]:
            if "tuned_us" in r and r["tuned_us"] < (1.0 - args.confirm_gain) * min(r["default_us"], r.get("default_us_again", r["default_us"])):
                earlier[(r["family"], r["n"], r["k"], r["m"])] = r["tuned"]
    t0 = time.time()
    for fam in args.families.split(","):
        fmt, dt = fam.split(":")
        dtype = torch.bfloat16 if dt == "bf16" else torch.float16
        for name in args.shapes.split(","):
            n, k = (int(x) for x in name.split("x"))
            w = BL.Weights(fmt, n, k, 1280, dev)
            for m in [int(x) for x in args.ms.split(",")]:
                g = BL.Gemm(w, m, dtype, dev)
                dflt = g.resolve(_lib.PETIT_SOLUTION_AUTO)
                t_d1 = g.time(dflt, stream, reps=5)["us"]
                with torch.cuda.stream(stream):
                    tuned, _ = pk.tune_tensors(g.a, w.packed, g.gs, m, n, k, {"nv": "nvfp4", "mx": "mxfp4"}[fmt], persist=False)
                rec = {"family": fam, "n": n, "k": k, "m": m, "default": f"0x{dflt:x}", "default_us": t_d1, "tuned": f"0x{tuned:x}"}
                if tuned != dflt:
                    t_t = g.time(tuned, stream, reps=5)["us"]
                    t_d2 = g.time(dflt, stream, reps=5)["us"]
                    rec.update(tuned_us=t_t, default_us_again=t_d2)
                    confirmed = earlier.get((fam, n, k, m)) == f"0x{tuned:x}" and t_t < (1.0 - args.confirm_gain) * min(t_d1, t_d2)
                    rec["confirmed_by_earlier_session"] = confirmed
                    if confirmed or t_t < (1.0 - args.gain) * min(t_d1, t_d2):
                        rows.append((g.a_type, g.b_type, n, k, m, tuned))
                        rec["replaced"] = True
                        print(f"{fam} {name} M={m}: {min(t_d1, t_d2):.2f} -> {t_t:.2f} us  {_lib.describe_solution(tuned)}", flush=True)
                log.append(rec)
            del w
            torch.cuda.empty_cache()
    out = Path(args.out)
    out.parent.mkdir(parents=True, exist_ok=True)
    with open(out, "w") as f:
        f.write("# a_type b_type n k m_lo m_hi solution   (tools/refresh_table.py; $PETIT_AMD_TUNE_FILE format)\n")
        for (at, bt, n, k, m, sid) in rows:
            f.write(f"{at} {bt} {n} {k} {m} {m} {sid:x}\n")
    out.with_suffix(".json").write_text(json.dumps({"elapsed_s": time.time() - t0, "problems": len(log), "replaced": len(rows), "log": log}, indent=1))
    print(f"{len(log)} problems, {len(rows)} rows to replace, {time.time() - t0:.0f} s; wrote {out}")


if __name__ == "__main__":
    main()
