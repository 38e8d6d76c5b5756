#!/usr/bin/env python3
"""bench.py -- the headline measurement of this repo (driver contract).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Headline (the fields the driver reads; BASELINE.json configs[1], the configuration the metric is quoted on):
    one step = ONE call of the hot path, C[1,8192] = A[1,8192] . dequant(W[8192,8192])^T * gs,
    bf16 activations x NVFP4 weights (e4m3 scales, group 16), through the drop-in Python
    surface petit_kernel.mul_nvfp4_a16(..., solution_id=-1) -> C ABI -> HIP kernel.
    Inputs are synthetic (tests/ops/test_fp4_gemm_quark.py:41-46 distributions, seed 1234),
    resident in HBM before the timed region, and ROTATE over enough distinct (W, scales)
    copies (default 1.3 GB) that no launch re-reads weights still sitting in the 256 MB
    Infinity Cache -- the reference's own benchmark reuses a single buffer
    (tools/benchmarks/matmul/rocm/matmul_petit.cc:116-132), which on MI355X would measure the
    cache, not HBM.
    The K steps are one HIP-graph replay, bracketed by barrier + synchronize on both sides as the
    contract asks; that bracketed region is repeated `--repeats` times (default 11) and the MEDIAN is
    reported, so a short run (--steps 20 = 0.17 ms) is not a single sample.

The rest of the metric ("TFLOPS + achieved HBM GB/s, M in {1,8,16,512}, Llama-70B shapes") is in `cells`
(rank 0, N = 1 only; the list is tools/benchlib.py bench_cell_plan(), shared with tests/test_gpu_parity.py::
test_bench_cells_parity so that every timed cell is also checked against the oracle): for qkv / o / gate_up / down,
bf16 x NVFP4 at M in {1, 4, 8, 16, 512}, fp16 x NVFP4 (the reference benchmark's default dtype, tools/benchmarks/matmul.py:92-127)
at M in {16, 512}, fp16 x MXFP4 (BASELINE configs[3]) at M in {1, 16}, bf16 x MXFP4 at M = 512 -- all through solution_id = -1 --
plus at M = 512 the opt-in native-FP4 class through its own default pick (solution_id = -2 / -4 / -3: MXFP8 / MXFP6 / MXFP4 activations) and an explicit hipBLASLt
bf16 GEMM on a dense weight of the same shape (the reference's comparator, matmul/rocm/matmul_hipblaslt.cc:103-123), all
measured in one child process with tools/benchlib.py.

Multi-GPU: the op is a single-GPU primitive with no exchange step:
"replicas only" -- every rank runs the same workload on its own weights, no data-path
collective; value = (bytes all ranks streamed) / (max-over-ranks time); scaling "weak".

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT / "petit-kernel_amd"))
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 achievable
M, N, K = 1, 8192, 8192
GROUP = 16


def algorithmic_bytes(m: int, n: int, k: int, group: int) -> int:
    """every operand counted once."""
    return n * k // 2 + n * k // group + 2 * m * k + 2 * m * n + 4


def make_inputs(seed: int, copies: int):
    """CPU-generated (deterministic across boxes), then moved to the device."""
    g = torch.Generator().manual_seed(seed)
    a = torch.randn((M, K), generator=g, dtype=torch.float32).bfloat16()
    gs = torch.rand((1,), generator=g, dtype=torch.float32) * 1.5 + 0.5
    qs, ss = [], []
    for _ in range(copies):
        qs.append(torch.randint(0, 256, (N, K // 2), generator=g, dtype=torch.uint8))
        ss.append((torch.rand((N, K // GROUP), generator=g) * 3.5 + 0.25).to(torch.float8_e4m3fn))
    return a, gs, qs, ss


def cpu_baseline(a, gs, q, s, budget_s: float = 10.0):
    """The oracle ("port" of tests/ops/test_fp4_gemm_quark.py:9-24: LUT dequant + f32 matmul)
    timed on this box's host cores on the SAME workload (one full M=1 8192x8192 call per rep)."""
    from oracle import oracle as O
    a_bits = a.view(torch.int16).numpy().view(np.uint16)
    qn = q.numpy()
    sn = s.view(torch.uint8).numpy()
    O.fp4_gemm_cpu(a_bits, True, qn, sn, float(gs), "nvfp4")  # warm-up (page-in, thread pool)
    times = []
    t_end = time.perf_counter() + budget_s
    while len(times) < 3 or (time.perf_counter() < t_end and len(times) < 200):
        t0 = time.perf_counter()
        O.fp4_gemm_cpu(a_bits, True, qn, sn, float(gs), "nvfp4")
        times.append(time.perf_counter() - t0)
    t = float(np.median(times))
    return {
        "value": algorithmic_bytes(M, N, K, GROUP) / t / 1e9,
        "unit": "GB/s",
        "cores": O.num_threads(),
        "kind": "port",
        "sample": f"{len(times)} full calls of the same workload (M={M} N={N} K={K}), median {t * 1e3:.1f} ms/call, "
                  f"C oracle (LUT dequant + f32-weight matmul), OpenMP {O.num_threads()} threads of {os.cpu_count()} cpus",
        "ms_per_call": t * 1e3,
    }


def measure_cells(dev, stream, budget_s: float, sink=None, verbose: bool = False) -> dict:
    """The whole metric, cell list shared with the parity test (tools/benchlib.py bench_cell_plan): M in {1, 4, 8, 16, 512} x
    the four Llama-3-70B linears for bf16 x NVFP4, fp16 x NVFP4 (the reference benchmark's default dtype) at M = 16 / 512,
    fp16 x MXFP4 at M = 1 / 16, bf16 x MXFP4 at M = 512, the native-FP4 class through ITS default pick (solution_id -2 / -3:
    what an opted-in caller gets, quantiser launch included) and hipBLASLt bf16 on a dense weight.
    Every finished cell goes to the parent as one JSON object (with us_min, the kernel id and its description: the parent writes
    those to the side file gpurun_out/bench_cells_full.json); the parent prints compact rows (main())."""
    import benchlib as BL
    from petit_kernel import _lib
    t0 = time.time()
    notes = []

    class Cells(list):
        def append(self, cell):          # every finished cell goes to the parent at once (a GPU fault kills this process)
            super().append(cell)
            if sink is not None:
                sink.write(json.dumps(cell) + "\n")
                sink.flush()
    cells = Cells()
    mode_sid = {"auto": _lib.PETIT_SOLUTION_AUTO, "native_mxfp8": _lib.PETIT_SOLUTION_AUTO_NATIVE_MXFP8,
                "native_mxfp6": _lib.PETIT_SOLUTION_AUTO_NATIVE_MXFP6, "native_mxfp4": _lib.PETIT_SOLUTION_AUTO_NATIVE_MXFP4}
    peak = {"auto": BL.BF16_PEAK_TFLOPS, "native_mxfp8": BL.FP8_PEAK_TFLOPS, "native_mxfp6": BL.FP4_PEAK_TFLOPS, "native_mxfp4": BL.FP4_PEAK_TFLOPS, "hipblaslt": BL.BF16_PEAK_TFLOPS}
    weights = {}                         # one rotating weight set alive at a time: (shape, w)
    for cell in BL.bench_cell_plan():
        shape, m, a, w, mode = cell["shape"], cell["M"], cell["a"], cell["w"], cell["mode"]
        if time.time() - t0 > budget_s:
            notes.append(f"time budget reached before {shape} M={m} {a}x{w} {mode}")
            break
        dtype = torch.bfloat16 if a == "bf16" else torch.float16
        hbm = m <= BL.HBM_BOUND_MAX_M
        out = {"shape": shape, "M": m, "dt": f"{a}x{w}" + ("" if mode == "auto" else f" {mode}")}
        if shape in BL.ALL_SHAPES:
            n, k = BL.ALL_SHAPES[shape]
        try:
            if shape == "tp8_qkv_3x1280":   # three TP-8 q / k / v shards sharing the activation row(s): three launches vs one grouped launch
                key = ("grp", m)
                if key not in weights:
                    weights.clear()
                    torch.cuda.empty_cache()
                    weights[key] = BL.GroupedGemm(w, 3, 1280, 8192, m, dtype, dev)
                if verbose:
                    print(f"[bench] cell {shape} M={m} {mode}", file=sys.stderr, flush=True)
                r = weights[key].time(mode, stream)
                out.update({"us": round(r["us"], 2), "us_min": round(r["us_min"], 2), "GBs": round(r["gbs"]), "frac": round(r["gbs"] / BL.HBM_PEAK_GBS, 4)})
            elif shape == "tp8_layer":   # one decode layer's four GEMM launches at TP = 8 as a unit (BL.DecodeLayerTP8)
                key = ("tp8_layer", m)
                if key not in weights:
                    weights.clear()
                    torch.cuda.empty_cache()
                    weights[key] = BL.DecodeLayerTP8(m, dev)
                if verbose:
                    print(f"[bench] cell tp8_layer M={m}", file=sys.stderr, flush=True)
                r = weights[key].time(stream)
                out.update({"us": round(r["us"], 2), "us_min": round(r["us_min"], 2), "GBs": round(r["gbs"]), "frac": round(r["gbs"] / BL.HBM_PEAK_GBS, 4)})
            elif shape == "mlp":      # gate_up -> SiLU-mul -> down of Llama-3-70B as one unit (BL.MlpBlock)
                if "mlp" not in weights:
                    weights.clear()
                    torch.cuda.empty_cache()
                    weights["mlp"] = BL.MlpBlock(m, dev)
                blk = weights["mlp"]
                if verbose:
                    print(f"[bench] cell mlp M={m} {mode}", file=sys.stderr, flush=True)
                r = blk.time(mode[4:], stream)
                pk_peak = BL.BF16_PEAK_TFLOPS if mode == "mlp_exact" else BL.FP8_PEAK_TFLOPS if "mxfp8" in mode else BL.FP4_PEAK_TFLOPS
                out.update({"us": round(r["us"], 2), "us_min": round(r["us_min"], 2), "TF": round(r["tflops"], 1), "frac": round(r["tflops"] / pk_peak, 4)})
            elif mode in ("hipblaslt", "hipblaslt_fp8"):
                weights.clear()
                torch.cuda.empty_cache()
                hb = BL.HipblasLtGemm(m, n, k, torch.float8_e4m3fn if mode == "hipblaslt_fp8" else dtype, dev)
                hb.check()
                rr = hb.time(stream, reps=5)
                hb_peak = BL.FP8_PEAK_TFLOPS if mode == "hipblaslt_fp8" else BL.BF16_PEAK_TFLOPS
                out.update({"us": round(rr["us"], 2), "us_min": round(rr["us_min"], 2), "TF": round(rr["tflops"], 1), "frac": round(rr["tflops"] / hb_peak, 4)})
                # ... and the fastest of the heuristic's results (the reference's `-algo tune`): its own cell, "<mode>_best", right behind the first choice's
                rb = hb.time_best(stream, rr)
                hb.close()
                del hb
                cells.append(out)
                out = {"shape": shape, "M": m, "dt": f"{a}x{w} {mode}_best", "us": round(rb["us"], 2), "us_min": round(rb["us_min"], 2), "TF": round(rb["tflops"], 1),
                       "frac": round(rb["tflops"] / hb_peak, 4), "algo_index": rb["algo_index"], "algos_timed": rb["algos_timed"], "algos_found": rb["algos_found"]}
            else:
                if (shape, w) not in weights:
                    weights.clear()
                    torch.cuda.empty_cache()
                    weights[(shape, w)] = BL.Weights(w, n, k, 1280, dev)
                g = BL.Gemm(weights[(shape, w)], m, dtype, dev)
                sid = g.resolve(mode_sid[mode])
                if verbose:
                    print(f"[bench] cell {shape} M={m} {a}x{w} {mode}: 0x{sid:x} {_lib.describe_solution(sid)}", file=sys.stderr, flush=True)
                r = g.time(mode_sid[mode], stream, reps=7 if hbm else 5)
                out.update({"us": round(r["us"], 2), "us_min": round(r["us_min"], 2)})
                if hbm:
                    out.update({"GBs": round(r["gbs"]), "frac": round(r["gbs"] / BL.HBM_PEAK_GBS, 4)})
                else:
                    out.update({"TF": round(r["tflops"], 1), "frac": round(r["tflops"] / peak[mode], 4)})
                out["sid"] = f"{sid:x}"
                out["kernel"] = _lib.describe_solution(sid)
            cells.append(out)
        except Exception as exc:  # noqa: BLE001 -- one cell (e.g. the vendor comparator) must never take the table down
            notes.append(f"{shape} M={m} {a}x{w} {mode} failed: {exc}")
        torch.cuda.empty_cache()
    return {"cells": cells, "cells_notes": notes, "cells_seconds": round(time.time() - t0, 1),
            "cells_method": "tools/benchlib.py: HIP-graph replay, weights rotated over >= 1.28 GB, >= 20 ms warm-up, median of 5-7 replays; "
                            "rate = GB/s (M <= 64) or TFLOP/s; frac = rate / 8000 GB/s, or / 2500 TFLOP/s (native_mxfp8 / 5000, native_mxfp6 and native_mxfp4 / 10000, "
                            "both launches timed); hipblaslt = vendor dense bf16 GEMM (HIPBLAS_COMPUTE_32F, TRANSA=T, first heuristic algorithm), hipblaslt_fp8 = its e4m3 x e4m3 -> bf16 GEMM (/ 5000), *_best = the fastest of up to 24 heuristic results, each timed (the reference's -algo tune); "
                            "us_min, kernel id and description per cell: gpurun_out/bench_cells_full.json"}


COPY_CEILING_GBS = 6290.0      # the guide's measured HBM copy ceiling (MI355X_MICROARCH.md): what a pure stream reaches of the 8 TB/s spec


def compact_cells(cells: list) -> dict:
    """The line's form of the cell table (the driver's record keeps ~8 KB of it): per (shape, dtypes mode) ONE row [shape, "dtypes mode", [us per M],
    [rate per M]] over the M list `cells_m["dtypes mode"]` (rate = GB/s up to M = 64, TFLOP/s above; the fraction of the roofline is rate / the peak
    named in cells_method), and the metric's own 16 cells (bf16 x NVFP4, M in {1, 8, 16, 512}, the four Llama-3-70B linears) once more in full as
    `metric_cells` -- printed LAST, so that a record which keeps only the tail of stdout keeps them."""
    grouped, ms, metric = {}, {}, []
    for c in cells:
        rate = c.get("GBs", c.get("TF"))
        ms.setdefault(c["dt"], [])
        if c["M"] not in ms[c["dt"]]:
            ms[c["dt"]].append(c["M"])
        grouped.setdefault((c["shape"], c["dt"]), {})[c["M"]] = (round(c["us"], 1 if c["us"] >= 100 else 2), int(round(rate)))   # (the side file keeps every digit)
        if c["dt"] == "bf16xnv" and c["M"] in (1, 8, 16, 512) and c["shape"] in ("qkv", "o", "gate_up", "down"):
            row = [c["shape"], c["M"], c["us"], rate, round(c["frac"], 3)]
            if "GBs" in c:
                row.append(round(c["GBs"] / COPY_CEILING_GBS, 3))
            metric.append(row)
    rows = []
    for (shape, dt), by_m in grouped.items():
        rows.append([shape, dt, [by_m[m][0] if m in by_m else None for m in ms[dt]], [by_m[m][1] if m in by_m else None for m in ms[dt]]])
    return {"cells_cols": ["shape", "dtypes mode", "us per M of cells_m[dtypes mode]", "GB/s (M<=64) | TFLOP/s per M"], "cells_m": ms, "cells": rows,
            "metric_cells_cols": ["shape", "M", "us", "GB/s (M<=16) | TFLOP/s", "frac of 8 TB/s | 2.5 PFLOP/s",
                                  "frac of the 6.29 TB/s copy ceiling (M<=16)"],
            "metric_cells": metric}


def host_overhead(step, n_calls: int = 3000) -> dict:
    """Eager host cost per call of the Python surface (graph replay hides it): wall time of n_calls back-to-back
    enqueues divided by n_calls, while the GPU queue is never empty (so it is the HOST that is timed when the
    kernel is shorter than the call; otherwise the kernel time shows)."""
    for i in range(200):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n_calls):
        step(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return {"enqueue_us_per_call": (t1 - t0) / n_calls * 1e6, "wall_us_per_call_incl_drain": (t2 - t0) / n_calls * 1e6,
            "calls": n_calls}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=1000)
    ap.add_argument("--repeats", type=int, default=11, help="timed K-step regions; the median is reported")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a HIP graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cells", action="store_true", help="headline only (skip the M x shape table)")
    ap.add_argument("--no-host-overhead", action="store_true", help="skip the eager host-cost measurement (profiling runs)")
    ap.add_argument("--cells-budget-s", type=float, default=420.0)
    ap.add_argument("--verbose", action="store_true", help="one stderr line per cell (the kernel each call resolved to)")
    ap.add_argument("--cells-child", default="", help=argparse.SUPPRESS)   # internal: run the cell table, append JSON lines to this file
    ap.add_argument("--rotate-mb", type=int, default=1280,
                    help="rotate over at least this many MB of distinct weights; measured on MI355X: per-launch time "
                         "keeps rising until ~1.3 GB (8.3 us at 40 MB, 8.7 at 320 MB, 9.2 at >= 1.3 GB), i.e. the 256 MB "
                         "Infinity Cache still serves part of a 320 MB rotation")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if args.cells_child:
        torch.cuda.set_device(0)
        with open(args.cells_child, "a") as sink:
            meta = measure_cells(torch.device("cuda", 0), torch.cuda.Stream(), args.cells_budget_s, sink, args.verbose)
            meta.pop("cells")
            sink.write(json.dumps({"_meta": meta}) + "\n")
        return
    # one process per GPU; more ranks than GPUs (only useful to smoke-test this flow on a 1-GPU box) wrap around
    dev_index = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" is RCCL on ROCm; there is no data-path collective, it only carries the barrier and the
        # max-over-ranks of the timings.  $PETIT_BENCH_DIST_BACKEND=gloo lets two ranks share one GPU in a smoke run.
        backend = os.environ.get("PETIT_BENCH_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    if not (ROOT / "petit-kernel_amd" / "lib" / "libpetit_amd.so").exists() and rank == 0:
        import __graft_entry__            # a checkout without built artefacts: build once (hipcc), never fall back
        __graft_entry__.build()
    if world > 1:
        dist.barrier()
    import petit_kernel  # fails loudly when libpetit_amd.so is missing

    bytes_per_step = algorithmic_bytes(M, N, K, GROUP)
    copies = (args.rotate_mb * 1024 * 1024) // bytes_per_step + 2
    a, gs, qs, ss = make_inputs(1234 + rank, copies)
    a_d = a.to(dev)
    gs_d = gs.to(dev)
    packed = []
    for q, s in zip(qs, ss):
        b = petit_kernel.repack_nvfp4(q.to(dev).view(torch.int32), N, K)
        sp = petit_kernel.process_nvfp4_scales(s.to(dev), N, K)
        packed.append((b, sp))
    torch.cuda.synchronize()

    def step(i: int):
        b, sp = packed[i % copies]
        return petit_kernel.mul_nvfp4_a16(a_d, b, sp, gs_d, M, N, K, -1)

    def barrier():
        if world > 1:
            dist.barrier()

    stream = torch.cuda.Stream(dev)
    warmup_done = 0
    region_ms, region_wall_ms = [], []
    with torch.cuda.stream(stream):
        out = step(0)                              # first call: module load, arch table
        stream.synchronize()
        graph = None
        if not args.no_graph:
            try:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=stream):
                    for i in range(args.steps):
                        out = step(i)
            except Exception as exc:               # capture unsupported: time eager launches
                print(f"[bench] graph capture failed ({exc}); timing eager launches", file=sys.stderr)
                graph = None
        # untimed warm-up: the same steps, so clocks and caches are in steady state: --warmup steps, and at least
        # ~20 ms of them (measured on MI355X: a 400-step replay timed after only 400 warm-up steps reads 10.4 us/step,
        # after 2000 warm-up steps 9.2 -- DVFS ramp)
        t_w = time.perf_counter()
        while warmup_done < args.warmup or time.perf_counter() - t_w < 0.03:
            if graph is not None:
                graph.replay()
            else:
                for i in range(args.steps):
                    out = step(i)
            warmup_done += args.steps
            stream.synchronize()

        # events are recorded on the stream the kernels run on; every timed region is exactly --steps steps,
        # bracketed by barrier + synchronize on both sides
        for _ in range(max(1, args.repeats)):
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e0.record(stream)
            if graph is not None:
                graph.replay()                         # exactly args.steps steps
            else:
                for i in range(args.steps):
                    out = step(i)
            e1.record(stream)
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            barrier()
            region_ms.append(e0.elapsed_time(e1))
            region_wall_ms.append(wall * 1e3)
    # median region; whole-job time = slowest rank (petit_kernel/replicas.py; covered on CPU with gloo)
    order = sorted(range(len(region_ms)), key=lambda i: region_ms[i])
    mid = order[len(order) // 2]
    ev_ms, wall_ms = region_ms[mid], region_wall_ms[mid]
    from petit_kernel import replicas
    ev_ms_max = replicas.max_over_ranks(ev_ms, dev)
    wall_ms_max = replicas.max_over_ranks(wall_ms, dev)
    ms_per_step = ev_ms_max / args.steps

    # sanity: the timed kernel really computes the GEMM (checked against the oracle in smoke()/tests)
    assert out.shape == (M, N) and torch.isfinite(out.float()).all()

    if rank == 0:
        # HBM bytes per launch from the PMC passes of the same command (tools/collect_profiles.sh,
        # FETCH_SIZE x2 gfx950 correction + WRITE_SIZE); rocprofv3 cannot run inside this process
        traffic, traffic_src = None, None
        for cand in sorted((ROOT / "profiles").glob("r*_pmc_traffic.json"), reverse=True):
            try:
                traffic = json.loads(cand.read_text())["traffic_bytes_per_launch"]
                traffic_src = f"from profiles/{cand.name} (separate rocprofv3 --pmc passes of this command; not measured in this run)"
                break
            except Exception:
                pass
        per_gpu_gbs = bytes_per_step / (ms_per_step * 1e-3) / 1e9
        from petit_kernel import _lib
        hints = _lib.SolutionHints(_lib.CXX_DTYPE_BF16, _lib.CXX_DTYPE_FP4_E2M1, _lib.CXX_DTYPE_BF16, 0)
        sid = _lib.lib.petit_gemm_default_solution(C.byref(hints), M, N, K)
        line = {
            "metric": "bf16xnvfp4_gemm_achieved_hbm_bandwidth_m1_n8192_k8192",
            "value": per_gpu_gbs * world,
            "unit": "GB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16",
            "data": "synthetic",
            "config": {"workload": "M=1 N=8192 K=8192 bf16 x nvfp4 (e4m3 scales, g=16), one mul_nvfp4_a16 call per step",
                       "weights_rotated_over_copies": copies, "parallelism": f"replicas x{world} (no data-path collective)",
                       "launch": "hip graph replay" if graph is not None else "eager",
                       "solution": f"0x{sid:x} {_lib.describe_solution(sid)}",
                       "timed_regions": len(region_ms), "statistic": "median region",
                       "warmup_steps_run": warmup_done},
            "tflops": 2.0 * M * N * K / (ms_per_step * 1e-3) / 1e12 * world,
            "wall_ms_per_step": wall_ms_max / args.steps,
            "ms_per_step_regions": {"min": min(region_ms) / args.steps, "median": ev_ms / args.steps,
                                    "max": max(region_ms) / args.steps},
            "roofline": {
                "bound": "hbm",
                "achieved": per_gpu_gbs,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": per_gpu_gbs / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "kernel": "petit_amd::gemm_decode_kernel" if (sid >> 48) & 0xF in (4, 14, 15) else "petit_amd::gemm_stream_kernel",
                "bytes_per_launch": bytes_per_step,
                "us_per_launch": ms_per_step * 1e3,
            },
        }
        print(f"[bench] headline {ms_per_step * 1e3:.3f} us/step", file=sys.stderr, flush=True)
        if world == 1 and not args.no_host_overhead:
            # eager host cost per call of the two operator layers, on a TINY problem (M=1, N=256, K=1024: a ~2 us kernel), so
            # that the host -- not the GPU -- is what the loop waits for (under graph replay, above, the host is out of the loop)
            from petit_kernel import compiled, ops
            tn, tk = 256, 1024
            tq = torch.randint(0, 256, (tn, tk // 2), dtype=torch.uint8, device=dev)
            tb = ops.repack_nvfp4(tq.view(torch.int32), tn, tk)
            ts = ops.process_nvfp4_scales((torch.rand((tn, tk // 16), device=dev) * 3.5 + 0.25).to(torch.float8_e4m3fn), tn, tk)
            ta = torch.randn((1, tk), device=dev).bfloat16()
            with torch.cuda.stream(stream):
                line["host_us_per_call"] = {"problem": f"M=1 N={tn} K={tk} bf16 x nvfp4, solution_id=-1, eager, one stream",
                                            "ctypes": host_overhead(lambda i: ops.mul_nvfp4_a16(ta, tb, ts, gs_d, 1, tn, tk, -1))}
                if compiled.available():
                    line["host_us_per_call"]["compiled_torch_library_op"] = host_overhead(
                        lambda i: compiled.mul_nvfp4_a16(ta, tb, ts, gs_d, 1, tn, tk, -1))
                else:
                    line["host_us_per_call"]["compiled_torch_library_op"] = f"unavailable: {compiled.why_unavailable()}"
                line["host_us_per_call"]["package_default"] = "compiled" if petit_kernel._impl is compiled else "ctypes"
        del packed
        torch.cuda.empty_cache()
        full_cells = None
        if world == 1 and not args.no_cells:
            # the table runs in a CHILD process that reports every finished cell at once: a fault in any one kernel (or in
            # the vendor comparator) costs that cell, never the headline line or the cells already measured
            import subprocess
            import tempfile
            with tempfile.NamedTemporaryFile("r", suffix=".jsonl", dir=str(ROOT / "gpurun_out") if (ROOT / "gpurun_out").is_dir() else None) as tf:
                try:
                    rc = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--cells-child", tf.name, "--cells-budget-s",
                                         str(args.cells_budget_s)] + (["--verbose"] if args.verbose else []),
                                        timeout=args.cells_budget_s + 90, stdout=subprocess.DEVNULL).returncode
                except subprocess.TimeoutExpired:
                    rc = "timeout"
                cells, meta = [], {}
                for ln in Path(tf.name).read_text().splitlines():
                    rec = json.loads(ln)
                    if "_meta" in rec:
                        meta = rec["_meta"]
                    else:
                        cells.append(rec)
                line.update(meta)
                if rc != 0:
                    line["cells_error"] = f"cell process ended with {rc} after {len(cells)} cells (see stderr)"
                full_cells = cells
                try:  # everything a cell measured (us_min, kernel id + description): a side file, not the line
                    side = ROOT / "gpurun_out" / "bench_cells_full.json"
                    side.parent.mkdir(exist_ok=True)
                    side.write_text(json.dumps({"steps": args.steps, "cells": cells}, indent=0))
                except OSError:
                    pass
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(a, gs, qs[0], ss[0])
        if full_cells is not None:
            line.update(compact_cells(full_cells))   # LAST: the metric's own cells end the line
        print(json.dumps(line, separators=(",", ":")), flush=True)

    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
