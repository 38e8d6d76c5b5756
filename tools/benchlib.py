"""tools/benchlib.py -- measurement plumbing shared by bench.py and tools/tune.py (not part of the product).

Method, the same for every number this repo reports:
  * launches are captured into a HIP graph and replayed, so the host is out of the loop; events are recorded on the
    launch stream;
  * every launch reads a DIFFERENT copy of the weights, rotating over >= ~1.3 GB, so nothing is served by the 256 MB
    Infinity Cache (the reference's benchmark reuses one buffer, tools/benchmarks/matmul/rocm/matmul_petit.cc:116-132);
  * >= 20 ms of replays before timing (DVFS ramp), then `reps` timed replays, MEDIAN reported.
Replaces the timing core of the reference's tools/benchmarks/matmul/main.cc:230-325.
"""
from __future__ import annotations

import ctypes as C
import subprocess
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "petit-kernel_amd"))

from petit_kernel import _lib  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 achievable
BF16_PEAK_TFLOPS = 2500.0    # dense bf16/fp16 MFMA
FP8_PEAK_TFLOPS = 5000.0     # dense fp8 MFMA (what an FP4 x FP8 block-scaled MFMA runs at)
FP4_PEAK_TFLOPS = 10000.0    # dense fp4 MFMA

LLAMA70B = {"qkv": (10240, 8192), "o": (8192, 8192), "gate_up": (57344, 8192), "down": (8192, 28672),
            "sq4096": (4096, 4096), "sq8192": (8192, 8192)}


# The cell table of bench.py (the rest of BASELINE.json's metric: "TFLOPS + achieved HBM GB/s, M in {1, 8, 16, 512}, Llama-70B
# shapes" + configs[2] M = 4, configs[3] fp16 x MXFP4, configs[4] native FP4 / hipBLASLt, and the reference benchmark's own default
# dtype, fp16 x NVFP4, tools/benchmarks/matmul.py:92-127), in the order bench.py measures it.  tests/test_gpu_parity.py::
# test_bench_cells_parity iterates the SAME list, so every (shape, M, dtype, mode) that is timed is also checked against the oracle.
#   mode: "auto" = solution_id -1; "native_mxfp8" / "native_mxfp6" / "native_mxfp4" = the opt-in native class through its own default pick;
#   "hipblaslt" = the vendor's dense 16-bit GEMM on a dense weight of the same shape (comparator, no parity to check);
#   "hipblaslt_fp8" = the vendor's FP8 (e4m3 x e4m3 -> bf16) GEMM on the same shape: what the native class competes with
SHAPE_ORDER = ("qkv", "o", "gate_up", "down")
# The deployment the reference's README describes -- Llama-3.3-70B on 8 GPUs, TP = 8 -- in the shard shapes of its own benchmark list
# (tools/benchmarks/matmul.py:18-33): what ONE GPU runs per layer (round 6).  `o` has K = 1024: one span of 8 k-tiles.
TP8 = {"tp8_qkv": (1280, 8192), "tp8_o": (8192, 1024), "tp8_gate_up": (7168, 8192), "tp8_down": (8192, 3584)}
TP8_ORDER = tuple(TP8)
TP8_MS = (1, 16, 64, 512)
ALL_SHAPES = {**LLAMA70B, **TP8}


MID_MS = (32, 44, 64, 128)                  # batched decode: between the decode kernels and the large-M tiles (tools/benchmarks/matmul.py:8-90 lists 15, 44, ...)
PREFILL_MS = (1024, 2084, 4314, 16375)       # prefill chunks; the last three are entries of the reference's own list (ragged on purpose)
HBM_BOUND_MAX_M = 64                         # cells up to this M are reported as GB/s against the HBM roofline (intensity at M = 64: 228 FLOP/B, ridge 315), above as TFLOP/s


def bench_cell_plan() -> list:
    plan = []
    # the bandwidth-bound cells of every shape first (a decode cell timed right after seconds of 1.3 kW compute reads 5-10 % slow)
    for shape in SHAPE_ORDER:
        plan += [dict(shape=shape, M=m, a="bf16", w="nv", mode="auto") for m in (1, 4, 8, 16)]
        plan += [dict(shape=shape, M=16, a="fp16", w="nv", mode="auto")]
        plan += [dict(shape=shape, M=m, a="fp16", w="mx", mode="auto") for m in (1, 16)]     # BASELINE configs[3]; plain MXFP4 hints (the kernels test the scale range)
        plan += [dict(shape=shape, M=m, a="bf16", w="mx", mode="auto") for m in (1, 16)]     # the reference's only MX activation type (gemm_fp4_fp16_grid.cc:55-64)
    # mid M (round 5): the everyday batch sizes of a serving engine, 17 <= M <= 128
    for shape in SHAPE_ORDER:
        plan += [dict(shape=shape, M=m, a="bf16", w=w, mode="auto") for w in ("nv", "mx") for m in MID_MS]
    for shape in SHAPE_ORDER:
        # (M = 256 in the reference benchmark's default dtype: the middle column of its ENTRIES, tools/benchmarks/matmul.py:92-117)
        plan += [dict(shape=shape, M=256, a="fp16", w="nv", mode="auto"),
                 dict(shape=shape, M=512, a="bf16", w="nv", mode="auto"), dict(shape=shape, M=512, a="fp16", w="nv", mode="auto"),
                 dict(shape=shape, M=512, a="bf16", w="mx", mode="auto"), dict(shape=shape, M=512, a="fp16", w="mx", mode="auto"),
                 dict(shape=shape, M=512, a="bf16", w="mx", mode="native_mxfp8"), dict(shape=shape, M=512, a="bf16", w="mx", mode="native_mxfp6"),
                 dict(shape=shape, M=512, a="bf16", w="mx", mode="native_mxfp4"),
                 # (round 6) NVFP4 weights on the native class: the MFMA-native image of the same weights (petit_nvfp4_native_image, attached)
                 dict(shape=shape, M=512, a="bf16", w="nv", mode="native_mxfp8"), dict(shape=shape, M=512, a="bf16", w="nv", mode="native_mxfp6"),
                 dict(shape=shape, M=512, a="bf16", w="nv", mode="native_mxfp4"),
                 dict(shape=shape, M=512, a="bf16", w="dense", mode="hipblaslt"), dict(shape=shape, M=512, a="fp8", w="dense", mode="hipblaslt_fp8")]
    # prefill (round 5): M > 512 up to the reference list's largest entries, exact NV / MX, the three native classes, the vendor's bf16 and FP8 GEMMs
    for shape in SHAPE_ORDER:
        for m in PREFILL_MS:
            plan += [dict(shape=shape, M=m, a="bf16", w="nv", mode="auto"), dict(shape=shape, M=m, a="bf16", w="mx", mode="auto"),
                     dict(shape=shape, M=m, a="bf16", w="mx", mode="native_mxfp8"), dict(shape=shape, M=m, a="bf16", w="mx", mode="native_mxfp6"),
                     dict(shape=shape, M=m, a="bf16", w="mx", mode="native_mxfp4"),
                     dict(shape=shape, M=m, a="bf16", w="nv", mode="native_mxfp8"), dict(shape=shape, M=m, a="bf16", w="nv", mode="native_mxfp6"),
                     dict(shape=shape, M=m, a="bf16", w="nv", mode="native_mxfp4"),
                     dict(shape=shape, M=m, a="bf16", w="dense", mode="hipblaslt"), dict(shape=shape, M=m, a="fp8", w="dense", mode="hipblaslt_fp8")]
        # (fp16 x MXFP4 at the largest chunk: the family x regime whose table rows named a 10 x slower kernel until round 5 re-measured them -- no cell had timed it)
        plan += [dict(shape=shape, M=PREFILL_MS[-1], a="fp16", w="mx", mode="auto")]
    # TP = 8 (round 6): the four shard shapes at decode and small-batch M, and ONE decode layer's four launches as a unit (DecodeLayerTP8)
    for shape in TP8_ORDER:
        plan += [dict(shape=shape, M=m, a="bf16", w="nv", mode="auto") for m in TP8_MS]
    plan += [dict(shape="tp8_layer", M=m, a="bf16", w="nv", mode="layer") for m in (1, 16)]
    # launch-gap-bound shapes: the q / k / v shards of a TP-8 deployment (1280 x 8192 each) as three launches and as one grouped launch
    plan += [dict(shape="tp8_qkv_3x1280", M=m, a="bf16", w="nv", mode=mode) for m in (1, 16) for mode in ("separate", "grouped")]
    # the gated-MLP block (gate_up -> SiLU-mul -> down) as a unit: what the quantising epilogue buys the native class
    plan += [dict(shape="mlp", M=512, a="bf16", w="mx", mode="mlp_" + mode) for mode in MlpBlock.MODES]
    return plan


def alg_bytes(m: int, n: int, k: int, group: int) -> int:
    """every operand counted once."""
    return n * k // 2 + n * k // group + 2 * m * k + 2 * m * n + 4


def time_graph(launch, launches: int, reps: int, stream, warm_s: float = 0.02) -> list:
    """Capture `launches` calls of launch(i), replay; returns us per launch for each of `reps` timed replays."""
    with torch.cuda.stream(stream):
        launch(0)
        stream.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            for i in range(launches):
                launch(i)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < warm_s:
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


def median(xs):
    s = sorted(xs)
    return s[len(s) // 2]


class Weights:
    """`copies` distinct packed (W, scales) pairs of one shape, random bytes generated on the device (packed tensors
    are opaque: any bytes are a valid weight matrix; scales are drawn valid: e4m3 in [0.25, 3.75], e8m0 in 119..135)."""

    def __init__(self, fmt: str, n: int, k: int, rotate_mb: int, dev, seed: int = 1234, max_copies: int = 64):
        self.fmt, self.n, self.k = fmt, n, k
        self.group = 16 if fmt == "nv" else 32
        wbytes = n * k // 2 + n * k // self.group
        self.copies = int(max(2, min(max_copies, (rotate_mb << 20) // wbytes + 2))) if rotate_mb else int(max_copies)   # (rotate_mb = 0: exactly max_copies)
        gen = torch.Generator(device=dev).manual_seed(seed)
        self.packed = []
        for _ in range(self.copies):
            b = torch.randint(-2 ** 31, 2 ** 31 - 1, (n // 16, 2 * k), generator=gen, dtype=torch.int32, device=dev)
            if fmt == "nv":
                sp = (torch.rand((n, k // 16), generator=gen, device=dev) * 3.5 + 0.25).to(torch.float8_e4m3fn)
            else:
                sp = torch.randint(119, 136, (n // 32, k), generator=gen, dtype=torch.uint8, device=dev)
            self.packed.append((b, sp))
        self.images = []

    def __getitem__(self, i):
        return self.packed[i % self.copies]

    def attach_native(self):
        """NVFP4 weights on the native class: the MFMA-native image of every copy (petit_nvfp4_native_image: one launch each, load-time work, not
        timed), attached to the copy's packed weight pointer -- what a serving stack does once after repack_nvfp4."""
        if self.fmt != "nv" or self.images:
            return
        nbytes = int(_lib.lib.petit_nvfp4_native_image_bytes(self.k, self.n))
        for b, sp in self.packed:
            img = torch.empty(nbytes, dtype=torch.uint8, device=b.device)
            rc = _lib.lib.petit_nvfp4_native_image(C.c_void_p(img.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(sp.data_ptr()), self.k, self.n,
                                                   C.c_void_p(torch.cuda.current_stream().cuda_stream))
            assert rc == 0, rc
            assert _lib.lib.petit_nvfp4_native_attach(C.c_void_p(b.data_ptr()), C.c_void_p(img.data_ptr())) == 0
            self.images.append(img)
        torch.cuda.synchronize()

    def detach_native(self):
        for (b, _), _img in zip(self.packed, self.images):
            _lib.lib.petit_nvfp4_native_attach(C.c_void_p(b.data_ptr()), None)
        self.images = []

    def __del__(self):
        try:
            self.detach_native()
        except Exception:  # noqa: BLE001 -- interpreter shutdown
            pass


class Gemm:
    """One (weights, M, dtype) problem launched through the C ABI (petit_gemm_*_ws), with a per-problem workspace."""

    def __init__(self, weights: Weights, m: int, dtype, dev, seed: int = 7):
        self.w, self.m, self.dtype, self.dev = weights, m, dtype, dev
        gen = torch.Generator(device=dev).manual_seed(seed)
        self.a = torch.randn((m, weights.k), generator=gen, device=dev, dtype=torch.float32).to(dtype)
        self.c = torch.empty((m, weights.n), dtype=dtype, device=dev)
        self.gs = torch.tensor([1.0], dtype=torch.float32, device=dev)
        self.a_type = _lib.CXX_DTYPE_BF16 if dtype == torch.bfloat16 else _lib.CXX_DTYPE_FP16
        self.b_type = _lib.CXX_DTYPE_FP4_E2M1 if weights.fmt == "nv" else _lib.CXX_DTYPE_MXFP4_E2M1
        self.hints = _lib.SolutionHints(self.a_type, self.b_type, self.a_type, 0)
        self.fn = _lib.lib.petit_gemm_fp4_fp16_grid_ws if weights.fmt == "nv" else _lib.lib.petit_gemm_mxfp4_fp16_grid_ws
        self._ws = {}

    def default_solution(self) -> int:
        return int(_lib.lib.petit_gemm_default_solution(C.byref(self.hints), self.m, self.w.n, self.w.k))

    def resolve(self, sid: int, workspace_bytes: int = 1 << 62) -> int:
        """the concrete kernel a call with `sid` (an AUTO sentinel or an explicit id) runs, given that much scratch"""
        return int(_lib.lib.petit_gemm_resolve_solution(C.byref(self.hints), self.m, self.w.n, self.w.k, C.c_uint64(sid), None,
                                                        C.c_uint64(workspace_bytes)))

    def solutions(self) -> list:
        count = C.c_uint(0)
        _lib.lib.petit_gemm_get_solutions(C.byref(self.hints), self.m, self.w.n, self.w.k, None, C.byref(count))
        buf = (C.c_uint64 * max(count.value, 1))()
        _lib.lib.petit_gemm_get_solutions(C.byref(self.hints), self.m, self.w.n, self.w.k, buf, C.byref(count))
        return [int(buf[i]) for i in range(count.value)]

    def workspace(self, sid: int):
        need = int(_lib.lib.petit_gemm_workspace_bytes(C.byref(self.hints), self.m, self.w.n, self.w.k, C.c_uint64(sid)))
        if need == 0:
            return None, 0
        if need not in self._ws:
            self._ws[need] = torch.empty(need, dtype=torch.uint8, device=self.dev)
        return self._ws[need], need

    def launcher(self, sid: int):
        if self.w.fmt == "nv" and sid in (_lib.PETIT_SOLUTION_AUTO_NATIVE_MXFP8, _lib.PETIT_SOLUTION_AUTO_NATIVE_MXFP6, _lib.PETIT_SOLUTION_AUTO_NATIVE_MXFP4):
            self.w.attach_native()
        ws, need = self.workspace(sid)
        wsp = C.c_void_p(ws.data_ptr()) if ws is not None else None
        m, n, k = self.m, self.w.n, self.w.k

        def launch(i):
            b, sp = self.w[i]
            rc = self.fn(C.c_void_p(self.c.data_ptr()), C.c_void_p(self.a.data_ptr()), C.c_void_p(b.data_ptr()),
                         C.c_void_p(sp.data_ptr()), C.c_void_p(self.gs.data_ptr()), m, n, k, C.byref(self.hints),
                         C.c_uint64(sid), None, wsp, C.c_uint64(need), C.c_void_p(torch.cuda.current_stream().cuda_stream))
            if rc != 0:
                raise RuntimeError(f"rc={rc} ({_lib.error_string(rc)}) for solution 0x{sid:x}")
        return launch

    def time(self, sid: int, stream, reps: int = 7, launches: int = 0) -> dict:
        nbytes = alg_bytes(self.m, self.w.n, self.w.k, self.w.group)
        ideal_us = max(nbytes / (HBM_PEAK_GBS * 1e3), 2.0 * self.m * self.w.n * self.w.k / (BF16_PEAK_TFLOPS * 1e6))
        launches = launches or int(max(10, min(400, 3000.0 / max(ideal_us, 1.0))))
        us = time_graph(self.launcher(sid), launches, reps, stream)
        med = median(us)
        return {"us": med, "us_min": min(us), "us_max": max(us), "launches": launches, "reps": reps,
                "gbs": nbytes / med / 1e3, "tflops": 2.0 * self.m * self.w.n * self.w.k / med / 1e6, "bytes": nbytes}


class GroupedGemm:
    """`count` weight matrices [n, k] sharing one activation block, as `count` launches (solution_id -1 each) or ONE grouped launch
    (petit_gemm_fp4_fp16_grouped); every launch of the timed graph reads different weight copies."""

    def __init__(self, fmt: str, count: int, n: int, k: int, m: int, dtype, dev, rotate_mb: int = 1280):
        self.count, self.n, self.k, self.m, self.dev = count, n, k, m, dev
        self.w = Weights(fmt, n, k, rotate_mb, dev, max_copies=192)
        self.g = Gemm(self.w, m, dtype, dev)
        self.cs = [torch.empty((m, n), dtype=dtype, device=dev) for _ in range(count)]
        self.bytes = count * (n * k // 2 + n * k // self.w.group + 2 * m * n) + 2 * m * k

    def launch(self, mode: str, i: int):
        g, stream = self.g, C.c_void_p(torch.cuda.current_stream().cuda_stream)
        if mode == "separate":
            for j in range(self.count):
                b, sp = self.w[i * self.count + j]
                rc = g.fn(C.c_void_p(self.cs[j].data_ptr()), C.c_void_p(g.a.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(sp.data_ptr()),
                          C.c_void_p(g.gs.data_ptr()), self.m, self.n, self.k, C.byref(g.hints), C.c_uint64(_lib.PETIT_SOLUTION_AUTO), None, None,
                          C.c_uint64(0), stream)
                assert rc == 0, rc
        else:
            arr = (_lib.GroupMember * self.count)()
            for j in range(self.count):
                b, sp = self.w[i * self.count + j]
                arr[j] = _lib.GroupMember(self.cs[j].data_ptr(), b.data_ptr(), sp.data_ptr(), g.gs.data_ptr(), None, self.n, 0)
            rc = _lib.lib.petit_gemm_fp4_fp16_grouped(arr, self.count, C.c_void_p(g.a.data_ptr()), self.m, self.k, C.byref(g.hints),
                                                      C.c_uint64(_lib.PETIT_SOLUTION_AUTO), stream)
            assert rc == 0, rc

    def time(self, mode: str, stream, reps: int = 7, launches: int = 100) -> dict:
        us = time_graph(lambda i: self.launch(mode, i), launches, reps, stream)
        med = median(us)
        return {"us": med, "us_min": min(us), "gbs": self.bytes / med / 1e3}


class DecodeLayerTP8:
    """The four GEMM launches of ONE Llama-3-70B layer on one GPU of a TP = 8 deployment, at M tokens, bf16 x NVFP4, chained the way the layer chains
    them (attention itself is not this library's): q / k / v shards (1024 / 128 / 128 x 8192) as ONE grouped launch -> `o` (8192 x 1024, on the q output:
    a stand-in for the attention output of the same shape) -> gate_up (7168 x 8192) with fused SiLU-mul -> down (8192 x 3584).  Every replay reads
    a different copy of all four weight sets (nothing is served by the Infinity Cache).  Reported: microseconds per layer and the fraction of
    (sum of the algorithmic bytes of the four GEMMs) / 8 TB/s."""

    def __init__(self, m: int, dev, rotate_mb: int = 1280):
        self.m, self.dev = m, dev
        per_layer = sum(n * k * 9 // 16 for n, k in ((1280, 8192), (8192, 1024), (7168, 8192), (8192, 3584)))
        copies = int(max(2, min(32, (rotate_mb << 20) // per_layer + 2)))
        self.wq = Weights("nv", 1024, 8192, 0, dev, seed=21, max_copies=copies)
        self.wk = Weights("nv", 128, 8192, 0, dev, seed=22, max_copies=copies)
        self.wv = Weights("nv", 128, 8192, 0, dev, seed=23, max_copies=copies)
        self.wo = Weights("nv", 8192, 1024, 0, dev, seed=24, max_copies=copies)
        self.wgu = Weights("nv", 7168, 8192, 0, dev, seed=25, max_copies=copies)
        self.wd = Weights("nv", 8192, 3584, 0, dev, seed=26, max_copies=copies)
        for w in (self.wq, self.wk, self.wv, self.wo, self.wgu, self.wd):
            w.copies = copies
        gen = torch.Generator(device=dev).manual_seed(5)
        self.x = torch.randn((m, 8192), generator=gen, device=dev, dtype=torch.float32).bfloat16()
        self.gs = torch.tensor([0.02], dtype=torch.float32, device=dev)
        self.bytes = sum(alg_bytes(m, n, k, 16) for n, k in ((1280, 8192), (8192, 1024), (7168, 8192), (8192, 3584)))
        self.flops = 2.0 * m * sum(n * k for n, k in ((1280, 8192), (8192, 1024), (7168, 8192), (8192, 3584)))

    def run(self, i: int = 0):
        import petit_kernel as pk
        m, gs = self.m, self.gs
        members = [(self.wq[i][0], self.wq[i][1], gs, 1024), (self.wk[i][0], self.wk[i][1], gs, 128), (self.wv[i][0], self.wv[i][1], gs, 128)]
        if m <= 16:
            q, k_, v = pk.mul_fp4_a16_grouped("nvfp4", self.x, members, m, 8192, -1)
        else:
            q, k_, v = (pk.mul_nvfp4_a16(self.x, b, sp, g, m, n, 8192, -1) for b, sp, g, n in members)
        o = pk.mul_nvfp4_a16(q, self.wo[i][0], self.wo[i][1], gs, m, 8192, 1024, -1)
        h = pk.mul_nvfp4_a16(o, self.wgu[i][0], self.wgu[i][1], gs, m, 7168, 8192, -1, activation="silu_mul")
        return q, k_, v, o, h, pk.mul_nvfp4_a16(h, self.wd[i][0], self.wd[i][1], gs, m, 8192, 3584, -1)

    def time(self, stream, reps: int = 7, launches: int = 40) -> dict:
        us = time_graph(lambda i: self.run(i), launches, reps, stream)
        med = median(us)
        return {"us": med, "us_min": min(us), "gbs": self.bytes / med / 1e3, "tflops": self.flops / med / 1e6, "launches": launches, "reps": reps}


class MlpBlock:
    """One gated-MLP block of Llama-3-70B at M tokens, MXFP4 weights: h = silu_mul(x . Wgu^T), y = h . Wd^T, three ways:
      exact                     default kernels (bf16 activations), fused SiLU-mul: 2 launches
      native_mxfp4_4launch      the native class call by call: quantiser + GEMM, twice: 4 launches
      native_mxfp4_pipeline     quantise x once, gate_up emits the quantised h (out_quantized), down reads it: 3 launches
      native_mxfp8_pipeline / native_mxfp6_pipeline   the same with MXFP8 / MXFP6 (e2m3) activations: the deployable accuracy class
    (petit_kernel.mul_mxfp4_native; weights of both GEMMs rotate over copies so that nothing is served by the Infinity Cache)."""

    MODES = ("exact", "native_mxfp4_4launch", "native_mxfp4_pipeline", "native_mxfp8_pipeline", "native_mxfp6_pipeline")

    def __init__(self, m: int, dev, hidden: int = 8192, inter: int = 28672, rotate_mb: int = 1280):
        self.m, self.hidden, self.inter = m, hidden, inter
        self.w1 = Weights("mx", 2 * inter, hidden, rotate_mb // 2, dev, seed=11)
        self.w2 = Weights("mx", hidden, inter, rotate_mb // 2, dev, seed=12)
        gen = torch.Generator(device=dev).manual_seed(3)
        self.x = torch.randn((m, hidden), generator=gen, device=dev, dtype=torch.float32).bfloat16()
        self.gs = torch.tensor([0.02], dtype=torch.float32, device=dev)
        self.flops = 2.0 * m * hidden * (2 * inter) + 2.0 * m * inter * hidden

    def run(self, mode: str, i: int = 0):
        import petit_kernel as pk
        (b1, s1), (b2, s2) = self.w1[i], self.w2[i]
        m, hid, inter, gs = self.m, self.hidden, self.inter, self.gs
        if mode == "exact":
            h = pk.mul_mxfp4_a16(self.x, b1, s1, gs, m, 2 * inter, hid, -1, activation="silu_mul")
            return pk.mul_mxfp4_a16(h, b2, s2, gs, m, hid, inter, -1)
        fmt = "mxfp8" if "mxfp8" in mode else "mxfp6" if "mxfp6" in mode else "mxfp4"
        sid = {"mxfp8": pk.SOLUTION_AUTO_NATIVE_MXFP8, "mxfp6": pk.SOLUTION_AUTO_NATIVE_MXFP6, "mxfp4": pk.SOLUTION_AUTO_NATIVE_MXFP4}[fmt]
        if mode.endswith("4launch"):
            h = pk.mul_mxfp4_native(self.x, b1, s1, gs, m, 2 * inter, hid, sid, activation="silu_mul")
            return pk.mul_mxfp4_native(h, b2, s2, gs, m, hid, inter, sid)
        hq = pk.mul_mxfp4_native(pk.quantize_activations(self.x, fmt), b1, s1, gs, m, 2 * inter, hid, sid, activation="silu_mul", out_quantized=fmt)
        return pk.mul_mxfp4_native(hq, b2, s2, gs, m, hid, inter, sid)

    def time(self, mode: str, stream, reps: int = 5, launches: int = 8) -> dict:
        us = time_graph(lambda i: self.run(mode, i), launches, reps, stream)
        med = median(us)
        return {"us": med, "us_min": min(us), "tflops": self.flops / med / 1e6, "launches": launches, "reps": reps}


# --- the dense 16-bit GEMM comparator: hipBLASLt, explicitly (tools/comparators/hipblaslt_gemm.cc) ---------------------

_HBL_SRC = ROOT / "tools" / "comparators" / "hipblaslt_gemm.cc"
_HBL_LIB = ROOT / "tools" / "comparators" / "libhipblaslt_gemm.so"


def build_hipblaslt_comparator(force: bool = False) -> Path:
    if force or not _HBL_LIB.exists() or _HBL_LIB.stat().st_mtime < _HBL_SRC.stat().st_mtime:
        subprocess.run(["hipcc", "-O2", "-fPIC", "-shared", "--offload-arch=gfx950", str(_HBL_SRC), "-o", str(_HBL_LIB),
                        "-lhipblaslt"], check=True)
    return _HBL_LIB


class HipblasLtGemm:
    """C[m][n] = A[m][k] . W[n][k]^T, 16-bit operands, f32 compute, hipBLASLt's first heuristic algorithm
    (the reference's comparator: tools/benchmarks/matmul/rocm/matmul_hipblaslt.cc:103-123,249-263)."""

    _lib = None

    def __init__(self, m: int, n: int, k: int, dtype, dev, rotate_mb: int = 1280, max_copies: int = 8):
        if HipblasLtGemm._lib is None:
            lib = C.CDLL(str(build_hipblaslt_comparator()))
            lib.hbl_create.restype = C.c_void_p
            lib.hbl_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
            lib.hbl_run.restype = C.c_int
            lib.hbl_run.argtypes = [C.c_void_p] * 5
            lib.hbl_destroy.argtypes = [C.c_void_p]
            lib.hbl_count.restype = C.c_int
            lib.hbl_count.argtypes = [C.c_void_p]
            lib.hbl_select.restype = C.c_int
            lib.hbl_select.argtypes = [C.c_void_p, C.c_int]
            HipblasLtGemm._lib = lib
        self.m, self.n, self.k = m, n, k
        fp8 = dtype == torch.float8_e4m3fn          # (the vendor's 8-bit GEMM: e4m3 operands, unit scales, bf16 output)
        self.h = HipblasLtGemm._lib.hbl_create(m, n, k, 2 if fp8 else int(dtype == torch.bfloat16))
        if not self.h:
            raise RuntimeError("hipBLASLt: no algorithm")
        copies = int(max(2, min(max_copies, (rotate_mb << 20) // (n * k * (1 if fp8 else 2)) + 2)))
        self.w = [torch.randn((n, k), device=dev, dtype=torch.float32).to(dtype) for _ in range(copies)]
        self.a = torch.randn((m, k), device=dev, dtype=torch.float32).to(dtype)
        self.c = torch.empty((m, n), dtype=torch.bfloat16 if fp8 else dtype, device=dev)

    def launch(self, i):
        w = self.w[i % len(self.w)]
        rc = HipblasLtGemm._lib.hbl_run(self.h, self.a.data_ptr(), w.data_ptr(), self.c.data_ptr(),
                                        torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            raise RuntimeError(f"hipblasLtMatmul rc={rc}")

    def check(self):
        """The comparator computes what we think it computes (guards the operand order)."""
        self.launch(0)
        torch.cuda.synchronize()
        ref = self.a[:8].float() @ self.w[0][:64].float().t()
        got = self.c[:8, :64].float()
        assert torch.allclose(got, ref, rtol=2e-2, atol=2e-2 * ref.abs().max().item()), "hipBLASLt comparator layout"

    def time(self, stream, reps: int = 7, launches: int = 0) -> dict:
        flops = 2.0 * self.m * self.n * self.k
        launches = launches or int(max(5, min(100, 3000.0 / max(flops / (BF16_PEAK_TFLOPS * 1e6), 1.0))))
        # eager launches (each >= 50 us of GPU work against ~5 us of host work per call: the GPU queue never runs dry);
        # hipBLASLt under stream capture faulted on one shape on this stack, and the comparator must not be fragile
        mode = "eager, back-to-back"
        us = []
        with torch.cuda.stream(stream):
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.02:
                self.launch(0)
                stream.synchronize()
            for _ in range(reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for i in range(launches):
                    self.launch(i)
                e1.record(stream)
                stream.synchronize()
                us.append(e0.elapsed_time(e1) * 1e3 / launches)
        med = median(us)
        return {"us": med, "us_min": min(us), "tflops": flops / med / 1e6, "launches": launches, "reps": reps, "launch": mode}

    def time_best(self, stream, first: dict, max_algos: int = 24) -> dict:
        """The fastest of the heuristic's results (the reference's `bench_matmul -algo tune`, matmul_hipblaslt.cc:220-247): every usable result gets a
        short timing (3 x a few launches), the best one the full treatment of time().  `first` = time() of result 0.  Returns its dict + index / count."""
        count = int(HipblasLtGemm._lib.hbl_count(self.h))
        quick = {0: first["us"]}
        flops = 2.0 * self.m * self.n * self.k
        launches = int(max(3, min(30, 1000.0 / max(flops / (BF16_PEAK_TFLOPS * 1e6), 1.0))))
        with torch.cuda.stream(stream):
            for i in range(1, min(count, max_algos)):
                if HipblasLtGemm._lib.hbl_select(self.h, i) != 0:
                    continue
                try:
                    self.launch(0)
                    stream.synchronize()
                    us = []
                    for _ in range(3):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record(stream)
                        for j in range(launches):
                            self.launch(j)
                        e1.record(stream)
                        stream.synchronize()
                        us.append(e0.elapsed_time(e1) * 1e3 / launches)
                    quick[i] = median(us)
                except RuntimeError:
                    continue
        best = min(quick, key=quick.get)
        HipblasLtGemm._lib.hbl_select(self.h, best)
        out = dict(first) if best == 0 else self.time(stream, reps=5)
        if best != 0 and out["us"] > first["us"]:      # (the short timing flattered it)
            out, best = dict(first), 0
        HipblasLtGemm._lib.hbl_select(self.h, 0)
        out.update({"algo_index": best, "algos_timed": len(quick), "algos_found": count})
        return out

    def close(self):
        if self.h:
            HipblasLtGemm._lib.hbl_destroy(self.h)
            self.h = None
        self.w = None
