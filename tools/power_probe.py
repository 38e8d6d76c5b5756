#!/usr/bin/env python3
"""tools/power_probe.py -- board power and shader clock while one kernel runs back to back (is the M = 512 regime power-limited?).

For each of: the default bf16 x NVFP4 kernel, the default bf16 x MXFP4 kernel, hipBLASLt bf16 dense, the native MXFP6 / MXFP4 picks -- on gate_up
(57344 x 8192) at M = 512: replay a graph of launches for ~2.5 s, sample the GPU's hwmon power / sclk files (else `rocm-smi`) every 50 ms from a
thread, report median power, median clock and the TFLOP/s of the same window.  Output: gpurun_out/power_probe.json."""
import glob
import json
import subprocess
import sys
import threading
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "petit-kernel_amd"))
sys.path.insert(0, str(ROOT / "tools"))
import torch

import benchlib as BL
import petit_kernel as pk


def find_sensors():
    power, freq, cap = [], [], []
    for hw in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        for name in ("power1_average", "power1_input"):
            p = Path(hw) / name
            if p.exists():
                power.append(p)
                break
        f = Path(hw) / "freq1_input"
        if f.exists():
            freq.append(f)
        c = Path(hw) / "power1_cap"
        if c.exists():
            cap.append(c)
    return power, freq, cap


class Sampler(threading.Thread):
    def __init__(self, power, freq):
        super().__init__(daemon=True)
        self.power, self.freq, self.stop_flag, self.samples = power, freq, False, []

    def read_smi(self):
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout
            d = json.loads(out)
            card = next(iter(d.values()))
            pw = next((float(v) for k, v in card.items() if "Power" in k and "W" in k), None)
            ck = next((v for k, v in card.items() if "sclk" in k.lower()), None)
            return pw, ck
        except Exception:   # noqa: BLE001
            return None, None

    def run(self):
        while not self.stop_flag:
            if self.power:
                try:
                    pw = int(self.power[0].read_text()) / 1e6
                    ck = int(self.freq[0].read_text()) / 1e6 if self.freq else None
                except Exception:   # noqa: BLE001
                    pw, ck = None, None
            else:
                pw, ck = self.read_smi()
            self.samples.append((time.time(), pw, ck))
            time.sleep(0.05)


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--nv-solution", default="", help="run ONLY bf16 x NVFP4 with this explicit kernel id (hex): for ablation libraries ($PETIT_AMD_LIB, tools/ablate_wide.sh)")
    ap.add_argument("--out", default=str(ROOT / "gpurun_out" / "power_probe.json"))
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(dev)
    power, freq, cap = find_sensors()
    # the host shows every GPU of the node; ours is the one whose power rises when this process loads the device (other tenants may be running)
    if len(power) > 1:
        def read_all():
            return [int(p.read_text()) / 1e6 for p in power]
        idle = [read_all() for _ in range(10) if not time.sleep(0.05)]
        x = torch.randn((8192, 8192), device=dev, dtype=torch.bfloat16)
        t_end, busy = time.time() + 1.5, []
        while time.time() < t_end:
            for _ in range(20):
                x @ x
            busy.append(read_all())
        torch.cuda.synchronize()
        med = lambda rows, i: sorted(r[i] for r in rows)[len(rows) // 2]
        delta = [med(busy[len(busy) // 2:], i) - med(idle, i) for i in range(len(power))]
        mine = max(range(len(power)), key=lambda i: delta[i])
        print("power rise per card under a bf16 GEMM loop:", [round(d) for d in delta], "-> card", power[mine], flush=True)
        card_dir = power[mine].parent
        power, freq, cap = [power[mine]], [f for f in freq if f.parent == card_dir], [c for c in cap if c.parent == card_dir]
        del x
    out = {"sensors": {"power": [str(p) for p in power], "freq": [str(f) for f in freq],
                       "power_cap_w": [int(c.read_text()) / 1e6 for c in cap] if cap else None}, "runs": []}
    n, k, m = 57344, 8192, 512
    flops = 2.0 * m * n * k
    cases = []
    if args.nv_solution:
        w = BL.Weights("nv", n, k, 1280, dev)
        g = BL.Gemm(w, m, torch.bfloat16, dev)
        sid = int(args.nv_solution, 16)
        from petit_kernel import _lib as _l
        cases.append((f"bf16 x nvfp4 {_l.describe_solution(sid)[:60]}", g, sid))
    for fmt, dt in (() if args.nv_solution else (("nv", torch.bfloat16), ("mx", torch.bfloat16))):
        w = BL.Weights(fmt, n, k, 1280, dev)
        g = BL.Gemm(w, m, dt, dev)
        cases.append((f"bf16 x {fmt}fp4 default", g, None))
    from petit_kernel import _lib
    extra = []
    if not args.nv_solution:
        wmx = BL.Weights("mx", n, k, 1280, dev)
        for name, sent in (("native mxfp6", _lib.PETIT_SOLUTION_AUTO_NATIVE_MXFP6), ("native mxfp4", _lib.PETIT_SOLUTION_AUTO_NATIVE_MXFP4), ("native mxfp8", _lib.PETIT_SOLUTION_AUTO_NATIVE_MXFP8)):
            g = BL.Gemm(wmx, m, torch.bfloat16, dev)
            cases.append((name, g, sent))
        extra = [("hipBLASLt bf16 dense", BL.HipblasLtGemm(m, n, k, torch.bfloat16, dev, 1280), None)]
    for name, g, sent in cases + extra:
        if isinstance(g, BL.HipblasLtGemm):
            launch = g.launch
        else:
            launch = g.launcher(sent if sent is not None else _lib.PETIT_SOLUTION_AUTO)
        launches = 64
        with torch.cuda.stream(stream):
            launch(0)
            stream.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream):
                for i in range(launches):
                    launch(i)
            for _ in range(3):
                graph.replay()
            stream.synchronize()
            s = Sampler(power, freq)
            s.start()
            t0 = time.time()
            reps = 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            while time.time() - t0 < 2.5:
                graph.replay()
                reps += 1
                if reps % 4 == 0:
                    stream.synchronize()
            e1.record(stream)
            stream.synchronize()
            s.stop_flag = True
            s.join()
        ms = e0.elapsed_time(e1)
        us = ms * 1e3 / (reps * launches)
        tail = [x for x in s.samples if x[0] - t0 > 0.8]     # steady state
        pws = sorted(x[1] for x in tail if x[1] is not None)
        cks = sorted(x[2] for x in tail if isinstance(x[2], (int, float)))
        rec = {"kernel": name, "us_per_launch": us, "tflops": flops / us / 1e6, "samples": len(tail),
               "power_w_median": pws[len(pws) // 2] if pws else None, "power_w_max": pws[-1] if pws else None,
               "sclk_mhz_median": cks[len(cks) // 2] if cks else None, "raw_clock_sample": tail[-1][2] if tail else None}
        print(rec, flush=True)
        out["runs"].append(rec)
        time.sleep(1.0)
    Path(ROOT / "gpurun_out").mkdir(exist_ok=True)
    Path(args.out).write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
