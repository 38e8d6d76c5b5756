#!/usr/bin/env python3
"""How long must a kernel run before its timing is stable (DVFS), and what clock does the chip settle at?
Times the default M=512 8192^2 GEMM after 0.02 / 0.2 / 1.0 / 3.0 s of warm-up and samples rocm-smi while it runs."""
import subprocess, sys, time, threading
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT / "tools")); sys.path.insert(0, str(ROOT / "petit-kernel_amd"))
import torch
import benchlib as BL
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(dev)
w = BL.Weights("nv", 8192, 8192, 1280, dev)
g = BL.Gemm(w, int(sys.argv[1]) if len(sys.argv) > 1 else 512, torch.bfloat16, dev)
sid = int(sys.argv[2], 16) if len(sys.argv) > 2 else g.default_solution()
samples = []
stop = False
def poll():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            samples.append([l.strip() for l in out.splitlines() if "sclk" in l or "Power" in l or "power" in l])
        except Exception as e:
            samples.append([repr(e)])
        time.sleep(0.3)
t = threading.Thread(target=poll); t.start()
for warm in (0.02, 0.2, 1.0, 3.0):
    us = BL.time_graph(g.launcher(sid), 100, 7, stream, warm_s=warm)
    print(f"warm {warm:5.2f} s: median {BL.median(us):8.2f} us  min {min(us):8.2f}  max {max(us):8.2f}", flush=True)
stop = True; t.join()
for s in samples[:3] + samples[-4:]:
    print(s)
