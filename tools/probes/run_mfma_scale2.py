import ctypes as C
from pathlib import Path
import numpy as np, torch
import importlib.util, sys
spec = importlib.util.spec_from_file_location("p1", str(Path(__file__).resolve().parent / "run_mfma_scale.py"))
# reuse helpers without re-running: minimal copies
lib = C.CDLL(str(Path(__file__).resolve().parent / "libmfmaprobe.so"))
dev = "cuda"
def run(a, bw, sa, sb, opa=0, opb=0):
    ta, tb = torch.from_numpy(a.astype(np.int32)).to(dev), torch.from_numpy(bw.view(np.int32).copy()).to(dev)
    tsa, tsb = torch.from_numpy(sa.astype(np.int32)).to(dev), torch.from_numpy(sb.astype(np.int32)).to(dev)
    out = torch.zeros(256, dtype=torch.float32, device=dev)
    lib.run_mfma_scale(opa, opb, C.c_void_p(ta.data_ptr()), C.c_void_p(tb.data_ptr()), C.c_void_p(tsa.data_ptr()), C.c_void_p(tsb.data_ptr()), C.c_void_p(out.data_ptr()), None)
    torch.cuda.synchronize()
    return out.cpu().numpy().reshape(64, 4)
ones_scale = np.full(64, 0x7F7F7F7F, dtype=np.int64)
a = np.full((64, 8), 0x22222222, dtype=np.int64); a[:, 4:] = 0   # every fp4 = 1.0
ONE = 0x38  # e4m3 1.0
sb = np.array([0x7F7F7F00 | (127 + (l >> 4)) for l in range(64)], dtype=np.int64)
for name, g_sel, byte_lo in (("k in [16,32): lane group 1 bytes 0-15 (block 0)", 1, 0), ("k in [64,80): lane group 0 bytes 16-31 (block 2)", 0, 16),
                             ("k in [112,128): lane group 3 bytes 16-31 (block 3)", 3, 16), ("k in [32,48): lane group 2 bytes 0-15 (block 1)", 2, 0)):
    b = np.zeros((64, 32), dtype=np.uint8)
    for l in range(64):
        if (l >> 4) == g_sel:
            b[l, byte_lo:byte_lo + 16] = ONE
    o = run(a, b.view(np.uint32).reshape(64, 8), ones_scale, sb)
    print(name, "-> D =", o[0, 0], " (16 = x1, 32 = x2, 64 = x4, 128 = x8)")
