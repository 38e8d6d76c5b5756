import ctypes as C
from pathlib import Path
import numpy as np, torch
lib = C.CDLL(str(Path(__file__).resolve().parent / "libmfmaprobe.so"))
dev = "cuda"
def run(a, bw, sa, sb, opa=0, opb=0):
    ta, tb = torch.from_numpy(a.astype(np.int32)).to(dev), torch.from_numpy(bw.view(np.int32).copy()).to(dev)
    tsa, tsb = torch.from_numpy(sa.astype(np.int32)).to(dev), torch.from_numpy(sb.astype(np.int32)).to(dev)
    out = torch.zeros(256, dtype=torch.float32, device=dev)
    lib.run_mfma_scale(opa, opb, C.c_void_p(ta.data_ptr()), C.c_void_p(tb.data_ptr()), C.c_void_p(tsa.data_ptr()), C.c_void_p(tsb.data_ptr()), C.c_void_p(out.data_ptr()), None)
    torch.cuda.synchronize()
    o = out.cpu().numpy().reshape(64, 4)
    D = np.zeros((16, 16))   # D[n][m]: lane l -> m = l&15, n = 4*(l>>4)+reg
    for l in range(64):
        for r in range(4):
            D[4 * (l >> 4) + r, l & 15] = o[l, r]
    return D
ones = np.full(64, 0x7F7F7F7F, dtype=np.int64)
a = np.full((64, 8), 0x22222222, dtype=np.int64); a[:, 4:] = 0       # all fp4 = 1.0
b = np.full((64, 32), 0x38, dtype=np.uint8)                            # all fp8 = 1.0
bw = b.view(np.uint32).reshape(64, 8)
# B scale depends on the lane's low index (l&15): 127 + (l&15)%4
sb = np.array([0x7F7F7F00 | (127 + ((l & 15) % 4)) for l in range(64)], dtype=np.int64)
D = run(a, bw, ones, sb)
print("B scale = 2^((l&15)%4): D[n=0][m=0..7] =", D[0, :8], " D[n=0..7][m=0] =", D[:8, 0])
sa = np.array([0x7F7F7F00 | (127 + ((l & 15) % 4)) for l in range(64)], dtype=np.int64)
D = run(a, bw, sa, ones)
print("A scale = 2^((l&15)%4): D[n=0][m=0..7] =", D[0, :8], " D[n=0..7][m=0] =", D[:8, 0])
# columns m >= 2 are "absent": data 0 and scale byte 0 (what an out-of-range activation row looks like)
b2 = np.zeros((64, 32), dtype=np.uint8)
sb2 = np.zeros(64, dtype=np.int64)
for l in range(64):
    if (l & 15) < 2:
        b2[l, :] = 0x38
        sb2[l] = 127
D = run(a, b2.view(np.uint32).reshape(64, 8), ones, sb2)
print("absent columns (scale byte 0): D[:, 0] =", D[:, 0], " D[:, 1] =", D[:, 1], " D[:, 2] =", D[:, 2])
sb3 = np.where(np.arange(64) % 16 < 2, 127, 127).astype(np.int64)
D = run(a, b2.view(np.uint32).reshape(64, 8), ones, sb3)
print("absent columns (scale byte 127): D[:, 0] =", D[:, 0])
print("== absent columns with opsel variations")
def runo(a, bw, sa, sb, opa, opb):
    ta, tb = torch.from_numpy(a.astype(np.int32)).to(dev), torch.from_numpy(bw.view(np.int32).copy()).to(dev)
    tsa, tsb = torch.from_numpy(sa.astype(np.int32)).to(dev), torch.from_numpy(sb.astype(np.int32)).to(dev)
    out = torch.zeros(256, dtype=torch.float32, device=dev)
    lib.run_mfma_scale(opa, opb, C.c_void_p(ta.data_ptr()), C.c_void_p(tb.data_ptr()), C.c_void_p(tsa.data_ptr()), C.c_void_p(tsb.data_ptr()), C.c_void_p(out.data_ptr()), None)
    torch.cuda.synchronize()
    o = out.cpu().numpy().reshape(64, 4)
    D = np.zeros((16, 16))
    for l in range(64):
        for r in range(4):
            D[4 * (l >> 4) + r, l & 15] = o[l, r]
    return D
for opa in (0, 1, 2, 3):
    D = runo(a, b2.view(np.uint32).reshape(64, 8), ones, sb2, opa, 0)
    print("opsel_a", opa, "sb absent=0:", D[:6, 0])
# B scale register of absent lanes holds 0 in ALL bytes vs only byte 0
sb4 = np.where(np.arange(64) % 16 < 2, 0x7F, 0).astype(np.int64)   # valid lanes: byte0=127, other bytes 0
for opa in (0, 1, 2, 3):
    D = runo(a, b2.view(np.uint32).reshape(64, 8), ones, sb4, opa, 0)
    print("opsel_a", opa, "sb valid lanes = 0x0000007F:", D[:6, 0])
