import ctypes as C
from pathlib import Path
import numpy as np, torch
lib = C.CDLL(str(Path(__file__).resolve().parent / "libmfmaprobe.so"))
dev = "cuda"
def fp8(x):  # e4m3 byte of a float
    return int(torch.tensor([x], dtype=torch.float32).to(torch.float8_e4m3fn).view(torch.uint8).item())
def run(a, b, sa, sb, opa=0, opb=0):
    ta, tb = torch.from_numpy(a.astype(np.int32)).to(dev), torch.from_numpy(b.view(np.int32).copy()).to(dev)
    tsa, tsb = torch.from_numpy(sa.astype(np.int32)).to(dev), torch.from_numpy(sb.astype(np.int32)).to(dev)
    out = torch.zeros(256, dtype=torch.float32, device=dev)
    lib.run_mfma_scale(opa, opb, C.c_void_p(ta.data_ptr()), C.c_void_p(tb.data_ptr()), C.c_void_p(tsa.data_ptr()), C.c_void_p(tsb.data_ptr()), C.c_void_p(out.data_ptr()), None)
    torch.cuda.synchronize()
    return out.cpu().numpy().reshape(64, 4)
ones_scale = np.full(64, 0x7F7F7F7F, dtype=np.int64)
# B bytes: lane (m = l&15, g = l>>4), byte index bi in 0..31
def make_b(f):
    b = np.zeros((64, 32), dtype=np.uint8)
    for l in range(64):
        for bi in range(32):
            b[l, bi] = fp8(f(l & 15, l >> 4, bi))
    return b.reshape(64, 8, 4).view(np.uint32).reshape(64, 8).view(np.int32) if False else b
def b_words(b):
    return b.reshape(64, 32).copy().view(np.uint32).reshape(64, 8)
print("== which B byte pairs with A nibble (reg r0, nibble i0) of lane group g0")
for g0 in (0, 2):
    for r0 in (0, 1, 3):
        for i0 in (0, 1, 5):
            a = np.zeros((64, 8), dtype=np.int64)
            for l in range(64):
                if (l >> 4) == g0:
                    a[l, r0] = 2 << (4 * i0)      # fp4 code 2 = 1.0
            lo = run(a, b_words(make_b(lambda m, g, bi: bi % 8 + 1)), ones_scale, ones_scale)
            hi = run(a, b_words(make_b(lambda m, g, bi: bi // 8 + 1)), ones_scale, ones_scale)
            gg = run(a, b_words(make_b(lambda m, g, bi: g + 1)), ones_scale, ones_scale)
            print(f"g0={g0} r0={r0} i0={i0}: byte%8+1={lo[0,0]} byte//8+1={hi[0,0]} sum_over_matching_bytes g+1={gg[0,0]}  (uniform over outputs: {np.all(lo==lo[0,0])})")
print("== output layout: A row n nonzero only in lane n0=5 (all g), B = m+1")
a = np.zeros((64, 8), dtype=np.int64)
for l in range(64):
    if (l & 15) == 5:
        a[l, 0] = 2
o = run(a, b_words(make_b(lambda m, g, bi: (m + 1) if bi == 0 else 0)), ones_scale, ones_scale)
nz = np.argwhere(o != 0)
print("nonzero outputs at (lane, reg):", nz[:8].tolist(), "values", [o[i, j] for i, j in nz[:8]])
print("== scales: A scale byte0=128 (x2), byte1=127, byte2=126(x0.5), byte3=129(x4); B scale all 127")
a = np.zeros((64, 8), dtype=np.int64); a[:, 0] = 2
bw = b_words(make_b(lambda m, g, bi: 1.0 if bi == 0 else 0))
sa = np.full(64, (129 << 24) | (126 << 16) | (127 << 8) | 128, dtype=np.int64)
for opa in (0, 1, 2, 3):
    print("opsel_a", opa, run(a, bw, sa, ones_scale, opa, 0)[0, 0])
sbv = np.full(64, (129 << 24) | (126 << 16) | (127 << 8) | 128, dtype=np.int64)
for opb in (0, 1, 3):
    print("opsel_b", opb, run(a, bw, ones_scale, sbv, 0, opb)[0, 0])
print("== per-lane scale: A scale differs per lane group g (127+g), A nonzero all g, B byte0=1")
sa = np.array([0x7F7F7F00 | (127 + (l >> 4)) for l in range(64)], dtype=np.int64)
print(run(a, bw, sa, ones_scale)[0, 0], "expected 1+2+4+8=15 if lane-group g's scale applies to its own 32 k")
