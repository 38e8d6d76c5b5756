"""Run the hardware-semantics probes on the GPU box; prints JSON."""
import ctypes as C
import json
import struct
from pathlib import Path

import torch

lib = C.CDLL(str(Path(__file__).resolve().parent / "libprobes.so"))
dev = "cuda"
res = {}

# 1. does v_cvt_scalef32_pk_*_fp4 multiply by the full f32 scale or only its exponent?
scales = [1.0, 2.0, 1.5, 3.0, 0.75, 1.875, 0.0, 2.0 ** -126, float.fromhex("0x1p-127"), 2.0 ** 100, float("inf")]
words = [0x00000072] * len(scales)   # byte0: lo nibble 2 (=1.0), hi nibble 7 (=6.0)
w = torch.tensor(words, dtype=torch.int64).to(torch.int32).to(dev)
s = torch.tensor(scales, dtype=torch.float32, device=dev)
of = torch.zeros(2 * len(scales), dtype=torch.float32, device=dev)
ob = torch.zeros(len(scales), dtype=torch.int32, device=dev)
lib.run_probe_cvt_scale(C.c_void_p(w.data_ptr()), C.c_void_p(s.data_ptr()), C.c_void_p(of.data_ptr()),
                        C.c_void_p(ob.data_ptr()), len(scales), None)
torch.cuda.synchronize()
of = of.cpu().tolist()
ob = ob.cpu().tolist()
res["cvt_scale"] = [{"scale": sc, "f32": of[2 * i:2 * i + 2],
                     "bf16": [struct.unpack("f", struct.pack("I", (ob[i] & 0xFFFF) << 16))[0],
                              struct.unpack("f", struct.pack("I", (ob[i] & 0xFFFF0000) & 0xFFFFFFFF))[0]]}
                    for i, sc in enumerate(scales)]

# 2. raw buffer range check: buffer of 256 B inside a 4 KiB allocation filled with index+1
buf = (torch.arange(1024, dtype=torch.int32, device=dev) + 1)
out = torch.zeros(64, dtype=torch.int32, device=dev)
cases = {"in_range": (0, 0), "voff_oob": (256, 0), "soff_oob": (0, 256), "voff_huge_soff0": (0x80000000, 0),
         "soff_huge": (0, 0x80000000 - 4096)}
for name, (vo, so) in cases.items():
    if name == "soff_huge":
        continue  # would fault if soffset is not range-checked
    out.zero_()
    lib.run_probe_buffer_oob(C.c_void_p(buf.data_ptr()), 256, vo, so, C.c_void_p(out.data_ptr()), None)
    torch.cuda.synchronize()
    res.setdefault("buffer_oob", {})[name] = out[:4].cpu().tolist()
print(json.dumps(res, indent=1))
