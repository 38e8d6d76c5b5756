import ctypes as C
from pathlib import Path
import torch
lib = C.CDLL(str(Path(__file__).resolve().parent / "libvalu.so"))
dev = torch.device("cuda", 0)
names = ["cvt_scalef32_pk_f32_fp4 (+add)", "cvt_scalef32_pk_bf16_fp4 (+add)", "v_pk_mul_f32", "cvt_pk_bf16_f32 (+add)", "v_fma_f32",
         "cvt_scalef32_pk_f16_fp4 (+add)", "v_pk_mul_f16", "cvt_f32_fp8 (+add)", "lshr+add (2 int ops)"]
for threads in (64, 256):
    for op, nm in enumerate(names):
        out = torch.zeros(2 * 4096, dtype=torch.int32, device=dev)
        blocks = 256 if threads == 256 else 1024
        lib.run_valu_rate(op, blocks, threads, C.c_void_p(out.data_ptr()), None)
        torch.cuda.synchronize()
        cyc = out[0::2][:blocks].float()
        per = cyc.median().item() / (64 * 8)
        print(f"threads/block {threads:4d}  {nm:34s} {per:6.2f} s_memtime ticks per loop-body op (incl. helper op)")
