import ctypes as C
import time
from pathlib import Path
import torch
lib = C.CDLL(str(Path(__file__).resolve().parent / "libvalu.so"))
dev = torch.device("cuda", 0)
names = ["v_cvt_scalef32_pk_f32_fp4", "v_cvt_scalef32_pk_bf16_fp4", "v_pk_mul_f32", "v_cvt_pk_bf16_f32", "v_fma_f32",
         "v_cvt_scalef32_pk_f16_fp4", "v_pk_mul_f16", "v_cvt_f32_fp8", "v_add_u32", "v_dot2_f32_bf16", "v_pk_fma_f32",
         "v_permlane32_swap_b32", "v_cvt_pkrtz_f16_f32", "v_perm_b32"]
N = 256 * 16
for waves_per_simd in (1, 4):
    threads = 64 * 4 * waves_per_simd  # one block per CU
    for op, nm in enumerate(names):
        out = torch.zeros(4096, dtype=torch.int32, device=dev)
        lib.run_valu_rate(op, 256, threads, C.c_void_p(out.data_ptr()), None)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lib.run_valu_rate(op, 256, threads, C.c_void_p(out.data_ptr()), None)
        e1.record()
        torch.cuda.synchronize()
        ticks = out[:256].float().median().item()
        print(f"{waves_per_simd} wave/SIMD  {nm:30s} {ticks / N:6.2f} ticks/instr/wave   kernel {e0.elapsed_time(e1)*1e3:7.1f} us "
              f"-> {e0.elapsed_time(e1) * 1e-3 * 2.4e9 / (N * waves_per_simd):6.2f} cyc@2.4GHz per instr per SIMD")
