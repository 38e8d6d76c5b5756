#!/usr/bin/env python3
"""tools/fuzz_parity.py [seed] [seconds] -- random problems against the oracle through the public Python surface (needs an MI355X).

Draws (weight format, activation dtype, M, N, K) from lists that cover every span size (K % 1024 / 512 / 256), ragged M and N, every M bucket, and one of
six call forms: solution_id = -1 plain / with a bias / with the SiLU-mul epilogue, and the three native classes through their sentinels (MXFP4 weights raw; NVFP4 weights on their attached image)
(checked for exact semantics against the oracle run on the CPU-quantised activations).  Bounds and helpers are the test suite's own (tests/test_gpu_parity.py).
Prints every failure with its margin (error / bound) and a summary line; profiles/r04_fuzz.txt is such a log."""
import sys, numpy as np, torch, time
sys.path.insert(0, "petit-kernel_amd"); sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import conftest  # noqa
import test_gpu_parity as T
import petit_kernel as pk
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
t0 = time.time(); n_ok = 0; fails = []
Ns = [16 * i for i in (1, 2, 3, 5, 7, 9, 12, 20, 33, 64, 100, 129, 255, 256, 257, 400, 512, 641)]
while time.time() - t0 < float(sys.argv[2]) if len(sys.argv) > 2 else 150:
    kind = rng.choice(["nv", "mx"]); is_bf16 = bool(rng.integers(0, 2))
    n = int(rng.choice(Ns)); 
    if kind == "mx" and n % 32: n += 16
    k = int(rng.choice([256, 512, 768, 1024, 1280, 1536, 2048, 3072, 4096, 5120, 7168, 8192]))
    m = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 32, 33, 44, 48, 63, 64, 65, 100, 127, 128, 129, 200, 255, 256, 257, 300, 511, 512, 513, 600, 700, 1024, 1100, 1500, 2084, 2200]))   # (the last five: ragged prefill M, bulk + tail launches)
    if m * n * k > 4e9: continue
    a, q, s, gs = T.random_problem(kind, m, n, k, int(rng.integers(1 << 30)), is_bf16)
    # (round 6: NVFP4 weights take the native classes too -- on their MFMA-native image, through the attached-weights route of the reference's entry point)
    mode = rng.choice(["auto", "explicit", "explicit", "bias", "silu", "fp6", "fp8", "fp4"])
    try:
        if m <= 16 and rng.random() < 0.25:    # a grouped launch: 2-4 weight matrices of random N on the same activation rows
            mode = "grouped"
            dtype = torch.bfloat16 if is_bf16 else torch.float16
            ad = T.from_bits(a, dtype).to("cuda")
            members, refs = [], []
            for j in range(int(rng.integers(2, 5))):
                nj = int(rng.choice(Ns))
                if kind == "mx" and nj % 32: nj += 16
                _, qj, sj, gsj = T.random_problem(kind, 1, nj, k, int(rng.integers(1 << 30)), is_bf16)
                qd = torch.from_numpy(qj).to("cuda")
                if kind == "nv":
                    bj, spj = pk.repack_nvfp4(qd.view(torch.int32), nj, k), pk.process_nvfp4_scales(torch.from_numpy(sj).to("cuda").view(torch.float8_e4m3fn), nj, k)
                else:
                    bj, spj = pk.repack_mxfp4(qd.view(torch.int32), nj, k), pk.process_mxfp4_scales(torch.from_numpy(sj).to("cuda"), nj, k)
                members.append((bj, spj, torch.tensor([gsj], dtype=torch.float32, device="cuda"), nj))
                refs.append((T.oracle_ref(kind, a, is_bf16, qj, sj, gsj), T.oracle_sum_abs(kind, a, is_bf16, qj, sj, gsj)))
            outs = pk.mul_fp4_a16_grouped("nvfp4" if kind == "nv" else "mxfp4", ad, members, m, k, -1)
            for c, (ref, sa) in zip(outs, refs):
                T.check_gemm(T.bits(c), ref, is_bf16, sa)
        elif mode == "explicit":     # a random enumerated kernel of the exact class, with a random K split when the kernel takes one
            h = pk.PetitSolutionHints()
            h.a_type = h.c_type = torch.bfloat16 if is_bf16 else torch.float16
            h.b_type = pk.DataType.float4_e2m1 if kind == "nv" else pk.DataType.mxfloat4_e2m1
            sols = pk.ops.get_fp4_solutions(h, m, n, k)
            sid = int(sols[int(rng.integers(len(sols)))])
            sk = int(rng.choice([1, 1, 2, 4]))
            sid_k = (sid & ~(0xF << 60)) | (sk << 60)
            try:
                c = T.run_case(pk, kind, a, is_bf16, q, s, gs, m, n, k, sid_k)
            except RuntimeError:      # (this kernel kind has no such split / K too short)
                c = T.run_case(pk, kind, a, is_bf16, q, s, gs, m, n, k, sid)
                sid_k = sid
            mode = f"explicit {sid_k:#x}"
            T.check_gemm(c, T.oracle_ref(kind, a, is_bf16, q, s, gs), is_bf16, T.oracle_sum_abs(kind, a, is_bf16, q, s, gs))
        elif mode == "auto":
            c = T.run_case(pk, kind, a, is_bf16, q, s, gs, m, n, k)
            T.check_gemm(c, T.oracle_ref(kind, a, is_bf16, q, s, gs), is_bf16, T.oracle_sum_abs(kind, a, is_bf16, q, s, gs))
        else:
            dtype = torch.bfloat16 if is_bf16 else torch.float16
            ad = T.from_bits(a, dtype).to("cuda"); qd = torch.from_numpy(q).to("cuda"); gsd = torch.tensor([gs], dtype=torch.float32, device="cuda")
            if kind == "nv":
                b, sp, mul = pk.repack_nvfp4(qd.view(torch.int32), n, k), pk.process_nvfp4_scales(torch.from_numpy(s).to("cuda").view(torch.float8_e4m3fn), n, k), pk.mul_nvfp4_a16
            else:
                b, sp, mul = pk.repack_mxfp4(qd.view(torch.int32), n, k), pk.process_mxfp4_scales(torch.from_numpy(s).to("cuda"), n, k), pk.mul_mxfp4_a16
            ref = T.oracle_ref(kind, a, is_bf16, q, s, gs).astype(np.float64)
            if mode == "bias":
                bias = (torch.randn(n) * 0.5).to(dtype)
                c = mul(ad, b, sp, gsd, m, n, k, -1, bias=bias.cuda())
                T.check_gemm(T.bits(c), ref + bias.float().numpy()[None, :], is_bf16, T.oracle_sum_abs(kind, a, is_bf16, q, s, gs))
            elif mode == "silu":
                if n % 32: continue
                sc = 1.0 / max(1.0, np.sqrt(np.mean(ref ** 2)))
                gsd2 = torch.tensor([gs * sc], dtype=torch.float32, device="cuda")
                y = ref * sc
                g, u = y[:, : n // 2], y[:, n // 2:]
                want = g / (1.0 + np.exp(-g)) * u
                c = mul(ad, b, sp, gsd2, m, n, k, -1, activation="silu_mul")
                got = T.to_f32(T.bits(c), is_bf16).astype(np.float64)
                tol = np.maximum(2e-2, 2e-2 * np.abs(want)) + 1e-4 * (np.abs(T.to_f32(a, is_bf16)) @ np.abs(T.O.dequant_nvfp4(q, s) if kind == "nv" else T.O.dequant_mxfp4(q, s)).T * gs * sc)[:, : n // 2]
                assert (np.abs(got - want) <= tol).all(), f"silu max err {np.abs(got - want).max()}"
            else:
                fmt = {"fp6": "mxfp6", "fp8": "mxfp8", "fp4": "mxfp4"}[mode]
                quant = {"mxfp6": T.quantize_act_mxfp6, "mxfp8": T.quantize_act_mxfp8, "mxfp4": T.quantize_act_mxfp4}[fmt]
                a_q = quant(T.to_f32(a, is_bf16))
                dq = T.O.dequant_mxfp4(q, s) if kind == "mx" else T.O.nv6_reencode(q, s)[0]     # (NVFP4: the image's values)
                _, exact = T.O.gemm_ref(T.O.f32_to_bf16_bits(a_q), True, dq, gs)
                sum_abs = (np.abs(T.to_f32(a, is_bf16)) @ np.abs(dq).T) * gs
                if kind == "mx":
                    c = pk.mul_mxfp4_native(ad, b, sp, gsd, m, n, k, T.NATIVE_SENTINEL(pk, fmt))
                else:
                    image = pk.nvfp4_native_image(b, sp, n, k)
                    pk.attach_nvfp4_native(b, image)
                    try:
                        c = pk.mul_nvfp4_a16(ad, b, sp, gsd, m, n, k, T.NATIVE_SENTINEL(pk, fmt))
                    finally:
                        pk.attach_nvfp4_native(b, None)
                got = T.to_f32(T.bits(c), is_bf16).astype(np.float64)
                # a ragged prefill M may run as bulk (in the class) + a short tail through the exact default pick (petit_gemm_row_split): tail rows against the exact oracle
                hh = pk.PetitSolutionHints(); hh.a_type = hh.c_type = torch.bfloat16 if is_bf16 else torch.float16
                hh.b_type = pk.DataType.float4_e2m1 if kind == "nv" else pk.DataType.mxfloat4_e2m1
                m1 = pk.ops.auto_row_split(hh, m, n, k, solution_id=T.NATIVE_SENTINEL(pk, fmt))
                if m1:
                    dq_true = T.O.dequant_mxfp4(q, s) if kind == "mx" else T.O.dequant_nvfp4(q, s)
                    _, tail_ref = T.O.gemm_ref(a[m1:], is_bf16, dq_true, gs)
                    T.check_gemm(T.bits(c[m1:]), tail_ref, is_bf16, (np.abs(T.to_f32(a[m1:], is_bf16)) @ np.abs(dq_true).T) * gs if kind == "mx" else None)
                    got, exact, a_q, sum_abs = got[:m1], exact[:m1], a_q[:m1], sum_abs[:m1]
                fin = np.isfinite(exact) & (np.abs(exact) < (3e38 if is_bf16 else 6e4))
                err = np.abs(got - exact)[fin]
                bound = np.maximum(np.maximum(1e-2, 1e-2 * np.abs(exact)), T.native_exact_bound(a_q, dq, gs, fmt))[fin]   # (as the tests: derived from the instruction)
                assert (err <= bound).all(), (f"native {fmt}: {int((err > bound).sum())} of {err.size} elements over the bound, worst margin err / bound = {(err / bound).max():.2f} "
                                              f"(there: |exact| = {np.abs(exact)[fin][np.argmax(err / bound)]:.3g}, sum|a||w| = {sum_abs[fin][np.argmax(err / bound)]:.3g})")
        n_ok += 1
    except Exception as e:  # noqa
        fails.append((kind, is_bf16, m, n, k, mode, repr(e)[:320]))
        print("FAIL", fails[-1], flush=True)
print("ok", n_ok, "fails", len(fails))
