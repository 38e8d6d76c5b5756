"""CPU tests: the oracle (oracle/petit_oracle.c) against everything that pins it.

  * golden vectors produced by the reference's own Python oracle
    (tests/golden/make_golden.py imports tests/ops/test_fp4_gemm_quark.py:9-24)
  * the reference's software floats compiled from its tree (oracle/_ref)
  * the reference's known-answer tables and repack invariants
    (quantization_utils_fp4_test.cc:103-133,246-278,311-365)
  * a literal evaluation of the reference's TAL (CuTe) layouts
"""
import hashlib

import numpy as np
import pytest
import torch

from oracle import oracle as O

NV_CASES = ["nv_64_128_256_1234", "nv_96_64_512_2026"]
MX_CASES = ["mx_64_128_256_1234", "mx_96_96_512_2026"]


def ulp_diff_bf16(a_bits, b_bits):
    """distance in representable values between two same-sign bf16/fp16 bit arrays"""
    a = a_bits.astype(np.int32)
    b = b_bits.astype(np.int32)
    a = np.where(a & 0x8000, 0x8000 - a, a)
    b = np.where(b & 0x8000, 0x8000 - b, b)
    return np.abs(a - b)


@pytest.mark.parametrize("suffix", ["", "_f16"])
@pytest.mark.parametrize("case", NV_CASES)
def test_nv_golden(golden_dir, case, suffix):
    g = np.load(golden_dir / f"{case}{suffix}.npz")
    dq = O.dequant_nvfp4(g["q"], g["s"])
    # dequant is LUT x e4m3: exactly representable, must be bit-identical
    assert np.array_equal(dq.view(np.uint32), g["b_dq"].view(np.uint32))
    c, _ = O.gemm_ref(g["a"], bool(g["a_is_bf16"]), dq, float(g["gs"][0]))
    assert_close_16bit(c, g["c_ref"], bool(g["a_is_bf16"]))


def assert_close_16bit(c_bits, ref_bits, is_bf16):
    """torch accumulates in f32 in BLAS order, the oracle in f64: a result that sits on a
    rounding boundary lands one 16-bit step apart, and where the sum cancels to ~0 the f32
    accumulation error (~1e-6 of the term magnitudes) shows as a few steps of a tiny value.
    Everything else must be bit-identical."""
    to_f = O.bf16_bits_to_f32 if is_bf16 else O.f16_bits_to_f32
    cf, rf = to_f(c_bits), to_f(ref_bits)
    d = ulp_diff_bf16(c_bits, ref_bits)
    scale = np.sqrt(np.mean(rf.astype(np.float64) ** 2))
    ok = (d <= 1) | (np.abs(cf - rf) <= 1e-5 * scale)
    assert ok.all(), (d.max(), np.abs(cf - rf).max(), scale)
    assert (d == 0).mean() > 0.99


@pytest.mark.parametrize("suffix", ["", "_f16"])
@pytest.mark.parametrize("case", MX_CASES)
def test_mx_golden(golden_dir, case, suffix):
    g = np.load(golden_dir / f"{case}{suffix}.npz")
    dq = O.dequant_mxfp4(g["q"], g["s"])
    assert np.array_equal(dq.view(np.uint32), g["b_dq"].view(np.uint32))
    c, _ = O.gemm_ref(g["a"], bool(g["a_is_bf16"]), dq, float(g["gs"][0]))
    ref = g["c_ref"]
    is_bf16 = bool(g["a_is_bf16"])
    cf = O.bf16_bits_to_f32(c) if is_bf16 else O.f16_bits_to_f32(c)
    rf = O.bf16_bits_to_f32(ref) if is_bf16 else O.f16_bits_to_f32(ref)
    fin = np.isfinite(rf) & np.isfinite(cf)
    # e8m0 1..237 spans 2^-126..2^110: with fp16 outputs most results overflow the 16-bit type, and they
    # must do so in the SAME places with the same sign (no NaN: no inf - inf, the f32 sums stay finite)
    assert np.array_equal(np.isfinite(rf), np.isfinite(cf))
    assert not np.isnan(rf).any() and not np.isnan(cf).any()
    assert np.array_equal(np.sign(rf[~fin]), np.sign(cf[~fin]))
    assert is_bf16 <= bool(fin.all())          # bf16 has f32's exponent range: nothing may overflow there
    # the MX fixture scales the f32 matmul result by gs AFTER the product
    # (tests/ops/test_fp4_gemm_quark.py:87), the oracle scales the weights: f32-rounding apart
    assert np.allclose(cf[fin], rf[fin], rtol=2 ** -7, atol=0)
    # the reference's own _gemm_ref on the gs-scaled weights (the NV test's formulation, :49-50) rounds exactly
    # like the oracle does: bit-identical up to f32-vs-f64 accumulation (see assert_close_16bit)
    if "c_ref_gemm" in g.files and fin.all():
        assert_close_16bit(c, g["c_ref_gemm"], is_bf16)


def config1_inputs():
    m, n, k = 1, 4096, 4096
    rng = np.random.default_rng(1234)
    a = torch.from_numpy(rng.standard_normal((m, k), dtype=np.float32)).bfloat16()
    q = rng.integers(0, 256, (n, k // 2), dtype=np.uint8)
    s_f = rng.random((n, k // 16), dtype=np.float32) * 3.5 + 0.25
    s = torch.from_numpy(s_f).to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    gs = np.float32(rng.random() * 1.5 + 0.5)
    a_bits = a.view(torch.int16).numpy().view(np.uint16)
    return a_bits, q, s, gs


def test_config1_golden(golden_dir):
    """BASELINE.json configs[0]: M=1, K=N=4096 bf16 x nvfp4 via the CPU path."""
    g = np.load(golden_dir / "config1_nv_1_4096_4096.npz")
    a, q, s, gs = config1_inputs()
    sha = lambda x: hashlib.sha256(np.ascontiguousarray(x).tobytes()).hexdigest()
    assert sha(a) == str(g["sha_a"]) and sha(q) == str(g["sha_q"]) and sha(s) == str(g["sha_s"])
    assert np.float32(gs) == g["gs"][0]
    c = O.fp4_gemm_cpu(a, True, q, s, gs, "nvfp4")
    assert_close_16bit(c, g["c_ref"], True)


def test_dequant_tables(golden_dir):
    """16 codes x 126 positive e4m3 scales / e8m0 1..237
    (quantization_utils_fp4_test.cc:246-278,344-365)."""
    t = np.load(golden_dir / "dequant_tables.npz")
    L = O.lib()
    nv = np.array([[L.po_fp4_to_f32(c) * L.po_e4m3_to_f32(s) for s in range(1, 0x7F)] for c in range(16)],
                  dtype=np.float32)
    mx = np.array([[L.po_fp4_to_f32(c) * L.po_e8m0_to_f32(s) for s in range(1, 238)] for c in range(16)],
                  dtype=np.float32)
    assert np.array_equal(nv.view(np.uint32), t["nv"].view(np.uint32))
    assert np.array_equal(mx.view(np.uint32), t["mx"].view(np.uint32))
    # every product survives the 16-bit types exactly (the reference's exhaustive test asserts
    # equality after rounding): bf16 for both, fp16 for NV
    assert np.array_equal(O.bf16_bits_to_f32(O.f32_to_bf16_bits(nv)), nv)
    assert np.array_equal(O.bf16_bits_to_f32(O.f32_to_bf16_bits(mx)), mx)
    assert np.array_equal(nv.astype(np.float16).astype(np.float32), nv)


def test_scalar_formats_vs_torch():
    L = O.lib()
    bits = torch.arange(256, dtype=torch.uint8)
    e4 = bits.view(torch.float8_e4m3fn).float().numpy()
    e5 = bits.view(torch.float8_e5m2).float().numpy()
    e8 = bits.view(torch.float8_e8m0fnu).float().numpy()
    for i in range(256):
        for mine, ref in ((L.po_e4m3_to_f32(i), e4[i]), (L.po_e5m2_to_f32(i), e5[i])):
            assert (np.isnan(mine) and np.isnan(ref)) or mine == ref, (i, mine, ref)
    for i in range(1, 255):  # 0 and 255: reference semantics differ from torch's (0 -> 2^-127, nan)
        assert L.po_e8m0_to_f32(i) == e8[i]
    assert L.po_e8m0_to_f32(0) == 0.0  # dequant.cuh:198-203: exponent field 0, mantissa 0
    rng = np.random.default_rng(7)
    x = np.concatenate([rng.standard_normal(20000).astype(np.float32) * s
                        for s in (1e-8, 1e-6, 1e-4, 1.0, 300.0, 7e4)])
    tb = torch.from_numpy(x).bfloat16().view(torch.int16).numpy().view(np.uint16)
    th = torch.from_numpy(x).half().view(torch.int16).numpy().view(np.uint16)
    mb = np.array([L.po_f32_to_bf16(float(v)) for v in x], dtype=np.uint16)
    mh = np.array([L.po_f32_to_f16(float(v)) for v in x], dtype=np.uint16)
    assert np.array_equal(tb, mb) and np.array_equal(th, mh)
    assert np.array_equal(O.f32_to_bf16_bits(x), mb)


def test_against_reference_floats():
    """oracle/_ref: the reference's lib/tests/floating_points.h compiled from its tree."""
    R = O.ref_lib()
    if R is None:
        pytest.skip("oracle/_ref not built (no /root/reference on this machine)")
    L = O.lib()
    for i in range(256):
        for mine, ref in ((L.po_e4m3_to_f32(i), R.ref_e4m3_to_f32(i)), (L.po_e5m2_to_f32(i), R.ref_e5m2_to_f32(i))):
            assert (np.isnan(mine) and np.isnan(ref)) or (mine == ref and np.signbit(mine) == np.signbit(ref))
    for i in range(1, 0x7F):  # every positive, finite e4m3 (lib/tests/quantization.cc:117-130)
        assert L.po_e4m3_to_e5m3(i) == R.ref_e4m3_to_e5m3(i)
        assert L.po_e5m3_to_f32(L.po_e4m3_to_e5m3(i)) == L.po_e4m3_to_f32(i)
    rng = np.random.default_rng(11)
    raw = rng.integers(0, 2 ** 32, 50000, dtype=np.uint64).astype(np.uint32).view(np.float32)
    for v in raw:
        v = float(v)
        if v != v:
            continue
        assert L.po_f32_to_bf16(v) == R.ref_f32_to_bf16(v)
        assert L.po_f32_to_f16(v) == R.ref_f32_to_f16(v)


def test_petit_format_truth_table():
    """Each byte of a re-encoded word, read as OCP e5m2, is fp4 * 2^-14; -0 -> +0
    (quantization_utils.cu:183-206, dequant.cuh:113-125)."""
    L = O.lib()
    import ctypes as C
    out = (C.c_float * 8)()
    rng = np.random.default_rng(3)
    words = list(rng.integers(0, 2 ** 32, 2000, dtype=np.uint64)) + [0, 0xFFFFFFFF, 0x88888888, 0x76543210, 0xFEDCBA98]
    for w in words:
        w = int(w)
        L.po_petit_word_decode(L.po_petit_format(w), out)
        for i in range(8):
            code = (w >> (4 * i)) & 15
            want = L.po_fp4_to_f32(code)
            assert out[i] == want
            if want == 0:
                assert not np.signbit(out[i])  # -0 squashed
    # lib/tests/quantization.cc:11-26 masks exactly the words PetitFormat never produces
    for w in words[:200]:
        p = L.po_petit_format(int(w))
        for b in range(4):
            assert (p >> (8 * b)) & 0x8E != 0x80 and (p >> (8 * b)) & 0x71 != 0x01


@pytest.mark.parametrize("n,k", [(512, 512), (64, 256), (128, 1024)])
def test_reference_repack_invariant_nv(n, k):
    """NvFp4ToPetitFp4Test (quantization_utils_fp4_test.cc:103-133,370-376), seed 42:
    dequant(native) == petit_dequant(repack(native)), bit exact."""
    rng = np.random.default_rng(42)
    qw = rng.integers(0, 2 ** 32, (n, k // 8), dtype=np.uint64).astype(np.uint32)
    s = rng.integers(1, 0x7F, (n, k // 16), dtype=np.uint8)  # all positive e4m3 (:64-70)
    native = O.dequant_nvfp4(qw.view(np.uint8).reshape(n, k // 2), s)
    packed = O.petit_dequant(O.petit_repack_weights(qw), O.petit_repack_nvscales(s, k), "nvfp4", n, k)
    assert np.array_equal(native == 0, packed == 0)
    assert np.array_equal(np.abs(native).view(np.uint32), np.abs(packed).view(np.uint32))
    nz = native != 0
    assert np.array_equal(native[nz], packed[nz])


@pytest.mark.parametrize("n,k", [(256, 256), (96, 512)])
def test_reference_repack_invariant_mx(n, k):
    """MxFp4DequantTest (quantization_utils_fp4_test.cc:266-278,311-342): row/col-mixing
    scale generator so a transpose in the scale shuffle cannot cancel."""
    rows = np.arange(n)[:, None]
    cols = np.arange(k // 32)[None, :]
    s = (1 + (cols + 29 * rows) % 237).astype(np.uint8)
    rng = np.random.default_rng(42)
    qw = rng.integers(0, 2 ** 32, (n, k // 8), dtype=np.uint64).astype(np.uint32)
    native = O.dequant_mxfp4(qw.view(np.uint8).reshape(n, k // 2), s)
    packed = O.petit_dequant(O.petit_repack_weights(qw), O.petit_repack_mxscales(s, k), "mxfp4", n, k)
    nz = native != 0
    assert np.array_equal(native[nz], packed[nz]) and np.all(packed[~nz] == 0)


# --- literal evaluation of the reference's TAL layouts ---------------------------------

def crd2idx(coord, shape, stride):
    """CuTe/TAL crd2idx (include/causalflow/petit/tal/tensor/stride.h:53-113): an integral
    coordinate into a tuple shape is decomposed colexicographically (leftmost mode fastest);
    a size-1 mode passes a dynamic coordinate straight through."""
    if isinstance(shape, tuple):
        if isinstance(coord, tuple):
            assert len(coord) == len(shape)
            return sum(crd2idx(c, s, d) for c, s, d in zip(coord, shape, stride))
        idx = 0
        last = len(shape) - 1
        for pos, (s, d) in enumerate(zip(shape, stride)):
            size = int(np.prod(np.array(flatten(s), dtype=np.int64)))
            if pos == last:
                idx += crd2idx(coord, s, d)  # the last mode absorbs the remainder
            else:
                idx += crd2idx(coord % size, s, d)
                coord //= size
        return idx
    return coord * stride


def flatten(t):
    if isinstance(t, tuple):
        out = []
        for x in t:
            out.extend(flatten(x))
        return out
    return [t]


def test_reference_layout_closed_forms_match_literal_tal():
    """quantization_utils.cu:20-181: shapes/strides transcribed as DATA and evaluated with a
    generic crd2idx; must agree with the closed forms in oracle/petit_oracle.c."""
    L = O.lib()
    n, k = 128, 512
    # -- weights: RepackQWeightLayout<64,32,2,2,4,1> (:178), kernel :208-253
    gm, gn, pack = 256, 32, 8
    shm_shape = ((2, 2), (4, 1), (16, 4))
    shm_stride = ((1, 16 * gm // pack), (64 // pack, gm * 32 // pack), (gm // pack, 2))
    out_shape = (1, 1, 64, (4, 1))
    out_stride = (n * gm // pack // 4, 64 * gn // pack // 4, 1, (n * 64 // pack // 4, 64))
    for id_m in range(k // gm):
        for id_n in range(n // gn):
            for tid in range(256):
                wid, wtid = divmod(tid, 64)
                o = crd2idx((id_m, id_n, wtid, wid), out_shape, out_stride)
                for i in range(4):
                    c = crd2idx((i, wid, wtid), shm_shape, shm_stride)
                    row, k8 = divmod(c, gm // pack)
                    nn, kk = id_n * gn + row, id_m * gm + k8 * 8
                    assert L.po_petit_weight_word_index(n, nn, kk) == o * 4 + i
    # -- NV scales: RepackScaleLayout<7,64,32,1,2> (:180), kernel :255-304
    gm, gn = 64, 64
    shm_shape = ((2, 2), ((8, 4), (1, 2)))
    shm_stride = ((16 * gm // 16, gm // 16), ((2 * gm // 16, 1), (64 // 16, gm * 32 // 16)))
    out_shape = (1, 1, (32, (1, 2)))
    out_stride = (n * gm // 16 // 4, 64 * gn // 16 // 4, (1, (n * 64 // 16 // 4, 32)))
    for id_m in range(k // gm):
        for id_n in range(n // gn):
            for idx in range(64):
                o = crd2idx((id_m, id_n, idx), out_shape, out_stride)
                for i in range(4):
                    b = crd2idx((i, idx), shm_shape, shm_stride)
                    row, grp = divmod(b, gm // 16)
                    nn, kk = id_n * gn + row, id_m * gm + grp * 16
                    assert L.po_petit_nvscale_byte_index(n, nn, kk) == o * 4 + i
    # -- MX scales: RepackMxScaleLayout<64,32,4,1> (:181)
    gm, gn = 256, 32
    shm_shape = ((2, 2), ((8, 2), (4, 1)))
    shm_stride = ((16 * gm // 32, gm // 32), ((2 * gm // 32, 1), (64 // 32, gm * 32 // 32)))
    out_shape = (1, 1, (16, (4, 1)))
    out_stride = (n * gm // 32 // 4, 64 * gn // 32 // 4, (1, (n * 64 // 32 // 4, 16)))
    for id_m in range(k // gm):
        for id_n in range(n // gn):
            for idx in range(64):
                o = crd2idx((id_m, id_n, idx), out_shape, out_stride)
                for i in range(4):
                    b = crd2idx((i, idx), shm_shape, shm_stride)
                    row, blk = divmod(b, gm // 32)
                    nn, kk = id_n * gn + row, id_m * gm + blk * 32
                    assert L.po_petit_mxscale_byte_index(n, nn, kk) == o * 4 + i
