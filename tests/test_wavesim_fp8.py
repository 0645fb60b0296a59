"""CPU-side (wave simulator) check of the OCP e4m3 forward path: the quantiser against a numpy restatement of e4m3 round-to-nearest-even
with per-tensor scaling, and the fp8 GEMM / conv index math (64-byte K slabs of 64 elements, ds_read_b64 fragments, v_mfma_f32_32x32x16_fp8_fp8
emulated lane-accurately) against numpy on the de-quantised operands — which the kernel must reproduce up to f32 summation order, because
products of two e4m3 values are exact in f32. Runs without a GPU."""
import ctypes as C

import numpy as np
import pytest

from simlib import Bn, Conv, bf16_round, from_bf16, lib, make_ep, ptr, to_bf16

BF16, F32 = 0, 1


def e4m3_values():
    v = np.zeros(256, np.float32)
    for b in range(256):
        s, e, m = b >> 7, (b >> 3) & 15, b & 7
        if e == 15 and m == 7:
            r = np.nan
        elif e == 0:
            r = m * 2.0 ** -9
        else:
            r = (1 + m / 8) * 2.0 ** (e - 7)
        v[b] = -r if s else r
    return v


E4M3 = e4m3_values()


def quantize_ref(x):
    """numpy restatement: scale = 448 / amax; nearest e4m3 value (ties to even mantissa) of clamp(x * scale)."""
    amax = np.abs(x).max()
    scale = np.float32(448.0) / amax if amax > 0 else np.float32(1)
    y = np.clip(x.astype(np.float32) * scale, -448, 448).astype(np.float32)
    pos = E4M3[:127]                                          # 0 .. 448 ascending (0x00..0x7e)
    idx = np.searchsorted(pos, np.abs(y), side="left").clip(1, 126)
    lo, hi = pos[idx - 1], pos[idx]
    dl, dh = np.abs(y) - lo, hi - np.abs(y)
    pick_hi = (dh < dl) | ((dh == dl) & (idx % 2 == 0))       # ties -> even code
    code = np.where(pick_hi, idx, idx - 1).astype(np.uint8)
    code = np.where(np.abs(y) == 0, 0, code).astype(np.uint8)
    return (code | np.where(np.signbit(y), 0x80, 0).astype(np.uint8)), scale


def _quant(L, x, dtype):
    buf = to_bf16(x) if dtype == BF16 else np.ascontiguousarray(x, np.float32)
    q = np.zeros(x.shape, np.uint8)
    amax, scales = np.zeros(1, np.float32), np.zeros(2, np.float32)
    assert L.clite_fp8_quantize(dtype, ptr(buf), x.size, ptr(amax), ptr(scales), ptr(q), None) == 0
    return q, scales, amax


@pytest.mark.parametrize("dtype", [BF16, F32])
def test_quantizer_matches_e4m3_round_to_nearest_even(dtype):
    L = lib()
    L.clite_fp8_quantize.argtypes = [C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(3)
    x = (rng.standard_normal((37, 64)) * np.exp(rng.standard_normal((37, 64)) * 3)).astype(np.float32)
    x[0, :8] = [0.0, -0.0, 1e-30, -1e-30, 1.0, -1.0, 0.5, 3.0]
    if dtype == BF16:
        x = bf16_round(x)
    q, scales, amax = _quant(L, x, dtype)
    ref, scale = quantize_ref(x)
    assert amax[0] == np.abs(x).max() and np.isclose(scales[0], scale, rtol=1e-6) and np.isclose(scales[1], 1 / scale, rtol=1e-6)
    assert np.array_equal(q & 0x7f, ref & 0x7f) and np.array_equal((q >> 7)[np.abs(x) > 1e-20], (ref >> 7)[np.abs(x) > 1e-20])
    assert np.abs(E4M3[q]).max() == 448.0                      # the largest element lands exactly on the format's maximum
    z = np.zeros((8, 16), np.float32)                          # all-zero tensor: scale 1, zeros
    qz, sz, _ = _quant(L, z, F32)
    assert not qz.any() and sz[0] == 1.0 and sz[1] == 1.0


@pytest.mark.parametrize("dtype", [BF16, F32])
@pytest.mark.parametrize("bad", [np.nan, np.inf, -np.inf])
def test_quantizer_keeps_non_finite_inputs_visible(dtype, bad):
    """fmaxf drops a NaN and the +-448 clamp turned a NaN element into -448, so a diverged tensor quantised to finite e4m3 values and a
    NaN-loss guard never fired. A NaN or an infinity anywhere in the tensor now makes amax non-finite, both scales NaN (every product of the
    consuming GEMM is NaN) and every output code e4m3's NaN; n % 8 != 0 is refused."""
    L = lib()
    L.clite_fp8_quantize.argtypes = [C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    x = np.random.default_rng(5).standard_normal((9, 64)).astype(np.float32)
    x[4, 17] = bad
    q, scales, amax = _quant(L, x, dtype)
    assert not np.isfinite(amax[0]) and np.isnan(scales).all()
    assert ((q & 0x7f) == 0x7f).all()                           # 0x7F / 0xFF: the two NaN codes of e4m3fn
    buf = np.zeros(12, np.float32)
    assert L.clite_fp8_quantize(F32, ptr(buf), 12, ptr(np.zeros(1, np.float32)), ptr(np.zeros(2, np.float32)), ptr(np.zeros(12, np.uint8)), None) == -1


@pytest.mark.parametrize("pol", [0, 2])          # 2: the 256 x 128 tile, which the shape rule takes only for launches of >= 448 such tiles
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 136, 208), (300, 64, 96), (70, 1000, 64)])
def test_gemm_nt_fp8(M, N, K, pol):
    """clite_gemm_nt_fp8 on v_mfma_scale_f32_32x32x64_f8f6f4 (unit block scales; the simulator executes the operand map
    tools/micro/mfma_scale_probe.hip measured on the hardware) against the f64 product of the dequantised operands."""
    if pol and (N <= 64 or M * N > 70000):
        pytest.skip("narrow outputs have one tile; the wide case repeats the 200 x 136 one")
    L = lib()
    assert L.clite_set_tile_policy(pol) == 0
    L.clite_fp8_quantize.argtypes = [C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.clite_gemm_nt_fp8.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(M + N + K)
    A, B = rng.standard_normal((M, K)).astype(np.float32), (rng.standard_normal((N, K)) * 0.1).astype(np.float32)
    qa, sa, _ = _quant(L, A, F32)
    qb, sb, _ = _quant(L, B, F32)
    bias = rng.standard_normal(N).astype(np.float32)
    out = np.zeros((M, N), np.float32)
    ep = make_ep(out, N, out_f32=True, bias=bias, act=1)
    assert L.clite_gemm_nt_fp8(ptr(qa), K, ptr(qb), K, M, N, K, ptr(sa), ptr(sb), C.byref(ep), None) == 0
    ref = np.maximum((E4M3[qa].astype(np.float64) @ E4M3[qb].astype(np.float64).T) * (sa[1] * sb[1]) + bias, 0)
    assert np.abs(out - ref).max() <= 2e-5 * max(np.abs(ref).max(), 1e-6)
    # and the quantisation itself costs what e4m3 costs: a few percent of the f32 product's scale
    exact = np.maximum(A.astype(np.float64) @ B.astype(np.float64).T + bias, 0)
    assert np.abs(out - exact).max() <= 0.08 * np.abs(exact).max()
    # plain epilogue (bf16 store + column statistics)
    o16 = np.zeros((M, N), np.uint16)
    cs = np.zeros((2, N), np.float32)
    assert L.clite_gemm_nt_fp8(ptr(qa), K, ptr(qb), K, M, N, K, ptr(sa), ptr(sb), C.byref(make_ep(o16, N, colsum=cs)), None) == 0
    ref2 = (E4M3[qa].astype(np.float64) @ E4M3[qb].astype(np.float64).T) * (sa[1] * sb[1])
    got = from_bf16(o16)
    assert np.abs(got - ref2).max() <= 6e-3 * np.abs(ref2).max()
    assert np.abs(cs[0] - got.sum(0)).max() <= 1e-4 * max(np.abs(got.sum(0)).max(), 1e-6)
    assert L.clite_set_tile_policy(0) == 0


@pytest.mark.parametrize("N,H,W,Cc,K,R,st,pad", [(2, 8, 8, 64, 64, 3, 1, 1), (3, 6, 6, 48, 136, 1, 1, 0), (2, 9, 7, 128, 32, 3, 2, 1)])
@pytest.mark.parametrize("pol", [0, 2])
def test_conv_fwd_fp8(N, H, W, Cc, K, R, st, pad, pol):
    from test_wavesim_igemm import conv_ref
    if pol and K <= 64:
        pytest.skip("narrow outputs have one tile")
    L = lib()
    assert L.clite_set_tile_policy(pol) == 0
    L.clite_fp8_quantize.argtypes = [C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.clite_conv_fwd_fp8.argtypes = [C.c_void_p] * 7
    rng = np.random.default_rng(H * W + K)
    Ho, Wo = (H + 2 * pad - R) // st + 1, (W + 2 * pad - R) // st + 1
    cv = Conv(BF16, N, H, W, Cc, K, R, R, st, pad, Ho, Wo)
    x = rng.standard_normal((N, H, W, Cc)).astype(np.float32)
    w = (rng.standard_normal((K, R, R, Cc)) * 0.1).astype(np.float32)
    qx, sx, _ = _quant(L, x, F32)
    qw, sw, _ = _quant(L, w, F32)
    y = np.zeros((N, Ho, Wo, K), np.float32)
    assert L.clite_conv_fwd_fp8(ptr(qx), ptr(qw), C.byref(cv), ptr(sx), ptr(sw), C.byref(make_ep(y, K, out_f32=True)), None) == 0
    ref = conv_ref(E4M3[qx], E4M3[qw], st, pad) * (sx[1] * sw[1])
    assert np.abs(y - ref).max() <= 2e-5 * np.abs(ref).max()
    assert L.clite_set_tile_policy(0) == 0


SLOT = 16 * 32          # one amax slot: CLITE_FP8_AMAX_REPLICAS x CLITE_FP8_AMAX_STRIDE words (include/clite.h)


def quantize_at(x, scale):
    """e4m3 codes of clamp(x * scale, +-448), round to nearest even (quantize_ref with the scale given: delayed scaling)."""
    y = np.clip(x.astype(np.float32) * np.float32(scale), -448, 448).astype(np.float32)
    pos = E4M3[:127]
    idx = np.searchsorted(pos, np.abs(y), side="left").clip(1, 126)
    lo, hi = pos[idx - 1], pos[idx]
    dl, dh = np.abs(y) - lo, hi - np.abs(y)
    pick_hi = (dh < dl) | ((dh == dl) & (idx % 2 == 0))
    code = np.where(pick_hi, idx, idx - 1).astype(np.uint8)
    code = np.where(np.abs(y) == 0, 0, code).astype(np.uint8)
    return code | np.where(np.signbit(y), 0x80, 0).astype(np.uint8)


@pytest.mark.parametrize("M,Cc,res", [(70, 64, False), (33, 256, True), (19, 16, True)])
def test_bn_apply_writes_the_e4m3_copy_and_the_amax(M, Cc, res):
    """The producer-fused quantiser (clite_bn.fp8_out / fp8_scale / fp8_amax): bn_apply's e4m3 copy equals the stand-alone
    quantiser's codes of the STORED bf16 output at the given (delayed) scale — values beyond the scale's range saturate at +-448 — and fp8_amax
    receives max |out| of the call (folded into what was there). The bf16 output and the ReLU bits are those of the plain call, bit for bit.
    Ragged row counts and C = 16 (byte-store form of the bits) included."""
    L = lib()
    rng = np.random.default_rng(M + Cc)
    y = bf16_round(rng.standard_normal((M, Cc)).astype(np.float32) * 2 + 0.5)
    r = bf16_round(rng.standard_normal((M, Cc)).astype(np.float32))
    yb, rb = to_bf16(y), to_bf16(r)
    gamma = (1 + 0.1 * rng.standard_normal(Cc)).astype(np.float32)
    beta = (0.1 * rng.standard_normal(Cc)).astype(np.float32)
    stats = np.zeros((1, 3, Cc), np.float32)
    stats[0, 0], stats[0, 1] = y.sum(0), (y * y).sum(0)

    def run(fp8):
        rm, rv = np.zeros(Cc, np.float32), np.ones(Cc, np.float32)
        p = Bn(M, Cc, ptr(stats), ptr(gamma), ptr(beta), ptr(rm), ptr(rv), 1, 1, 0.1, 1e-5, 1, 1, 3 * Cc, 0)
        out = np.zeros((M, Cc), np.uint16)
        bits = np.zeros((M, Cc // 8), np.uint8)
        p.relu_bits = ptr(bits)
        if fp8 is not None:
            p.fp8_out, p.fp8_scale, p.fp8_amax = ptr(fp8[0]), ptr(fp8[1]), ptr(fp8[2])
        assert L.clite_bn_apply(C.byref(p), BF16, ptr(yb), ptr(rb) if res else None, ptr(out), None) == 0
        return out, bits

    out0, bits0 = run(None)
    a = from_bf16(out0)
    scale = np.float32(448.0 / (0.6 * np.abs(a).max()))          # a stale scale: the top 40 % of the range saturates
    q = np.full((M, Cc), 0x55, np.uint8)
    scales = np.array([scale, 1 / scale], np.float32)
    amax = np.zeros(SLOT, np.float32)
    amax[32] = 0.125                                             # (something smaller recorded earlier in the step, by another workgroup)
    out1, bits1 = run((q, scales, amax))
    assert np.array_equal(out0, out1) and np.array_equal(bits0, bits1)
    assert amax.max() == np.abs(a).max() and not amax.reshape(16, 32)[:, 1:].any()
    ref = quantize_at(a, scale)
    assert np.array_equal(q & 0x7f, ref & 0x7f) and np.array_equal((q >> 7)[a != 0], (ref >> 7)[a != 0])
    assert (np.abs(E4M3[q]) == 448.0).any()
    # amax only (the first step: no scale yet), and a NaN in the input stays visible in the amax
    amax2 = np.zeros(SLOT, np.float32)
    run((None, None, amax2))
    assert amax2.max() == np.abs(a).max()
    yb.reshape(-1)[5] = 0x7FC0
    amax3 = np.zeros(SLOT, np.float32)
    run((None, None, amax3))
    assert np.isnan(amax3).any()
    # f32 activations have no fused copy
    p = Bn(M, Cc, ptr(stats), ptr(gamma), ptr(beta), ptr(np.zeros(Cc, np.float32)), ptr(np.ones(Cc, np.float32)), 1, 0, 0.1, 1e-5, 1, 1, 3 * Cc, 0)
    p.fp8_amax = ptr(amax3)
    assert L.clite_bn_apply(C.byref(p), F32, ptr(y), None, ptr(np.zeros((M, Cc), np.float32)), None) != 0


@pytest.mark.parametrize("M,N,K", [(200, 136, 128), (70, 264, 64)])
def test_gemm_nt_fp8_epilogue_leaves_the_e4m3_copy_and_the_amax(M, N, K):
    """clite_epilogue.fp8_out / fp8_scale / fp8_amax (ABI v11, clite_gemm_nt_fp8 only): BERT's FFN1 launch - bias, pre-activation store, GELU, bf16
    store - also leaves the e4m3 codes of the STORED bf16 output at a delayed scale (values beyond its range saturate) and max |out| of the call in
    the amax slot; the bf16 output and the pre-activation are those of the call without the three fields, bit for bit. Every other GEMM entry
    point refuses the fields."""
    L = lib()
    L.clite_fp8_quantize.argtypes = [C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.clite_gemm_nt_fp8.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(M + N)
    A, B = rng.standard_normal((M, K)).astype(np.float32), (rng.standard_normal((N, K)) * 0.2).astype(np.float32)
    qa, sa, _ = _quant(L, A, F32)
    qb, sb, _ = _quant(L, B, F32)
    bias = rng.standard_normal(N).astype(np.float32)

    def run(fp8):
        out, pre = np.zeros((M, N), np.uint16), np.zeros((M, N), np.uint16)
        assert L.clite_gemm_nt_fp8(ptr(qa), K, ptr(qb), K, M, N, K, ptr(sa), ptr(sb), C.byref(make_ep(out, N, bias=bias, act=2, preact=pre, fp8=fp8)), None) == 0
        return out, pre

    out0, pre0 = run(None)
    a = from_bf16(out0)
    scale = np.float32(448.0 / (0.6 * np.abs(a).max()))          # a stale scale: the top of the range saturates
    q = np.full((M, N), 0x55, np.uint8)
    scales = np.array([scale, 1 / scale], np.float32)
    amax = np.zeros(SLOT, np.float32)
    amax[64] = 0.0625
    out1, pre1 = run((q, scales, amax))
    assert np.array_equal(out0, out1) and np.array_equal(pre0, pre1)
    assert amax.max() == np.abs(a).max() and not amax.reshape(16, 32)[:, 1:].any()
    ref = quantize_at(a, scale)
    assert np.array_equal(q & 0x7f, ref & 0x7f) and np.array_equal((q >> 7)[a != 0], (ref >> 7)[a != 0])
    assert (np.abs(E4M3[q]) == 448.0).any()
    amax2 = np.zeros(SLOT, np.float32)          # amax only (the first step: no scale yet)
    run((None, None, amax2))
    assert amax2.max() == np.abs(a).max()
    # refusals: an e4m3 copy without a scale; an f32 output; the bf16 entry point
    o = np.zeros((M, N), np.uint16)
    assert L.clite_gemm_nt_fp8(ptr(qa), K, ptr(qb), K, M, N, K, ptr(sa), ptr(sb), C.byref(make_ep(o, N, fp8=(q, None, None))), None) == -1
    o32 = np.zeros((M, N), np.float32)
    assert L.clite_gemm_nt_fp8(ptr(qa), K, ptr(qb), K, M, N, K, ptr(sa), ptr(sb), C.byref(make_ep(o32, N, out_f32=True, fp8=(None, None, amax2))), None) == -1
    L.clite_gemm_nt.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    Ab, Bb = to_bf16(A), to_bf16(B)
    assert L.clite_gemm_nt(ptr(Ab), K, ptr(Bb), K, M, N, K, BF16, C.byref(make_ep(o, N, fp8=(None, None, amax2))), None) == -1


@pytest.mark.parametrize("M,Cc,p", [(37, 768, 0.0), (9, 264, 0.1)])
def test_layernorm_forward_leaves_the_e4m3_copy_and_the_amax(M, Cc, p):
    """clite_layernorm_fwd_q8 (ABI v11): the LayerNorm forward's bf16 output and statistics are clite_layernorm_fwd's bit for bit (with and without
    the output dropout); the e4m3 codes are those of the stored output at the given scale; the slot receives max |out|."""
    L = lib()
    L.clite_layernorm_fwd.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_uint64, C.c_uint32, C.c_void_p]
    L.clite_layernorm_fwd_q8.argtypes = L.clite_layernorm_fwd.argtypes[:-1] + [C.c_void_p] * 4
    rng = np.random.default_rng(Cc)
    x = to_bf16(rng.standard_normal((M, Cc)).astype(np.float32) * 3 + 1)
    gamma = (1 + 0.1 * rng.standard_normal(Cc)).astype(np.float32)
    beta = (0.1 * rng.standard_normal(Cc)).astype(np.float32)
    out0, st0 = np.zeros((M, Cc), np.uint16), np.zeros((M, 2), np.float32)
    assert L.clite_layernorm_fwd(BF16, ptr(x), ptr(gamma), ptr(beta), 1e-12, ptr(out0), ptr(st0), M, Cc, p, 1234, 5, None) == 0
    a = from_bf16(out0)
    scale = np.float32(448.0 / (0.7 * np.abs(a).max()))
    q = np.full((M, Cc), 0x55, np.uint8)
    scales = np.array([scale, 1 / scale], np.float32)
    amax = np.zeros(SLOT, np.float32)
    out1, st1 = np.zeros((M, Cc), np.uint16), np.zeros((M, 2), np.float32)
    assert L.clite_layernorm_fwd_q8(BF16, ptr(x), ptr(gamma), ptr(beta), 1e-12, ptr(out1), ptr(st1), M, Cc, p, 1234, 5, ptr(q), ptr(scales), ptr(amax), None) == 0
    assert np.array_equal(out0, out1) and np.array_equal(st0, st1)
    assert amax.max() == np.abs(a).max() and not amax.reshape(16, 32)[:, 1:].any()
    ref = quantize_at(a, scale)
    assert np.array_equal(q & 0x7f, ref & 0x7f) and np.array_equal((q >> 7)[a != 0], (ref >> 7)[a != 0])
    amax2 = np.zeros(SLOT, np.float32)
    assert L.clite_layernorm_fwd_q8(BF16, ptr(x), ptr(gamma), ptr(beta), 1e-12, ptr(out1), ptr(st1), M, Cc, p, 1234, 5, None, None, ptr(amax2), None) == 0
    assert amax2.max() == np.abs(a).max()
    assert L.clite_layernorm_fwd_q8(BF16, ptr(x), ptr(gamma), ptr(beta), 1e-12, ptr(out1), ptr(st1), M, Cc, p, 1234, 5, ptr(q), None, None, None) == -1
    assert L.clite_layernorm_fwd_q8(F32, ptr(x), ptr(gamma), ptr(beta), 1e-12, ptr(out1), ptr(st1), M, Cc, p, 1234, 5, None, None, ptr(amax2), None) == -1


def e5m2_values():
    v = np.zeros(256, np.float32)
    for b in range(256):
        s_, e, m = b >> 7, (b >> 2) & 31, b & 3
        if e == 31:
            r = np.nan if m else np.inf
        elif e == 0:
            r = m * 2.0 ** -16
        else:
            r = (1 + m / 4) * 2.0 ** (e - 15)
        v[b] = -r if s_ else r
    return v


E5M2 = e5m2_values()


def quantize_e5m2_at(x, scale):
    """e5m2 codes of clamp(x * scale, +-57344), round to nearest even (numpy restatement of the producer-fused gradient quantiser)."""
    y = np.clip(x.astype(np.float32) * np.float32(scale), -57344, 57344).astype(np.float32)
    pos = E5M2[:124]                                          # 0 .. 57344 ascending (0x00..0x7b)
    idx = np.searchsorted(pos, np.abs(y), side="left").clip(1, 123)
    lo, hi = pos[idx - 1], pos[idx]
    dl, dh = np.abs(y) - lo, hi - np.abs(y)
    pick_hi = (dh < dl) | ((dh == dl) & (idx % 2 == 0))
    code = np.where(pick_hi, idx, idx - 1).astype(np.uint8)
    code = np.where(np.abs(y) == 0, 0, code).astype(np.uint8)
    return code | np.where(np.signbit(y), 0x80, 0).astype(np.uint8)


@pytest.mark.parametrize("M,Cc", [(70, 64), (33, 256)])
def test_bn_bwd_apply_writes_the_e5m2_copy_and_the_amax(M, Cc):
    """clite_bn_bwd_apply with clite_bn.fp8_* (ABI v11): the e5m2 codes of the STORED bf16 dy at the given scale, max |dy| in the slot, dy itself
    bit-identical to the plain call's."""
    from simlib import pack_relu_bits
    L = lib()
    rng = np.random.default_rng(M * 3 + Cc)
    y = bf16_round(rng.standard_normal((M, Cc)).astype(np.float32) * 2 + 0.5)
    dout = bf16_round(rng.standard_normal((M, Cc)).astype(np.float32) * 1e-3)
    act = rng.standard_normal((M, Cc)).astype(np.float32)
    bits = pack_relu_bits(act)
    yb, db = to_bf16(y), to_bf16(dout)
    gamma = (1 + 0.1 * rng.standard_normal(Cc)).astype(np.float32)
    beta = np.zeros(Cc, np.float32)
    stats = np.zeros((1, 3, Cc), np.float32)
    stats[0, 0], stats[0, 1] = y.sum(0), (y * y).sum(0)
    dz = dout * (act > 0)
    dstats = np.zeros((1, 3, Cc), np.float32)
    dstats[0, 0], dstats[0, 1] = dz.sum(0), (dz * (y - y.mean(0))).sum(0)
    L.clite_bn_bwd_apply.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 10

    def run(fp8):
        rm, rv = np.zeros(Cc, np.float32), np.ones(Cc, np.float32)
        p = Bn(M, Cc, ptr(stats), ptr(gamma), ptr(beta), ptr(rm), ptr(rv), 1, 0, 0.1, 1e-5, 0, 1, 3 * Cc, 0)
        if fp8 is not None:
            p.fp8_out, p.fp8_scale, p.fp8_amax = ptr(fp8[0]), ptr(fp8[1]), ptr(fp8[2])
        dy = np.zeros((M, Cc), np.uint16)
        assert L.clite_bn_bwd_apply(C.byref(p), BF16, ptr(db), None, ptr(bits), ptr(yb), ptr(dstats), ptr(dy), None, None, None, None) == 0
        return dy

    dy0 = run(None)
    a = from_bf16(dy0)
    scale = np.float32(448.0 / (0.5 * np.abs(a).max()))          # stale by 2x: e5m2's headroom (57344 / 448) absorbs it, nothing saturates
    q = np.full((M, Cc), 0x55, np.uint8)
    scales = np.array([scale, 1 / scale], np.float32)
    amax = np.zeros(SLOT, np.float32)
    dy1 = run((q, scales, amax))
    assert np.array_equal(dy0, dy1)
    assert amax.max() == np.abs(a).max() and not amax.reshape(16, 32)[:, 1:].any()
    ref = quantize_e5m2_at(a, scale)
    assert np.array_equal(q & 0x7f, ref & 0x7f) and np.array_equal((q >> 7)[a != 0], (ref >> 7)[a != 0])
    assert np.abs(E5M2[q]).max() <= 2 * 448.0 * 1.25 and np.isfinite(E5M2[q]).all()
    amax2 = np.zeros(SLOT, np.float32)
    run((None, None, amax2))
    assert amax2.max() == np.abs(a).max()


@pytest.mark.parametrize("N,H,W,Cc,K,R,pad", [(2, 8, 8, 64, 64, 3, 1), (3, 6, 6, 136, 128, 1, 0), (2, 9, 7, 128, 64, 3, 1)])
def test_conv_dgrad_fp8_bn_backward_form(N, H, W, Cc, K, R, pad):
    """clite_conv_dgrad_fp8 (ABI v11): dx = relu'(bits) * (dy8 (*) wt8) * dy_scales[1] * w_scales[1] with dy in e5m2 and the transposed weights
    [C][R][S][K] in e4m3 on v_mfma_scale_f32_32x32x64_f8f6f4 (A e5m2 / B e4m3: the operand map the hardware probe measured), stored as bf16, plus
    the two BatchNorm-backward reductions - against numpy on the de-quantised operands (products of an e5m2 and an e4m3 value are exact in f32).
    Ragged tiles (M, C not multiples of 128) included; the refusals (anything but that epilogue form; stride 2; K % 64)."""
    from simlib import pack_relu_bits
    from test_wavesim_igemm import conv_dgrad_ref
    L = lib()
    L.clite_fp8_quantize.argtypes = [C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.clite_conv_dgrad_fp8.argtypes = [C.c_void_p] * 7
    rng = np.random.default_rng(H * 5 + K)
    Ho, Wo = H + 2 * pad - R + 1, W + 2 * pad - R + 1
    cv = Conv(BF16, N, H, W, Cc, K, R, R, 1, pad, Ho, Wo)
    M = N * H * W
    w = (rng.standard_normal((K, R, R, Cc)) * 0.1).astype(np.float32)
    wt = np.ascontiguousarray(w.transpose(3, 1, 2, 0))          # [C][R][S][K]
    qwt, sw, _ = _quant(L, wt, F32)
    dy = (rng.standard_normal((N, Ho, Wo, K)) * 1e-3).astype(np.float32)
    sdy = np.float32(448.0 / np.abs(dy).max())
    qdy = quantize_e5m2_at(dy, sdy)
    sy = np.array([sdy, 1 / sdy], np.float32)
    act = rng.standard_normal((M, Cc)).astype(np.float32)
    bits = pack_relu_bits(act)
    y = bf16_round(rng.standard_normal((M, Cc)).astype(np.float32) + 3.0)
    yb = to_bf16(y)
    Rr = 4
    fstats = np.zeros((Rr, 3, Cc), np.float32)
    fstats[:, 0] = y.sum(0) / Rr
    mean = fstats[:, 0].sum(0) / M
    out = np.zeros((M, Cc), np.uint16)
    dst = np.zeros((Rr, 3, Cc), np.float32)

    def mk(**kw):
        ep = make_ep(out, Cc, colsum=dst, relu_bits=bits, **kw)
        ep.colsum_replicas, ep.colsum_stride = Rr, 3 * Cc
        ep.bn_y, ep.bn_stats, ep.bn_replicas, ep.bn_rstride, ep.bn_inv_count = ptr(yb), ptr(fstats), Rr, 3 * Cc, 1.0 / M
        return ep

    assert L.clite_conv_dgrad_fp8(ptr(qdy), ptr(qwt), C.byref(cv), ptr(sy), ptr(sw), C.byref(mk()), None) == 0
    wdeq = np.ascontiguousarray(E4M3[qwt].transpose(3, 1, 2, 0))          # back to [K][R][S][C]
    g = conv_dgrad_ref(E5M2[qdy], wdeq, (N, H, W, Cc), 1, pad).reshape(M, Cc) * (sy[1] * sw[1])
    v = g * (act > 0)
    got = from_bf16(out)
    assert np.abs(got - v).max() <= 6e-3 * np.abs(v).max()
    d = dst.sum(0)
    assert np.abs(d[0] - got.sum(0)).max() <= 1e-3 * max(np.abs(got.sum(0)).max(), 1e-9)
    assert np.abs(d[1] - (got * (y - mean)).sum(0)).max() <= 2e-3 * np.abs((got * (y - mean)).sum(0)).max()
    # and the format costs what e5m2 x e4m3 costs: ~10 % of the gradient's scale on N(0, 1e-3) x N(0, 0.1) operands
    exact = conv_dgrad_ref(dy, w, (N, H, W, Cc), 1, pad).reshape(M, Cc) * (act > 0)
    assert np.abs(got - exact).max() <= 0.2 * np.abs(exact).max()
    res = np.zeros((M, Cc), np.uint16)
    assert L.clite_conv_dgrad_fp8(ptr(qdy), ptr(qwt), C.byref(cv), ptr(sy), ptr(sw), C.byref(mk(residual=res)), None) == -1
    assert L.clite_conv_dgrad_fp8(ptr(qdy), ptr(qwt), C.byref(cv), ptr(sy), ptr(sw), C.byref(make_ep(out, Cc)), None) == -1
    cv2 = Conv(BF16, N, H, W, Cc, K, R, R, 2, pad, (H + 2 * pad - R) // 2 + 1, (W + 2 * pad - R) // 2 + 1)
    assert L.clite_conv_dgrad_fp8(ptr(qdy), ptr(qwt), C.byref(cv2), ptr(sy), ptr(sw), C.byref(mk()), None) == -1
    cv3 = Conv(BF16, N, H, W, Cc, 48, R, R, 1, pad, Ho, Wo)
    assert L.clite_conv_dgrad_fp8(ptr(qdy), ptr(qwt), C.byref(cv3), ptr(sy), ptr(sw), C.byref(mk()), None) == -1


def test_grouped_weight_quantiser_and_scale_update():
    """clite_fp8_quantize_group: three tensors of one bf16 arena (one longer than a workgroup's 8192-element chunk, one all zero) get the
    stand-alone quantiser's codes and scales, untouched bytes between them stay untouched; clite_fp8_scale_update turns recorded amaxes into
    {448 / amax, amax / 448}, zeroes them, keeps the scales of a slot that recorded nothing and makes a non-finite amax's scales NaN."""
    L = lib()
    L.clite_fp8_quantize.argtypes = [C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.clite_fp8_quantize_group.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.clite_fp8_scale_update.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    rng = np.random.default_rng(11)
    total = 64 + 20000 + 128 + 512 + 64
    arena = bf16_round((rng.standard_normal(total) * np.exp(rng.standard_normal(total))).astype(np.float32))
    spans = [(64, 20000), (64 + 20000 + 128, 512), (total - 64, 64)]
    arena[spans[2][0]:] = 0.0
    buf = to_bf16(arena)
    items = np.array([v for o, n in spans for v in (o, n)], np.uint64)
    table = np.array([(i << 12) | c for i, (o, n) in enumerate(spans) for c in range((n + 8191) // 8192)], np.uint32)
    q = np.full(total, 0x55, np.uint8)
    amax, scales = np.full(3, 7.0, np.float32), np.zeros((3, 2), np.float32)
    partial = np.full(len(table), 0x7F7F7F7F, np.uint32)            # scratch: whatever it held before
    assert L.clite_fp8_quantize_group(ptr(buf), ptr(items), ptr(table), 3, len(table), ptr(partial), ptr(amax), ptr(scales), ptr(q), None) == 0
    covered = np.zeros(total, bool)
    for i, (o, n) in enumerate(spans):
        x = arena[o:o + n].reshape(-1, 8)
        q1, s1, a1 = _quant(L, x, BF16)
        assert np.array_equal(q[o:o + n], q1.reshape(-1)) and np.array_equal(scales[i], s1) and amax[i] == a1[0], i
        covered[o:o + n] = True
    assert (q[~covered] == 0x55).all()
    am = np.zeros((4, SLOT), np.float32)
    am[0, 0], am[0, 5 * 32] = 1.5, 2.0                             # a slot's value is the maximum of its words
    am[2, 32], am[3, 15 * 32], am[3, 0] = np.inf, np.nan, 7.0
    sc = np.tile(np.array([3.0, 5.0], np.float32), (4, 1))
    assert L.clite_fp8_scale_update(ptr(am), ptr(sc), 4, None) == 0
    assert np.allclose(sc[0], [224.0, 2.0 / 448.0]) and np.array_equal(sc[1], [3.0, 5.0]) and np.isnan(sc[2]).all() and np.isnan(sc[3]).all()
    assert not am.view(np.uint32).any()


@pytest.mark.parametrize("N,H,W,Cc,K,R,st,pad", [(2, 8, 8, 64, 64, 3, 1, 1), (3, 6, 6, 288, 272, 1, 1, 0), (2, 9, 7, 32, 64, 3, 2, 1), (2, 40, 40, 48, 32, 1, 1, 0)])
def test_grouped_weight_gradient_on_fp8_operands(N, H, W, Cc, K, R, st, pad):
    """clite_wgrad_group kind 2 (ABI v12, BASELINE configs[4]): dW += dy8^T x8 * a_scales[1] * b_scales[1] with dy in e5m2 and x in e4m3 — both operand
    images have the contraction index (the pixel) as the slow one, the fragments come out of them through ds_read_b64_tr_b8 (the layout
    tools/micro/tr8_probe.hip measured on the hardware, emulated lane-accurately), 64 pixels per v_mfma_scale_f32_32x32x64_f8f6f4 — against numpy on the
    de-quantised operands (an e5m2 times an e4m3 value is exact in f32). Ragged tiles both ways, a windowed and a strided member, a member whose
    pixel range is cut into two k-chunks (3200 pixels = 50 tiles of 64 under CLITE_WGRAD_SHORTK's 32-tile chunks), `+=` and the CLITE_WGRAD_ZEROED store form,
    and a row_scale."""
    from test_wavesim_igemm import conv_wgrad_ref
    L = lib()
    L.clite_fp8_quantize.argtypes = [C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.clite_wgrad_group.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    L.clite_wgrad_group_workspace.argtypes = [C.c_int, C.c_int64, C.c_void_p]

    class Item(C.Structure):
        _fields_ = [("kind", C.c_int32), ("a", C.c_void_p), ("b", C.c_void_p), ("out", C.c_void_p), ("cv", Conv),
                    ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("lda", C.c_int32), ("ldb", C.c_int32), ("ldc", C.c_int32),
                    ("a_scales", C.c_void_p), ("b_scales", C.c_void_p), ("row_scale", C.c_void_p)]
    rng = np.random.default_rng(H * 7 + K)
    Ho, Wo = (H + 2 * pad - R) // st + 1, (W + 2 * pad - R) // st + 1
    cv = Conv(BF16, N, H, W, Cc, K, R, R, st, pad, Ho, Wo)
    x = rng.standard_normal((N, H, W, Cc)).astype(np.float32)
    qx, sx, _ = _quant(L, x, F32)
    dy = (rng.standard_normal((N, Ho, Wo, K)) * 1e-3).astype(np.float32)
    sdy = np.float32(448.0 / np.abs(dy).max())
    qdy = quantize_e5m2_at(dy, sdy)
    sy = np.array([sdy, 1 / sdy], np.float32)
    ref = conv_wgrad_ref(E5M2[qdy], E4M3[qx], (K, R, R, Cc), st, pad) * (sy[1] * sx[1])
    nb = C.c_uint64(0)
    assert L.clite_wgrad_group_workspace(1, 4096, C.byref(nb)) == 0
    ws_dev, ws_host = np.zeros(nb.value, np.uint8), np.zeros(nb.value, np.uint8)
    for kind, init, rs in ((2 | 0x400, 1.0, None), (2 | 0x200 | 0x400, 0.0, None), (2, 0.25, (1 + 0.5 * rng.standard_normal(K)).astype(np.float32))):
        dw = np.full((K, R, R, Cc), init, np.float32)
        it = Item()
        it.kind, it.a, it.b, it.out, it.cv = kind, ptr(qdy).value, ptr(qx).value, ptr(dw).value, cv
        it.a_scales, it.b_scales = ptr(sy).value, ptr(sx).value
        it.row_scale = ptr(rs).value if rs is not None else None
        arr = (Item * 1)(it)
        assert L.clite_wgrad_group(BF16, arr, 1, ptr(ws_dev), ptr(ws_host), nb.value, None) == 0
        want = init + (ref if rs is None else ref * rs.reshape(-1, 1, 1, 1))
        assert np.abs(dw - want).max() <= 2e-5 * max(np.abs(want).max(), 1e-9), np.abs(dw - want).max()
    # what the formats cost on these operands (e5m2 carries two mantissa bits): within 15 % of the exact gradient's scale
    exact = conv_wgrad_ref(dy, x, (K, R, R, Cc), st, pad)
    assert np.abs(ref - exact).max() <= 0.15 * np.abs(exact).max()
    it.a_scales = None
    assert L.clite_wgrad_group(BF16, (Item * 1)(it), 1, ptr(ws_dev), ptr(ws_host), nb.value, None) == -1          # no scales: refused
    it.a_scales = ptr(sy).value
    assert L.clite_wgrad_group(BF16, (Item * 1)(it), 1, None, None, 0, None) == -1                                  # members one by one: no fp8 form
