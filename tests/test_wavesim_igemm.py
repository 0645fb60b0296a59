"""CPU-side check of the implicit-GEMM engine's index math: the same csrc/*.hip sources compiled against the
lane-accurate wave simulator (tests/wavesim/), compared with numpy on small problems. Runs without a GPU."""
import ctypes as C

import numpy as np
import pytest

from simlib import Conv, bf16_round, from_bf16, lib, make_ep, pack_relu_bits, ptr, to_bf16

BF16, F32 = 0, 1


@pytest.fixture(autouse=True, params=[1, 2, 3], ids=["w128", "w256x128", "w256"])
def tile_policy(request):
    """Every bf16 case runs once per wide-tile instantiation family (include/clite.h: clite_set_tile_policy forces the 128 x 128, 256 x 128 or
    256 x 256 8-wave kernel wherever one exists; launches it does not cover — N <= 64, f32 — stay on the 4-wave kernels). f32 cases are
    independent of the policy and run once."""
    dtype = request.node.callspec.params.get("dtype", BF16) if hasattr(request.node, "callspec") else BF16
    if dtype == F32 and request.param != 1:
        pytest.skip("f32 launches do not depend on the tile policy")
    if request.node.get_closest_marker("policy_independent") and request.param != 1:
        pytest.skip("this launch does not depend on the tile policy")
    assert lib().clite_set_tile_policy(request.param) == 0
    yield request.param
    lib().clite_set_tile_policy(0)


def _prep(x, dtype):
    """returns (exact value array the kernel sees, buffer to pass)"""
    if dtype == BF16:
        xr = bf16_round(x)
        return xr, to_bf16(xr)
    x = np.ascontiguousarray(x, np.float32)
    return x, x


def outbuf_like(dtype, shape):
    return np.zeros(shape, np.uint16 if dtype == BF16 else np.float32)


def _close(got, ref, rel=2e-3):
    assert np.abs(got - ref).max() <= rel * max(np.abs(ref).max(), 1e-6)


@pytest.mark.parametrize("dtype,M,N,K", [(BF16, 128, 128, 64), (BF16, 200, 136, 104), (BF16, 300, 64, 32), (BF16, 70, 1000, 200),
                                          (F32, 200, 136, 104), (F32, 300, 64, 32)])
def test_gemm_nt_epilogue(dtype, M, N, K):
    rng = np.random.default_rng(M + N)
    A, Ab = _prep(rng.standard_normal((M, K), dtype=np.float32), dtype)
    B, Bb = _prep(rng.standard_normal((N, K), dtype=np.float32), dtype)
    R, Rb = _prep(rng.standard_normal((M, N), dtype=np.float32), dtype)
    bias = rng.standard_normal(N).astype(np.float32)
    out = np.zeros((M, N), np.float32)
    pre = np.zeros((M, N), np.uint16 if dtype == BF16 else np.float32)
    ep = make_ep(out, N, out_f32=True, bias=bias, act=2, preact=pre, residual=Rb)
    assert lib().clite_gemm_nt(ptr(Ab), K, ptr(Bb), K, M, N, K, dtype, C.byref(ep), None) == 0
    z = A @ B.T + bias
    from scipy.special import erf
    ref = 0.5 * z * (1 + erf(z / np.sqrt(2))) + R
    _close(out, ref)
    _close(from_bf16(pre) if dtype == BF16 else pre, z, 6e-3)


@pytest.mark.parametrize("dtype,M,N,K", [(BF16, 128, 128, 64), (BF16, 200, 136, 104), (BF16, 300, 64, 40), (F32, 200, 136, 104), (F32, 300, 64, 40)])
def test_gemm_nn_dact(dtype, M, N, K):
    rng = np.random.default_rng(M + K)
    A, Ab = _prep(rng.standard_normal((M, K), dtype=np.float32), dtype)
    B, Bb = _prep(rng.standard_normal((K, N), dtype=np.float32), dtype)
    aux, auxb = _prep(rng.standard_normal((M, N), dtype=np.float32), dtype)
    out = np.zeros((M, N), np.float32)
    ep = make_ep(out, N, out_f32=True, dact_aux=auxb, dact=1)
    assert lib().clite_gemm_nn(ptr(Ab), K, ptr(Bb), N, M, N, K, dtype, C.byref(ep), None) == 0
    _close(out, (A @ B) * (aux > 0))


@pytest.mark.parametrize("kind,M,N,K", [("nt", 40, 136, 256), ("nn", 128, 200, 320), ("nt", 130, 64, 288)])
def test_gemm_splitk_workspace_form(kind, M, N, K):
    """clite_epilogue.splitk_ws: a GEMM of few output tiles runs as split-K into a zeroed f32 workspace plus splitk_finish_kernel. Same
    epilogue semantics as the fused path: bias, ReLU with pre-activation store / activation-derivative factor, residual, bf16 store,
    column statistics of the stored values, strided output rows."""
    rng = np.random.default_rng(M + N + K)
    A, Ab = _prep(rng.standard_normal((M, K), dtype=np.float32), BF16)
    Bm, Bb = _prep(rng.standard_normal((N, K) if kind == "nt" else (K, N), dtype=np.float32), BF16)
    R, Rb = _prep(rng.standard_normal((M, 2 * N), dtype=np.float32), BF16)
    bias = rng.standard_normal(N).astype(np.float32)
    ldc = 2 * N                                   # output rows scattered through ldc, as the BERT pooler's input gradient is
    out = np.zeros((M, ldc), np.uint16); pre = np.zeros((M, ldc), np.uint16)
    colsum = np.zeros((2, N), np.float32)
    ws = np.zeros((M, N), np.float32)
    z = A @ (Bm.T if kind == "nt" else Bm)
    if kind == "nt":
        ep = make_ep(out, ldc, bias=bias, act=1, preact=pre, residual=Rb, colsum=colsum, alpha=0.5)
        ep.splitk_ws = ptr(ws)
        assert lib().clite_gemm_nt(ptr(Ab), K, ptr(Bb), K, M, N, K, BF16, C.byref(ep), None) == 0
        zz = 0.5 * z + bias
        ref = np.maximum(zz, 0) + R[:, :N]
        _close(from_bf16(pre)[:, :N], zz, 6e-3)
    else:
        aux, auxb = _prep(rng.standard_normal((M, 2 * N), dtype=np.float32), BF16)
        ep = make_ep(out, ldc, dact_aux=auxb, dact=1, residual=Rb, colsum=colsum)
        ep.splitk_ws = ptr(ws)
        assert lib().clite_gemm_nn(ptr(Ab), K, ptr(Bb), N, M, N, K, BF16, C.byref(ep), None) == 0
        ref = z * (aux[:, :N] > 0) + R[:, :N]
    assert np.abs(ws).max() > 0                   # the split-K form ran (the fused path never touches the workspace)
    got = from_bf16(out)
    _close(got[:, :N], ref, 6e-3)
    assert not got[:, N:].any()
    _close(colsum[0], got[:, :N].sum(0), 1e-4)
    _close(colsum[1], (got[:, :N] ** 2).sum(0), 1e-4)


@pytest.mark.parametrize("dtype,N,H,W,Cc,K,R,S,st,pad", [(BF16, 2, 8, 8, 32, 64, 3, 3, 1, 1), (BF16, 3, 6, 6, 136, 64, 1, 1, 1, 0), (F32, 2, 9, 7, 64, 32, 3, 3, 2, 1),
                                                         (BF16, 5, 10, 10, 136, 64, 1, 1, 1, 0), (F32, 3, 11, 11, 40, 32, 1, 1, 1, 0)])
def test_conv_dgrad_bn_backward_reductions(dtype, N, H, W, Cc, K, R, S, st, pad):
    """Epilogue form used by the ResNet backward (conv dgrad only): v = (acc + residual) * (aux > 0) stored, and
    colsum <- (sum v, sum v*(y - mean)) with mean taken from replicated forward statistics (clite_epilogue.bn_y, mask_after_residual).
    The last two cases have more tiles than the simulator build's 4 resident slots (CLITE_BN_SLOTS), i.e. they run the row-range
    persistent form of igemm_dma_bn_kernel: several tiles per workgroup, the last one partial."""
    rng = np.random.default_rng(H * 3 + K)
    Ho = (H + 2 * pad - R) // st + 1
    Wo = (W + 2 * pad - S) // st + 1
    cv = Conv(dtype, N, H, W, Cc, K, R, S, st, pad, Ho, Wo)
    M = N * H * W
    w, wb = _prep(rng.standard_normal((K, R, S, Cc), dtype=np.float32) * 0.2, dtype)
    dy, dyb = _prep(rng.standard_normal((N, Ho, Wo, K), dtype=np.float32), dtype)
    aux, auxb = _prep(rng.standard_normal((M, Cc), dtype=np.float32), dtype)
    res, resb = _prep(rng.standard_normal((M, Cc), dtype=np.float32), dtype)
    y, yb = _prep(rng.standard_normal((M, Cc), dtype=np.float32) + 3.0, dtype)
    Rr = 4
    fstats = np.zeros((Rr, 3, Cc), np.float32)
    fstats[:, 0] = y.sum(0) / Rr * (1 + 0.1 * rng.standard_normal((Rr, 1)).astype(np.float32))      # replicas only sum to the total
    fstats[:, 0] += (y.sum(0) - fstats[:, 0].sum(0)) / Rr
    mean = fstats[:, 0].sum(0) / M
    for after in (1, 0):
        out = np.zeros((M, Cc), np.float32)
        dst = np.zeros((Rr, 3, Cc), np.float32)
        ep = make_ep(out, Cc, out_f32=True, dact_aux=auxb, dact=1, residual=resb, colsum=dst)
        ep.colsum_replicas, ep.colsum_stride = Rr, 3 * Cc
        ep.bn_y, ep.bn_stats, ep.bn_replicas, ep.bn_rstride, ep.bn_inv_count, ep.mask_after_residual = ptr(yb), ptr(fstats), Rr, 3 * Cc, 1.0 / M, after
        assert lib().clite_conv_dgrad(ptr(dyb), ptr(wb), C.byref(cv), C.byref(ep), None) == 0
        g = conv_dgrad_ref(dy, w, (N, H, W, Cc), st, pad).reshape(M, Cc)
        v = (g + res) * (aux > 0) if after else g * (aux > 0) + res
        _close(out, v)
        d = dst.sum(0)
        _close(d[0], v.sum(0), 5e-3)
        _close(d[1], (v * (y - mean)).sum(0), 5e-3)
        assert not d[2].any()
        # the same launch with the relu' mask as packed bits (clite_epilogue.relu_bits): identical output and reductions
        out2 = np.zeros((M, Cc), np.float32)
        dst2 = np.zeros((Rr, 3, Cc), np.float32)
        bits = pack_relu_bits(aux)
        ep2 = make_ep(out2, Cc, out_f32=True, residual=resb, colsum=dst2, relu_bits=bits)
        ep2.colsum_replicas, ep2.colsum_stride = Rr, 3 * Cc
        ep2.bn_y, ep2.bn_stats, ep2.bn_replicas, ep2.bn_rstride, ep2.bn_inv_count, ep2.mask_after_residual = ptr(yb), ptr(fstats), Rr, 3 * Cc, 1.0 / M, after
        assert lib().clite_conv_dgrad(ptr(dyb), ptr(wb), C.byref(cv), C.byref(ep2), None) == 0
        assert np.array_equal(out2, out)
        _close(dst2.sum(0), d, 1e-5)
        ep2.dact_aux, ep2.dact = ptr(auxb), 1                  # both forms of the mask at once: refused
        assert lib().clite_conv_dgrad(ptr(dyb), ptr(wb), C.byref(cv), C.byref(ep2), None) == -1


@pytest.mark.parametrize("dtype,M,N,K", [(BF16, 128, 128, 64), (BF16, 72, 136, 300), (BF16, 256, 8, 1000), (F32, 72, 136, 300)])
def test_gemm_tn_atomic(dtype, M, N, K):
    rng = np.random.default_rng(M + K)
    A, Ab = _prep(rng.standard_normal((K, M), dtype=np.float32), dtype)
    B, Bb = _prep(rng.standard_normal((K, N), dtype=np.float32), dtype)
    out = np.ones((M, N), np.float32)
    ep = make_ep(out, N, out_f32=True, atomic=True)
    assert lib().clite_gemm_tn(ptr(Ab), M, ptr(Bb), N, M, N, K, dtype, C.byref(ep), None) == 0
    _close(out, 1 + A.T @ B)


def conv_ref(x, w, stride, pad):
    N, H, W, Cc = x.shape
    K, R, S, _ = w.shape
    Ho = (H + 2 * pad - R) // stride + 1
    Wo = (W + 2 * pad - S) // stride + 1
    xp = np.zeros((N, H + 2 * pad, W + 2 * pad, Cc), np.float32)
    xp[:, pad:pad + H, pad:pad + W] = x
    y = np.zeros((N, Ho, Wo, K), np.float32)
    for r in range(R):
        for s in range(S):
            y += xp[:, r:r + stride * Ho:stride, s:s + stride * Wo:stride, :] @ w[:, r, s, :].T
    return y


def conv_dgrad_ref(dy, w, xshape, stride, pad):
    N, H, W, Cc = xshape
    K, R, S, _ = w.shape
    _, Ho, Wo, _ = dy.shape
    dxp = np.zeros((N, H + 2 * pad, W + 2 * pad, Cc), np.float32)
    for r in range(R):
        for s in range(S):
            dxp[:, r:r + stride * Ho:stride, s:s + stride * Wo:stride, :] += dy @ w[:, r, s, :]
    return dxp[:, pad:pad + H, pad:pad + W]


def conv_wgrad_ref(dy, x, wshape, stride, pad):
    N, H, W, Cc = x.shape
    K, R, S, _ = wshape
    _, Ho, Wo, _ = dy.shape
    xp = np.zeros((N, H + 2 * pad, W + 2 * pad, Cc), np.float32)
    xp[:, pad:pad + H, pad:pad + W] = x
    dw = np.zeros(wshape, np.float32)
    for r in range(R):
        for s in range(S):
            patch = xp[:, r:r + stride * Ho:stride, s:s + stride * Wo:stride, :]
            dw[:, r, s, :] = dy.reshape(-1, K).T @ patch.reshape(-1, Cc)
    return dw


CASES = [(2, 8, 8, 32, 64, 3, 3, 1, 1), (2, 9, 7, 64, 32, 3, 3, 2, 1), (3, 6, 6, 64, 128, 1, 1, 1, 0),
         (2, 8, 8, 32, 64, 1, 1, 2, 0), (1, 10, 10, 32, 160, 3, 3, 1, 1), (2, 9, 7, 128, 136, 1, 1, 2, 0), (2, 6, 6, 136, 64, 1, 1, 1, 0), (2, 6, 6, 32, 136, 1, 1, 1, 0)]


@pytest.mark.parametrize("dtype,N,H,W,Cc,K,R,S,st,pad", [(BF16,) + c for c in CASES] + [(F32,) + CASES[1], (F32,) + CASES[3]])
def test_conv_fwd_dgrad_wgrad(dtype, N, H, W, Cc, K, R, S, st, pad):
    rng = np.random.default_rng(H * W + K)
    Ho = (H + 2 * pad - R) // st + 1
    Wo = (W + 2 * pad - S) // st + 1
    cv = Conv(dtype, N, H, W, Cc, K, R, S, st, pad, Ho, Wo)
    x, xb = _prep(rng.standard_normal((N, H, W, Cc), dtype=np.float32), dtype)
    w, wb = _prep(rng.standard_normal((K, R, S, Cc), dtype=np.float32) * 0.2, dtype)
    dy, dyb = _prep(rng.standard_normal((N, Ho, Wo, K), dtype=np.float32), dtype)
    y = np.zeros((N, Ho, Wo, K), np.float32)
    csr = np.zeros((4, 3, K), np.float32)          # 4 replicated accumulators, stride 3*K
    ep = make_ep(y, K, out_f32=True, colsum=csr)
    ep.colsum_replicas, ep.colsum_stride = 4, 3 * K
    assert lib().clite_conv_fwd(ptr(xb), ptr(wb), C.byref(cv), C.byref(ep), None) == 0
    ref = conv_ref(x, w, st, pad)
    cs = csr.sum(0)
    assert not csr[:, 2].any()
    _close(y, ref)
    _close(cs[0], ref.reshape(-1, K).sum(0))
    _close(cs[1], (ref.reshape(-1, K) ** 2).sum(0))
    dx = np.zeros((N, H, W, Cc), np.float32)
    assert lib().clite_conv_dgrad(ptr(dyb), ptr(wb), C.byref(cv), C.byref(make_ep(dx, Cc, out_f32=True)), None) == 0
    _close(dx, conv_dgrad_ref(dy, w, x.shape, st, pad))
    dw = np.zeros((K, R, S, Cc), np.float32)
    assert lib().clite_conv_wgrad(ptr(dyb), ptr(xb), C.byref(cv), ptr(dw), None) == 0
    _close(dw, conv_wgrad_ref(dy, x, w.shape, st, pad))
    # in-place accumulation dx += dgrad (residual == out, both in the call's dtype): the strided 1x1 case takes the dense-GEMM +
    # scatter-add path
    r0, r0b = _prep(rng.standard_normal((N, H, W, Cc), dtype=np.float32), dtype)
    buf = r0b.copy()
    want = r0 + conv_dgrad_ref(dy, w, x.shape, st, pad)
    assert lib().clite_conv_dgrad(ptr(dyb), ptr(wb), C.byref(cv), C.byref(make_ep(buf, Cc, out_f32=False, residual=buf)), None) == 0
    _close(from_bf16(buf) if dtype == BF16 else buf, want, 8e-3 if dtype == BF16 else 2e-3)


@pytest.mark.policy_independent
@pytest.mark.parametrize("N,H,W,Cc,K,st", [(5, 16, 16, 32, 136, 1), (10, 32, 32, 32, 64, 2), (3, 21, 20, 96, 128, 1)])
def test_conv_fwd_1x1_row_range_persistent_form(N, H, W, Cc, K, st):
    """The plain FORWARD epilogue in the row-range persistent kernel (igemm_dma_bn_kernel / igemm_epilogue_bn FORM 4, round 4): 1 x 1 convolutions with
    a bf16 output, column statistics and more than twice as many tiles as resident slots (the simulator build has 4) leave one-tile-per-workgroup
    launches for it - several tiles per workgroup, the last one partial, ragged column tiles, the 256 x 64 tile (K <= 64 outputs) and a stride-2
    gather. Output = bf16(conv), statistics = sums of the STORED values and their squares, over replicated accumulators."""
    assert lib().clite_set_tile_policy(0) == 0
    rng = np.random.default_rng(H + K)
    Ho, Wo = (H - 1) // st + 1, (W - 1) // st + 1
    cv = Conv(BF16, N, H, W, Cc, K, 1, 1, st, 0, Ho, Wo)
    x, xb = _prep(rng.standard_normal((N, H, W, Cc), dtype=np.float32), BF16)
    w, wb = _prep(rng.standard_normal((K, 1, 1, Cc), dtype=np.float32) * 0.2, BF16)
    y = np.zeros((N, Ho, Wo, K), np.uint16)
    csr = np.zeros((4, 3, K), np.float32)
    ep = make_ep(y, K, colsum=csr)
    ep.colsum_replicas, ep.colsum_stride = 4, 3 * K
    assert lib().clite_conv_fwd(ptr(xb), ptr(wb), C.byref(cv), C.byref(ep), None) == 0
    ref = conv_ref(x, w, st, 0)
    got = from_bf16(y)
    assert np.abs(got - ref).max() <= 6e-3 * np.abs(ref).max()
    cs = csr.sum(0)
    assert not csr[:, 2].any()
    _close(cs[0], got.reshape(-1, K).sum(0), 1e-4)
    _close(cs[1], (got.reshape(-1, K) ** 2).sum(0), 1e-4)


@pytest.mark.parametrize("dtype,N,H,W,Cc,K", [(BF16, 2, 8, 8, 64, 64), (BF16, 1, 6, 10, 128, 64), (F32, 2, 4, 6, 64, 128)])
def test_conv_dgrad_stride2_parity_classes(dtype, N, H, W, Cc, K):
    """The four input-parity classes of a 3x3 / stride-2 / pad-1 dgrad (clite_conv_dgrad_s2class, each a stride-1 dgrad over the
    [N][H/2][W/2] sub-grid with its own taps, scattered by RowMap) together equal the full dgrad."""
    rng = np.random.default_rng(H * 7 + K)
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    cv = Conv(dtype, N, H, W, Cc, K, 3, 3, 2, 1, Ho, Wo)
    w, _ = _prep(rng.standard_normal((K, 3, 3, Cc), dtype=np.float32) * 0.2, dtype)
    dy, dyb = _prep(rng.standard_normal((N, Ho, Wo, K), dtype=np.float32), dtype)
    dx = np.full((N, H, W, Cc), 7.0, np.float32)          # every element must be overwritten by exactly one class
    for ph in (0, 1):
        for pw in (0, 1):
            sub = np.ascontiguousarray(w[:, (ph + 1) & 1::2, (pw + 1) & 1::2, :])
            _, subb = _prep(sub, dtype)
            assert lib().clite_conv_dgrad_s2class(ptr(dyb), ptr(subb), C.byref(cv), ph, pw, C.byref(make_ep(dx, Cc, out_f32=True)), None) == 0
    _close(dx, conv_dgrad_ref(dy, w, (N, H, W, Cc), 2, 1))


@pytest.mark.parametrize("dtype", [BF16, F32])
def test_stem_conv7x7(dtype):
    """7x7/2 pad-3 stem expressed as a 7x1 window over 32 virtual channels of the pre-padded NHWC4 image."""
    rng = np.random.default_rng(7)
    N, H, W = 2, 20, 18
    Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    Hp, Wp = H + 6, W + 6 + 2
    Wp += Wp % 2
    img = rng.standard_normal((N, 3, H, W), dtype=np.float32)
    w = (rng.standard_normal((64, 7, 7, 3), dtype=np.float32) * 0.1)
    xpad = np.zeros((N, Hp, Wp, 4), np.uint16 if dtype == BF16 else np.float32)
    assert lib().clite_image_to_nhwc4(dtype, ptr(img), ptr(xpad), N, H, W, 3, Hp, Wp, None) == 0
    wv = np.zeros((64, 7, 8, 4), np.uint16 if dtype == BF16 else np.float32)
    assert lib().clite_stem_pack(ptr(w), ptr(wv), dtype, None) == 0
    imgr = bf16_round(img) if dtype == BF16 else img
    wr = bf16_round(w) if dtype == BF16 else w
    y = np.zeros((N, Ho, Wo, 64), np.float32)
    assert lib().clite_stem_fwd(ptr(xpad), ptr(wv), dtype, N, Hp, Wp, Ho, Wo, C.byref(make_ep(y, 64, out_f32=True)), None) == 0
    ref = conv_ref(imgr.transpose(0, 2, 3, 1), wr, 2, 3)
    _close(y, ref)
    dy, dyb = _prep(rng.standard_normal((N, Ho, Wo, 64), dtype=np.float32), dtype)
    dwv = np.zeros((64, 7, 8, 4), np.float32)
    assert lib().clite_stem_wgrad(ptr(dyb), ptr(xpad), dtype, N, Hp, Wp, Ho, Wo, ptr(dwv), None) == 0
    dw = np.ones((64, 7, 7, 3), np.float32)
    assert lib().clite_stem_unpack_grad(ptr(dwv), ptr(dw), None) == 0
    _close(dw, 1 + conv_wgrad_ref(dy, imgr.transpose(0, 2, 3, 1), w.shape, 2, 3))


@pytest.mark.policy_independent
@pytest.mark.parametrize("N,H,W,extra_rows", [(2, 10, 32, 0), (3, 7, 64, 2)])
def test_stem_weight_gradient_patch_resident(N, H, W, extra_rows):
    """clite_stem_wgrad_patch (ABI v11, conv_patch.hip): dw [64][7][7][3] f32 += over strips of two output rows, both operands read from per-strip LDS
    images by transposed reads (the x patch as the 9 input rows lie in memory), per-workgroup slabs + the reduction that drops the packed layout's
    padding columns. Against the numpy reference and clite_stem_wgrad + clite_stem_unpack_grad on the same operands; an odd number of output rows
    (a one-row last strip per image), a padded image taller than the minimum, several strips per workgroup (the simulator build runs 2 workgroups);
    accumulation into a non-zero dw; the refusals (Wo % 16, f32, workspace too small: return 1, nothing launched)."""
    assert lib().clite_set_tile_policy(0) == 0
    rng = np.random.default_rng(H + W)
    Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    Hp, Wp = H + 6 + extra_rows, W + 6 + 2
    img = rng.standard_normal((N, 3, H, W), dtype=np.float32)
    xpad = np.full((N, Hp, Wp, 4), 0x7FC0, np.uint16)          # the padding arrives as NaNs where image_to_nhwc4 does not write: it writes everything
    assert lib().clite_image_to_nhwc4(BF16, ptr(img), ptr(xpad), N, H, W, 3, Hp, Wp, None) == 0
    imgr = bf16_round(img)
    dy, dyb = _prep(rng.standard_normal((N, Ho, Wo, 64), dtype=np.float32), BF16)
    nb = C.c_uint64(0)
    assert lib().clite_conv_wgrad_patch_workspace(C.byref(nb)) == 0
    ws = np.full(nb.value // 4, np.nan, np.float32)          # scratch arrives dirty
    dw = np.full((64, 7, 7, 3), 0.5, np.float32)
    L = lib()
    L.clite_stem_wgrad_patch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    assert L.clite_stem_wgrad_patch(ptr(dyb), ptr(xpad), BF16, N, Hp, Wp, Ho, Wo, ptr(dw), ptr(ws), nb.value, None) == 0
    ref = conv_wgrad_ref(dy, imgr.transpose(0, 2, 3, 1), (64, 7, 7, 3), 2, 3)
    _close(dw - 0.5, ref, 2e-3)
    dwv = np.zeros((64, 7, 8, 4), np.float32)
    assert L.clite_stem_wgrad(ptr(dyb), ptr(xpad), BF16, N, Hp, Wp, Ho, Wo, ptr(dwv), None) == 0
    dw2 = np.zeros((64, 7, 7, 3), np.float32)
    assert L.clite_stem_unpack_grad(ptr(dwv), ptr(dw2), None) == 0
    _close(dw - 0.5, dw2, 2e-3)
    assert L.clite_stem_wgrad_patch(ptr(dyb), ptr(xpad), BF16, N, Hp, Wp, Ho, Wo, ptr(dw), ptr(ws), 1024, None) == 1
    assert L.clite_stem_wgrad_patch(ptr(dyb), ptr(xpad), F32, N, Hp, Wp, Ho, Wo, ptr(dw), ptr(ws), nb.value, None) == 1
    Wo2 = Wo - 4          # a width that is not a multiple of 16
    assert L.clite_stem_wgrad_patch(ptr(dyb), ptr(xpad), BF16, N, Hp, Wp, Ho, Wo2, ptr(dw), ptr(ws), nb.value, None) == 1



@pytest.mark.policy_independent
def test_grouped_weight_gradients():
    """clite_wgrad_group: conv weight gradients of all three tile families (<= 64 output channels, <= 64 (r, s, ci) columns, general) and a
    linear weight gradient with a strided operand, as ONE grouped launch set, accumulate (+=) the same values as the per-member entry points
    compute — including a member whose K range is cut into several k-chunks."""
    from simlib import Conv

    class Item(C.Structure):
        _fields_ = [("kind", C.c_int32), ("a", C.c_void_p), ("b", C.c_void_p), ("out", C.c_void_p), ("cv", Conv),
                    ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("lda", C.c_int32), ("ldb", C.c_int32), ("ldc", C.c_int32),
                    ("a_scales", C.c_void_p), ("b_scales", C.c_void_p), ("row_scale", C.c_void_p)]

    rng = np.random.default_rng(5)
    L = lib()
    L.clite_wgrad_group.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    L.clite_wgrad_group_workspace.argtypes = [C.c_int, C.c_int64, C.c_void_p]
    items, refs, outs, hold = [], [], [], []
    # 4th: 1024 pixels = 32 K tiles -> two k-chunks (the simulator build's chunk is 24 K tiles, Makefile: CLITE_GROUP_KCHUNK; the product's 256). Then the
    # 8-wave wide bucket: 288 x 288 outputs (256 x 256 tiles, ragged both ways, a ragged last K tile of 64), and two 128-wide outputs that stay on
    # the 4-wave tiles
    for (N, H, W, Cc, K, R, st, pad) in [(2, 8, 8, 32, 64, 3, 1, 1), (3, 6, 6, 64, 136, 1, 1, 0), (2, 9, 7, 128, 160, 1, 2, 0), (1, 32, 32, 32, 64, 1, 1, 0),
                                         (2, 7, 7, 32, 288, 3, 1, 1), (1, 10, 10, 128, 256, 1, 1, 0), (1, 9, 9, 256, 128, 1, 1, 0)]:
        Ho, Wo = (H + 2 * pad - R) // st + 1, (W + 2 * pad - R) // st + 1
        cv = Conv(BF16, N, H, W, Cc, K, R, R, st, pad, Ho, Wo)
        x, xb = _prep(rng.standard_normal((N, H, W, Cc), dtype=np.float32), BF16)
        dy, dyb = _prep(rng.standard_normal((N, Ho, Wo, K), dtype=np.float32), BF16)
        dw = np.ones((K, R, R, Cc), np.float32)
        it = Item(); it.kind, it.a, it.b, it.out, it.cv = 0, ptr(dyb).value, ptr(xb).value, ptr(dw).value, cv
        items.append(it); outs.append(dw); refs.append(1 + conv_wgrad_ref(dy, x, dw.shape, st, pad)); hold += [xb, dyb]
    Kt, M, N = 300, 72, 136                                   # linear member, B rows strided (ldb > N)
    A, Ab = _prep(rng.standard_normal((Kt, M), dtype=np.float32), BF16)
    B, Bb = _prep(rng.standard_normal((Kt, 2 * N), dtype=np.float32), BF16)
    out = np.ones((M, N), np.float32)
    it = Item(); it.kind, it.a, it.b, it.out = 1, ptr(Ab).value, ptr(Bb).value, ptr(out).value
    it.M, it.N, it.K, it.lda, it.ldb, it.ldc = M, N, Kt, M, 2 * N, N
    items.append(it); outs.append(out); refs.append(1 + A.T @ B[:, :N]); hold += [Ab, Bb]
    Kt, M, N = 100, 256, 264                                  # linear member on the 256 x 256 wide tile
    A2, A2b = _prep(rng.standard_normal((Kt, M), dtype=np.float32), BF16)
    B2, B2b = _prep(rng.standard_normal((Kt, N), dtype=np.float32), BF16)
    out2 = np.ones((M, N), np.float32)
    it = Item(); it.kind, it.a, it.b, it.out = 1, ptr(A2b).value, ptr(B2b).value, ptr(out2).value
    it.M, it.N, it.K, it.lda, it.ldb, it.ldc = M, N, Kt, M, N, N
    items.append(it); outs.append(out2); refs.append(1 + A2.T @ B2); hold += [A2b, B2b]
    nb = C.c_uint64(0)
    assert L.clite_wgrad_group_workspace(len(items), 4096, C.byref(nb)) == 0
    ws_dev, ws_host = np.zeros(nb.value, np.uint8), np.zeros(nb.value, np.uint8)
    arr = (Item * len(items))(*items)
    assert L.clite_wgrad_group(BF16, arr, len(items), ptr(ws_dev), ptr(ws_host), nb.value, None) == 0
    for got, ref in zip(outs, refs):
        _close(got, ref)
    assert L.clite_wgrad_group(BF16, arr, len(items), ptr(ws_dev), ptr(ws_host), 64, None) == -2          # workspace too small: refused, nothing launched
    # CLITE_WGRAD_ZEROED (ABI v10): the caller vouches for zeroed outputs; single-chunk members then store instead of adding atomically, the
    # k-chunked member (the 4th) keeps its atomics. Same values.
    for it, o in zip(items, outs):
        it.kind |= 0x200
        o[...] = 0.0
    arr = (Item * len(items))(*items)
    assert L.clite_wgrad_group(BF16, arr, len(items), ptr(ws_dev), ptr(ws_host), nb.value, None) == 0
    for got, ref in zip(outs, refs):
        _close(got, ref - 1)
    # ABI v12: a per-output-row factor on a member's product (clite_wgrad_item.row_scale: the folded BatchNorm backward's ka) — such a member always adds
    # (its correction terms arrive from other launches), also with CLITE_WGRAD_ZEROED — and CLITE_WGRAD_SHORTK (quarter-length K chunks) on the
    # k-chunked member: same values
    scales = []
    for i, (it, o) in enumerate(zip(items, outs)):
        o[...] = 0.5
        sc = (1 + 0.25 * rng.standard_normal(o.shape[0])).astype(np.float32)
        scales.append(sc)
        it.row_scale = ptr(sc).value
        if i == 3:
            it.kind |= 0x400
    arr = (Item * len(items))(*items)
    assert L.clite_wgrad_group(BF16, arr, len(items), ptr(ws_dev), ptr(ws_host), nb.value, None) == 0
    for got, ref, sc in zip(outs, refs, scales):
        _close(got, 0.5 + (ref - 1) * sc.reshape((-1,) + (1,) * (ref.ndim - 1)))
    assert L.clite_wgrad_group(F32, arr, len(items), None, None, 0, None) == -1          # members one by one have no row_scale form


@pytest.mark.parametrize("M,K,Cin", [(200, 64, 64), (700, 128, 32), (300, 192, 128)])
def test_folded_batchnorm_backward(M, K, Cin):
    """ABI v12: the BatchNorm backward of a 1 x 1 conv -> BatchNorm unit folded into the convolution's two gradients. clite_bn_fold_prepare (row-scaled
    weights, constant row, coefficients, dgamma / dbeta), clite_conv_dgrad_bnfold (one GEMM over the K-concatenation [dz | y] through the BatchNorm-backward
    epilogue with a bias: every tile family; the 700-row case runs the row-range persistent form of the simulator build), and the weight-gradient side —
    clite_bn_apply's column sums of its output (clite_bn.out_sum), the group's row_scale member and Gram matrix, clite_bn_fold_wgrad_finish — against a
    numpy evaluation of BatchNorm backward followed by the convolution's backward."""
    from simlib import Bn
    rng = np.random.default_rng(M + K)
    L = lib()
    a, ab = _prep(np.maximum(rng.standard_normal((M, Cin), dtype=np.float32) + 0.3, 0), BF16)
    W, Wb = _prep(rng.standard_normal((K, Cin), dtype=np.float32) * (2.0 / Cin) ** 0.5, BF16)
    y, yb = _prep(a @ W.T + 0.5, BF16)
    dz, _ = _prep((rng.standard_normal((M, K), dtype=np.float32) * 0.1 + 0.02 * rng.standard_normal(K).astype(np.float32)) * (rng.random((M, K)) > 0.4), BF16)
    dzb = to_bf16(dz)
    gamma = (rng.random(K) + 0.5).astype(np.float32)
    R = 2
    st = np.zeros((R, 3, K), np.float32)
    st[0, 0], st[1, 0] = 0.25 * y.sum(0), 0.75 * y.sum(0)
    st[0, 1], st[1, 1] = 0.5 * (y * y).sum(0), 0.5 * (y * y).sum(0)
    mean = y.astype(np.float64).mean(0)
    var = np.maximum((y.astype(np.float64) ** 2).mean(0) - mean * mean, 0)
    rstd = 1.0 / np.sqrt(var + 1e-5)
    S1, S2 = dz.astype(np.float64).sum(0), (dz * (y - mean)).astype(np.float64).sum(0)
    pre = np.zeros((R, 3, K), np.float32)
    pre[0, 0], pre[1, 0] = 0.5 * S1, 0.5 * S1
    pre[0, 1], pre[1, 1] = 0.3 * S2, 0.7 * S2
    xhat = (y - mean) * rstd
    dy = gamma * rstd * (dz - S1 / M - xhat * (dz * xhat).sum(0) / M)
    da_ref, dW_ref = dy @ W, dy.T @ a
    # the unit in front: its BatchNorm input, forward sums, relu' bits
    y2, y2b = _prep(rng.standard_normal((M, Cin), dtype=np.float32) + 1.0, BF16)
    st2 = np.zeros((R, 3, Cin), np.float32)
    st2[0, 0], st2[1, 0] = 0.5 * y2.sum(0), 0.5 * y2.sum(0)
    mean2 = y2.mean(0)
    mask = rng.random((M, Cin)) > 0.3
    bits = pack_relu_bits(mask.astype(np.float32))
    pair = np.concatenate([dzb.reshape(1, M, K), yb.reshape(1, M, K)], axis=0).copy()
    Wt = np.ascontiguousarray(Wb.T)
    w2, bias, coef = np.zeros((Cin, 2, K), np.uint16), np.zeros(Cin, np.float32), np.zeros((3, K), np.float32)
    dgamma, dbeta = np.zeros(K, np.float32), np.zeros(K, np.float32)
    rm, rv = np.zeros(K, np.float32), np.ones(K, np.float32)
    p = Bn(M, K, ptr(st), ptr(gamma), ptr(np.zeros(K, np.float32)), ptr(rm), ptr(rv), 1, 0, 0.1, 1e-5, 0, R, 3 * K, 0)
    L.clite_bn_fold_prepare.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 6
    assert L.clite_bn_fold_prepare(C.byref(p), ptr(pre), ptr(Wt), Cin, ptr(w2), ptr(bias), ptr(coef), ptr(dgamma), ptr(dbeta), None) == 0
    _close(coef[0], gamma * rstd, 1e-4)
    _close(dgamma, (dz * xhat).sum(0), 2e-3)
    _close(dbeta, S1, 1e-4)
    dz2 = np.zeros((M, Cin), np.uint16)
    d2 = np.zeros((R, 3, Cin), np.float32)
    ep = make_ep(dz2, Cin, bias=bias, colsum=d2, relu_bits=bits)
    ep.colsum_replicas, ep.colsum_stride = R, 3 * Cin
    ep.bn_y, ep.bn_stats, ep.bn_replicas, ep.bn_rstride, ep.bn_inv_count = ptr(y2b), ptr(st2), R, 3 * Cin, 1.0 / M
    L.clite_conv_dgrad_bnfold.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    assert L.clite_conv_dgrad_bnfold(ptr(pair), ptr(w2), M, K, Cin, C.byref(ep), None) == 0
    got = from_bf16(dz2)
    _close(got, da_ref * mask, 6e-3)
    d = d2.sum(0)
    _close(d[0], got.sum(0), 5e-3)
    _close(d[1], (got * (y2 - mean2)).sum(0), 5e-3)
    ep.bias = None                                           # the folded form needs its constant row
    assert L.clite_conv_dgrad_bnfold(ptr(pair), ptr(w2), M, K, Cin, C.byref(ep), None) == -1
    # weight gradient. colsum(a) from an identity BatchNorm + ReLU over a (clite_bn.out_sum: the pass that writes a also sums it)
    Ra = 4
    asum = np.zeros((Ra, 3, Cin), np.float32)
    ist = np.zeros((1, 3, Cin), np.float32)
    ist[0, 1] = M * (1.0 - 1e-5)
    one, zero = np.ones(Cin, np.float32), np.zeros(Cin, np.float32)
    pa = Bn(M, Cin, ptr(ist), ptr(one), ptr(zero), ptr(zero.copy()), ptr(one.copy()), 1, 0, 0.1, 1e-5, 1, 1, 3 * Cin, 0)
    pa.out_sum, pa.out_sum_replicas, pa.out_sum_stride = ptr(asum), Ra, 3 * Cin
    a_out = np.zeros((M, Cin), np.uint16)
    assert L.clite_bn_apply(C.byref(pa), BF16, ptr(ab), None, ptr(a_out), None) == 0
    assert np.array_equal(a_out, ab)
    _close(asum[:, 0].sum(0), a.sum(0), 1e-4)

    class Item(C.Structure):
        _fields_ = [("kind", C.c_int32), ("a", C.c_void_p), ("b", C.c_void_p), ("out", C.c_void_p), ("cv", Conv),
                    ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("lda", C.c_int32), ("ldb", C.c_int32), ("ldc", C.c_int32),
                    ("a_scales", C.c_void_p), ("b_scales", C.c_void_p), ("row_scale", C.c_void_p)]
    dw, G = np.zeros((K, Cin), np.float32), np.zeros((Cin, Cin), np.float32)
    ka = np.ascontiguousarray(coef[0])
    it0 = Item(); it0.kind, it0.a, it0.b, it0.out, it0.cv = 0x200, ptr(dzb).value, ptr(ab).value, ptr(dw).value, Conv(BF16, 1, 1, M, Cin, K, 1, 1, 1, 0, 1, M)
    it0.row_scale = ptr(ka).value
    it1 = Item(); it1.kind, it1.a, it1.b, it1.out, it1.cv = 0x200 | 0x400, ptr(ab).value, ptr(ab).value, ptr(G).value, Conv(BF16, 1, 1, M, Cin, Cin, 1, 1, 1, 0, 1, M)
    L.clite_wgrad_group.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    L.clite_wgrad_group_workspace.argtypes = [C.c_int, C.c_int64, C.c_void_p]
    nb = C.c_uint64(0)
    assert L.clite_wgrad_group_workspace(2, 4096, C.byref(nb)) == 0
    ws_dev, ws_host = np.zeros(nb.value, np.uint8), np.zeros(nb.value, np.uint8)
    arr = (Item * 2)(it0, it1)
    assert L.clite_wgrad_group(BF16, arr, 2, ptr(ws_dev), ptr(ws_host), nb.value, None) == 0
    _close(G, a.T @ a, 1e-3)
    L.clite_bn_fold_wgrad_finish.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    assert L.clite_bn_fold_wgrad_finish(ptr(G), ptr(asum), Ra, 3 * Cin, ptr(coef), ptr(Wt), M, K, Cin, ptr(dw), None) == 0
    _close(dw, dW_ref, 3e-3)


@pytest.mark.policy_independent
@pytest.mark.parametrize("M", [96, 250, 1000])
def test_folded_dgrad_streaming_kernel_k256_c64(M):
    """fold_dgrad.hip: clite_conv_dgrad_bnfold for K = 256, Cin = 64 (ResNet-50 layer1's conv3) under the automatic tile policy — whole-row K tiles by
    LDS-DMA into XOR-swizzled [row][512 B] images, the weights resident in LDS, a two-tile ring, the two K parts of a tile multiplied by different waves
    and joined in LDS, the row epilogue with operands requested a tile ahead. Ragged pixel counts (the last tile partial; 1000 pixels = 32 tiles over the
    simulator build's 3 persistent workgroups); against numpy, and bit-identical in the stored output to the tile engine's launch of the same problem."""
    from simlib import Bn
    L = lib()
    assert L.clite_set_tile_policy(0) == 0
    K, Cin, R = 256, 64, 2
    rng = np.random.default_rng(M)
    dz, _ = _prep(rng.standard_normal((M, K), dtype=np.float32) * 0.1 * (rng.random((M, K)) > 0.4), BF16)
    y, _ = _prep(rng.standard_normal((M, K), dtype=np.float32) + 0.5, BF16)
    pair = np.concatenate([to_bf16(dz).reshape(1, M, K), to_bf16(y).reshape(1, M, K)], axis=0).copy()
    w2f, w2 = _prep(rng.standard_normal((Cin, 2, K), dtype=np.float32) * 0.05, BF16)
    bias = rng.standard_normal(Cin).astype(np.float32) * 0.1
    y2, y2b = _prep(rng.standard_normal((M, Cin), dtype=np.float32) + 1.0, BF16)
    st2 = np.zeros((R, 3, Cin), np.float32)
    st2[0, 0], st2[1, 0] = 0.5 * y2.sum(0), 0.5 * y2.sum(0)
    mean2 = y2.mean(0)
    mask = rng.random((M, Cin)) > 0.3
    bits = pack_relu_bits(mask.astype(np.float32))
    L.clite_conv_dgrad_bnfold.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]

    def run(policy):
        assert L.clite_set_tile_policy(policy) == 0
        out, d2 = np.zeros((M, Cin), np.uint16), np.zeros((R, 3, Cin), np.float32)
        ep = make_ep(out, Cin, bias=bias, colsum=d2, relu_bits=bits)
        ep.colsum_replicas, ep.colsum_stride = R, 3 * Cin
        ep.bn_y, ep.bn_stats, ep.bn_replicas, ep.bn_rstride, ep.bn_inv_count = ptr(y2b), ptr(st2), R, 3 * Cin, 1.0 / M
        assert L.clite_conv_dgrad_bnfold(ptr(pair), ptr(w2), M, K, Cin, C.byref(ep), None) == 0
        return from_bf16(out), d2.sum(0)
    got, d = run(0)                    # the streaming kernel
    ref = (y @ w2f[:, 0, :].T + dz @ w2f[:, 1, :].T + bias) * mask
    _close(got, ref, 6e-3)
    _close(d[0], got.sum(0), 5e-3)
    _close(d[1], (got * (y2 - mean2)).sum(0), 5e-3)
    got4, d4 = run(4)                  # the 4-wave tile engine on the same problem
    L.clite_set_tile_policy(0)
    _close(got, got4, 8e-3)            # (different summation order over k: within one bf16 rounding of each other)
    _close(d, d4, 5e-3)


class TransposeItem(C.Structure):
    _fields_ = [("src_off", C.c_uint64), ("dst_off", C.c_uint64)] + [(n, C.c_uint32) for n in
                ("rows", "cols", "src_ld", "dst_ld", "batch", "src_bstride", "dst_bstride", "first_tile")]


def test_transpose_weights_grouped_launch():
    """clite_transpose_weights: three items in one launch inside one flat buffer — a Linear [N][K] -> [K][N] with ragged 64-tiles, a conv
    [K][R][S][C] -> [C][R][S][K] (R*S batched [K][C] -> [C][K] with the two row strides), and three adjacent q/k/v-style weights as one
    [3N][K] matrix — against numpy transposes; untouched gaps of the destination stay as they were."""
    rng = np.random.default_rng(4)
    N1, K1 = 72, 136
    Kc, R, S, Cc = 40, 3, 3, 24
    N3, K3 = 24, 64
    o1, o2, o3 = 0, 10240, 20480
    total = 30720
    src = rng.integers(0, 65535, total, dtype=np.uint16)
    dst = np.full(total, 0x1234, np.uint16)
    items, tile = [], 0
    for (o, rows, cols, sld, dld, batch, sb, db) in [(o1, N1, K1, K1, N1, 1, 0, 0), (o2, Kc, Cc, R * S * Cc, R * S * Kc, R * S, Cc, Kc), (o3, 3 * N3, K3, K3, 3 * N3, 1, 0, 0)]:
        items.append(TransposeItem(o, o, rows, cols, sld, dld, batch, sb, db, tile))
        tile += batch * ((rows + 63) // 64) * ((cols + 63) // 64)
    arr = (TransposeItem * 3)(*items)
    assert lib().clite_transpose_weights(ptr(src), ptr(dst), C.byref(arr), 3, C.c_uint32(tile), None) == 0
    assert np.array_equal(dst[o1:o1 + N1 * K1].reshape(K1, N1), src[o1:o1 + N1 * K1].reshape(N1, K1).T)
    assert np.array_equal(dst[o2:o2 + Kc * R * S * Cc].reshape(Cc, R, S, Kc), src[o2:o2 + Kc * R * S * Cc].reshape(Kc, R, S, Cc).transpose(3, 1, 2, 0))
    assert np.array_equal(dst[o3:o3 + 3 * N3 * K3].reshape(K3, 3 * N3), src[o3:o3 + 3 * N3 * K3].reshape(3 * N3, K3).T)
    assert (dst[o1 + N1 * K1:o2] == 0x1234).all() and (dst[o3 + 3 * N3 * K3:] == 0x1234).all()


@pytest.mark.parametrize("dtype,N,H,W,Cc,K,R,S,st,pad", [(BF16, 2, 8, 8, 32, 64, 3, 3, 1, 1), (F32, 2, 9, 7, 64, 32, 3, 3, 2, 1), (BF16, 3, 6, 6, 136, 64, 1, 1, 1, 0),
                                                         (BF16, 2, 8, 8, 32, 64, 1, 1, 2, 0), (BF16, 1, 8, 8, 64, 64, 3, 3, 2, 1)])
def test_conv_dgrad_on_transposed_weights(dtype, N, H, W, Cc, K, R, S, st, pad):
    """clite_conv_dgrad_wt / clite_conv_dgrad_s2class_wt (weight given as [C][R][S][K]: a k-contiguous operand like the forward's) against
    the reference dgrad; with the BatchNorm-backward epilogue too (the form the ResNet backward launches)."""
    rng = np.random.default_rng(H + K)
    Ho, Wo = (H + 2 * pad - R) // st + 1, (W + 2 * pad - S) // st + 1
    cv = Conv(dtype, N, H, W, Cc, K, R, S, st, pad, Ho, Wo)
    M = N * H * W
    w, _ = _prep(rng.standard_normal((K, R, S, Cc), dtype=np.float32) * 0.2, dtype)
    wt, wtb = _prep(np.ascontiguousarray(w.transpose(3, 1, 2, 0)), dtype)
    dy, dyb = _prep(rng.standard_normal((N, Ho, Wo, K), dtype=np.float32), dtype)
    g = conv_dgrad_ref(dy, w, (N, H, W, Cc), st, pad).reshape(M, Cc)
    out = np.zeros((M, Cc), np.float32)
    ep = make_ep(out, Cc, out_f32=True)
    assert lib().clite_conv_dgrad_wt(ptr(dyb), ptr(wtb), C.byref(cv), C.byref(ep), None) == 0
    _close(out, g)
    if R == 1 and st == 2:       # the strided 1x1 shortcut accumulated in place: dense GEMM over the output pixels, rows scattered
        _, acc = _prep(np.ones((M, Cc), np.float32), dtype)          # in the compute dtype: the residual operand is read as such
        acc = acc.copy()
        ep = make_ep(acc, Cc, out_f32=False, residual=acc)
        assert lib().clite_conv_dgrad_wt(ptr(dyb), ptr(wtb), C.byref(cv), C.byref(ep), None) == 0
        _close(from_bf16(acc) if dtype == BF16 else acc, 1 + g, 8e-3)
    if R == 3 and st == 2 and H % 2 == 0 and Cc % 64 == 0 and K % 64 == 0:
        out2 = np.full((M, Cc), 7.0, np.float32)
        for ph in (0, 1):
            for pw in (0, 1):
                sub, subb = _prep(np.ascontiguousarray(wt[:, (ph + 1) & 1::2, (pw + 1) & 1::2, :]), dtype)
                ep = make_ep(out2, Cc, out_f32=True)
                assert lib().clite_conv_dgrad_s2class_wt(ptr(dyb), ptr(subb), C.byref(cv), ph, pw, C.byref(ep), None) == 0
        _close(out2, g)
    if st == 1:
        aux = rng.standard_normal((M, Cc)).astype(np.float32)
        res, resb = _prep(rng.standard_normal((M, Cc), dtype=np.float32), dtype)
        y, yb = _prep(rng.standard_normal((M, Cc), dtype=np.float32) + 3.0, dtype)
        fstats = np.zeros((4, 3, Cc), np.float32)
        fstats[:, 0] = y.sum(0) / 4
        out3 = np.zeros((M, Cc), np.float32)
        dst = np.zeros((4, 3, Cc), np.float32)
        bits = pack_relu_bits(aux)          # (kept in a variable: the epilogue holds a raw pointer into it)
        ep = make_ep(out3, Cc, out_f32=True, residual=resb, colsum=dst, relu_bits=bits)
        ep.colsum_replicas, ep.colsum_stride = 4, 3 * Cc
        ep.bn_y, ep.bn_stats, ep.bn_replicas, ep.bn_rstride, ep.bn_inv_count, ep.mask_after_residual = ptr(yb), ptr(fstats), 4, 3 * Cc, 1.0 / M, 1
        assert lib().clite_conv_dgrad_wt(ptr(dyb), ptr(wtb), C.byref(cv), C.byref(ep), None) == 0
        v = (g + res) * (aux > 0)
        _close(out3, v)
        _close(dst.sum(0)[1], (v * (y - fstats[:, 0].sum(0) / M)).sum(0), 5e-3)
        # the output in the compute dtype: in bf16 these are the two compile-time specialised forms of the epilogue that the ResNet backward
        # launches (igemm_epilogue_bn FORM 2: residual + mask after it; FORM 1: mask only) — same values, rounded once
        for with_res in (True, False):
            out4 = outbuf_like(dtype, (M, Cc))
            dst4 = np.zeros((4, 3, Cc), np.float32)
            ep = make_ep(out4, Cc, out_f32=False, residual=resb if with_res else None, colsum=dst4, relu_bits=bits)
            ep.colsum_replicas, ep.colsum_stride = 4, 3 * Cc
            ep.bn_y, ep.bn_stats, ep.bn_replicas, ep.bn_rstride, ep.bn_inv_count, ep.mask_after_residual = ptr(yb), ptr(fstats), 4, 3 * Cc, 1.0 / M, int(with_res)
            assert lib().clite_conv_dgrad_wt(ptr(dyb), ptr(wtb), C.byref(cv), C.byref(ep), None) == 0
            got = from_bf16(out4) if dtype == BF16 else out4
            want = (g + res) * (aux > 0) if with_res else g * (aux > 0)
            _close(got, want, 8e-3 if dtype == BF16 else 2e-3)
            _close(dst4.sum(0)[0], got.sum(0), 1e-3)
            _close(dst4.sum(0)[1], (got * (y - fstats[:, 0].sum(0) / M)).sum(0), 5e-3)


@pytest.mark.parametrize("dtype", [BF16, F32])
def test_colsum_rows_1_accumulates_the_column_sums_only(dtype):
    """clite_epilogue.colsum_rows = 1: only row 0 (the column sums of the stored values) is accumulated — straight into a bias gradient whose
    neighbour in memory is another tensor (BERT FFN: the gelu' input-gradient GEMM also produces intermediate.dense.bias.grad). The floats behind
    the N sums must stay untouched, and the sums ADD to what is there."""
    rng = np.random.default_rng(11)
    M, N, K = 200, 136, 104
    A, Ab = _prep(rng.standard_normal((M, K), dtype=np.float32), dtype)
    B, Bb = _prep(rng.standard_normal((K, N), dtype=np.float32) * 0.2, dtype)
    pre, preb = _prep(rng.standard_normal((M, N), dtype=np.float32), dtype)
    out = np.zeros((M, N), np.uint16 if dtype == BF16 else np.float32)
    sums = np.full(2 * N, 5.0, np.float32)
    ep = make_ep(out, N, dact_aux=preb, dact=2, colsum=sums)
    ep.colsum_rows = 1
    assert lib().clite_gemm_nn(ptr(Ab), K, ptr(Bb), N, M, N, K, dtype, C.byref(ep), None) == 0
    got = from_bf16(out) if dtype == BF16 else out
    from math import erf, exp, pi, sqrt
    gp = np.vectorize(lambda x: 0.5 * (1 + erf(x / sqrt(2))) + x * exp(-0.5 * x * x) / sqrt(2 * pi))(pre).astype(np.float32)
    _close(got, (A @ B) * gp, 8e-3 if dtype == BF16 else 2e-3)
    _close(sums[:N], 5.0 + got.sum(0), 1e-4)
    assert (sums[N:] == 5.0).all()



@pytest.mark.parametrize("M,N,K", [(200, 136, 104), (300, 264, 128)])
def test_bert_epilogue_forms_compiled_in(M, N, K, tile_policy):
    """The four compile-time epilogue forms of BERT's linears on the 8-wave kernels (igemm_wide.h WideEpiForm 3 - 6; gemm_wide.hip bert_form
    recognises them from the run-time description): FFN1 forward (bias + pre-activation store + GELU), FFN2's input gradient (acc x GELU'(aux)
    with the bias gradient's column sums, colsum_rows = 1), the output projections (bias + dropout + residual) and the residual-stream input
    gradients (+ residual). Against numpy, ragged edges included; the dropout form against the same launch on the 4-wave kernel's run-time-flag
    epilogue (policy 4): the mask is a function of (seed, site, element index) only, so the SAME elements must be zero."""
    from scipy.special import erf
    rng = np.random.default_rng(M + N + K)
    A, Ab = _prep(rng.standard_normal((M, K), dtype=np.float32) * 0.3, BF16)
    B, Bb = _prep(rng.standard_normal((N, K), dtype=np.float32) * 0.3, BF16)
    R, Rb = _prep(rng.standard_normal((M, N), dtype=np.float32), BF16)
    aux, auxb = _prep(rng.standard_normal((M, N), dtype=np.float32) * 1.5, BF16)
    bias = rng.standard_normal(N).astype(np.float32)
    z = A @ B.T
    cdf = lambda x: 0.5 * (1 + erf(x / np.sqrt(2)))
    # form 3
    out, pre = np.zeros((M, N), np.uint16), np.zeros((M, N), np.uint16)
    ep = make_ep(out, N, bias=bias, act=2, preact=pre)
    assert lib().clite_gemm_nt(ptr(Ab), K, ptr(Bb), K, M, N, K, BF16, C.byref(ep), None) == 0
    _close(from_bf16(pre), z + bias, 6e-3)
    _close(from_bf16(out), (z + bias) * cdf(z + bias), 6e-3)
    # form 4, with and without the column sums; the floats behind the sums (what colsum_rows = 1 must not touch) stay as they were
    for with_sums in (True, False):
        out = np.zeros((M, N), np.uint16)
        colsum = np.full((2, N), 7.0, np.float32)
        colsum[0] = 0.0
        ep = make_ep(out, N, dact_aux=auxb, dact=2, colsum=colsum if with_sums else None)
        ep.colsum_rows = 1
        assert lib().clite_gemm_nt(ptr(Ab), K, ptr(Bb), K, M, N, K, BF16, C.byref(ep), None) == 0
        gp = cdf(aux) + aux * np.exp(-0.5 * aux * aux) / np.sqrt(2 * np.pi)
        _close(from_bf16(out), z * gp, 6e-3)
        if with_sums:
            _close(colsum[0], from_bf16(out).sum(0), 1e-4)
            assert (colsum[1] == 7.0).all()
    # form 6
    out = np.zeros((M, N), np.uint16)
    ep = make_ep(out, N, residual=Rb)
    assert lib().clite_gemm_nt(ptr(Ab), K, ptr(Bb), K, M, N, K, BF16, C.byref(ep), None) == 0
    _close(from_bf16(out), z + R, 6e-3)
    # form 5 against the run-time-flag epilogue of the 4-wave kernel
    res = []
    for pol in (tile_policy, 4):
        assert lib().clite_set_tile_policy(pol) == 0
        out = np.zeros((M, N), np.uint16)
        ep = make_ep(out, N, bias=bias, drop_p=0.25, drop_seed=1234, drop_site=5, residual=Rb)
        assert lib().clite_gemm_nt(ptr(Ab), K, ptr(Bb), K, M, N, K, BF16, C.byref(ep), None) == 0
        res.append(from_bf16(out))
    assert lib().clite_set_tile_policy(tile_policy) == 0
    dropped = np.isclose(res[1], R, atol=0)                # a dropped element is the bare residual
    assert 0.15 < dropped.mean() < 0.35
    assert (np.isclose(res[0], R, atol=0) == dropped).all()
    _close(res[0], res[1], 6e-3)
    keep = ~dropped
    _close(res[0][keep], ((z + bias) / 0.75 + R)[keep], 6e-3)


@pytest.mark.policy_independent
@pytest.mark.parametrize("N,H,W", [(3, 10, 12), (2, 9, 40), (1, 8, 56)])
def test_patch_resident_conv3x3_64ch(N, H, W):
    """conv_patch.hip: the patch-resident 3 x 3 / stride 1 kernel for 64 -> 64 channels (taken under the automatic tile policy only) — forward with
    the per-channel statistics, the plain input gradient on transposed weights, and the BatchNorm-backward form (packed relu' bits + the two
    reductions) — against the numpy references, and bit-for-bit in its stored values' statistics. The simulator build runs 2 persistent
    workgroups, so these cases walk several strips per workgroup (double-buffered patches), ragged last strips (H = 9 with 6-row strips) and
    MFMA row blocks that cross strip rows (W = 12, 40, 56)."""
    assert lib().clite_set_tile_policy(0) == 0
    rng = np.random.default_rng(H * W)
    Cc = K = 64
    cv = Conv(BF16, N, H, W, Cc, K, 3, 3, 1, 1, H, W)
    M = N * H * W
    x, xb = _prep(rng.standard_normal((N, H, W, Cc), dtype=np.float32), BF16)
    w, wb = _prep(rng.standard_normal((K, 3, 3, Cc), dtype=np.float32) * 0.1, BF16)
    dy, dyb = _prep(rng.standard_normal((N, H, W, K), dtype=np.float32), BF16)
    # forward
    y = np.zeros((M, K), np.uint16)
    csr = np.zeros((4, 3, K), np.float32)
    ep = make_ep(y, K, colsum=csr)
    ep.colsum_replicas, ep.colsum_stride = 4, 3 * K
    assert lib().clite_conv_fwd(ptr(xb), ptr(wb), C.byref(cv), C.byref(ep), None) == 0
    ref = conv_ref(x, w, 1, 1).reshape(M, K)
    got = from_bf16(y)
    _close(got, ref, 6e-3)
    cs = csr.sum(0)
    _close(cs[0], got.sum(0), 1e-4)               # statistics of the STORED (bf16-rounded) values
    _close(cs[1], (got ** 2).sum(0), 1e-4)
    assert not csr[:, 2].any()
    # the 4-wave implicit-GEMM path on the same launch (policy 4) agrees
    assert lib().clite_set_tile_policy(4) == 0
    y4 = np.zeros((M, K), np.uint16)
    ep4 = make_ep(y4, K)
    assert lib().clite_conv_fwd(ptr(xb), ptr(wb), C.byref(cv), C.byref(ep4), None) == 0
    assert lib().clite_set_tile_policy(0) == 0
    _close(got, from_bf16(y4), 8e-3)
    # input gradient on the transposed weights, plain store
    wt, wtb = _prep(np.ascontiguousarray(w.transpose(3, 1, 2, 0)), BF16)
    g = conv_dgrad_ref(dy, w, (N, H, W, Cc), 1, 1).reshape(M, Cc)
    dx = np.zeros((M, Cc), np.uint16)
    assert lib().clite_conv_dgrad_wt(ptr(dyb), ptr(wtb), C.byref(cv), C.byref(make_ep(dx, Cc)), None) == 0
    _close(from_bf16(dx), g, 6e-3)
    # BatchNorm-backward form (igemm.h FORM 1): v = acc where the bit is set, sum v and sum v (bn_y - mean)
    aux = rng.standard_normal((M, Cc)).astype(np.float32)
    yb_v, ybb = _prep(rng.standard_normal((M, Cc), dtype=np.float32) + 3.0, BF16)
    fstats = np.zeros((2, 3, Cc), np.float32)
    fstats[:, 0] = yb_v.sum(0) / 2
    bits = pack_relu_bits(aux)
    out = np.zeros((M, Cc), np.uint16)
    dst = np.zeros((2, 3, Cc), np.float32)
    ep = make_ep(out, Cc, colsum=dst, relu_bits=bits)
    ep.colsum_replicas, ep.colsum_stride = 2, 3 * Cc
    ep.bn_y, ep.bn_stats, ep.bn_replicas, ep.bn_rstride, ep.bn_inv_count = ptr(ybb), ptr(fstats), 2, 3 * Cc, 1.0 / M
    assert lib().clite_conv_dgrad_wt(ptr(dyb), ptr(wtb), C.byref(cv), C.byref(ep), None) == 0
    gv = from_bf16(out)
    _close(gv, g * (aux > 0), 6e-3)
    assert ((gv == 0) | (aux > 0)).all()
    d = dst.sum(0)
    _close(d[0], gv.sum(0), 1e-4)
    _close(d[1], (gv * (yb_v - fstats[:, 0].sum(0) / M)).sum(0), 2e-3)
    assert not d[2].any()


@pytest.mark.policy_independent
@pytest.mark.parametrize("N,H,W", [(3, 10, 12)])          # 3 strips on 2 workgroups (double buffer + a lone strip); the full-width case (W = 56) runs on the GPU (test_gpu_igemm.py)
def test_patch_resident_wgrad3x3_64ch(N, H, W):
    """conv_patch.hip, weight gradient (clite_conv_wgrad_patch, ABI v10): dW [64][3][3][64] f32 += dy^T * im2col(x) with the whole output held by
    each persistent workgroup and both operands read from per-strip LDS images by transposed reads; per-workgroup slabs + the reduction kernel.
    Against the numpy reference and against clite_conv_wgrad on the same operands; accumulation into a non-zero dW; the workspace-too-small and
    wrong-shape refusals (return 1, nothing launched)."""
    assert lib().clite_set_tile_policy(0) == 0
    rng = np.random.default_rng(H + W)
    Cc = K = 64
    cv = Conv(BF16, N, H, W, Cc, K, 3, 3, 1, 1, H, W)
    x, xb = _prep(rng.standard_normal((N, H, W, Cc), dtype=np.float32), BF16)
    dy, dyb = _prep(rng.standard_normal((N, H, W, K), dtype=np.float32), BF16)
    nb = C.c_uint64(0)
    assert lib().clite_conv_wgrad_patch_workspace(C.byref(nb)) == 0 and nb.value == max(2 * 9 * 64 * 64 * 4, 6 * 64 * 224 * 4)          # the simulator build runs 2 (the stem's kernel 6) workgroups
    ws = np.full(nb.value // 4, np.nan, np.float32)          # scratch arrives dirty
    dw = np.full((K, 3, 3, Cc), 0.5, np.float32)
    lib().clite_conv_wgrad_patch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    assert lib().clite_conv_wgrad_patch(ptr(dyb), ptr(xb), C.byref(cv), ptr(dw), ptr(ws), nb.value, None) == 0
    ref = conv_wgrad_ref(dy, x, (K, 3, 3, Cc), 1, 1)
    _close(dw - 0.5, ref, 2e-3)
    dw2 = np.zeros((K, 3, 3, Cc), np.float32)
    assert lib().clite_conv_wgrad(ptr(dyb), ptr(xb), C.byref(cv), ptr(dw2), None) == 0
    _close(dw - 0.5, dw2, 2e-3)
    assert lib().clite_conv_wgrad_patch(ptr(dyb), ptr(xb), C.byref(cv), ptr(dw), ptr(ws), nb.value - 4, None) == 1
    cv2 = Conv(BF16, N, H, W, Cc, K, 3, 3, 2, 1, (H - 1) // 2 + 1, (W - 1) // 2 + 1)
    assert lib().clite_conv_wgrad_patch(ptr(dyb), ptr(xb), C.byref(cv2), ptr(dw), ptr(ws), nb.value, None) == 1


@pytest.mark.policy_independent
@pytest.mark.parametrize("dtype,N,H,W,Cc,K", [(BF16, 2, 8, 6, 64, 32), (BF16, 3, 6, 10, 136, 64), (F32, 2, 4, 6, 64, 32)])
def test_bn_backward_dgrad_with_subsampled_residual(dtype, N, H, W, Cc, K):
    """clite_epilogue.residual_subsample = 2 (ABI v10): the BatchNorm-backward form of a dense 1 x 1 dgrad whose residual is the COMPACT
    [N][H/2][W/2][C] gradient of the block's stride-2 shortcut — added at the even pixels only, then masked, with both reductions. bf16 with packed
    bits and a bf16 output runs the compile-time FORM 3 of igemm_epilogue_bn, the f32 case the run-time form; row-range persistent launches
    included (the simulator build has 4 resident slots). Entry points that do not implement it refuse."""
    rng = np.random.default_rng(H * W + Cc)
    cv = Conv(dtype, N, H, W, Cc, K, 1, 1, 1, 0, H, W)
    M, P = N * H * W, N * (H // 2) * (W // 2)
    w, _ = _prep(rng.standard_normal((K, 1, 1, Cc), dtype=np.float32) * 0.3, dtype)
    wt, wtb = _prep(np.ascontiguousarray(w.transpose(3, 1, 2, 0)), dtype)
    dy, dyb = _prep(rng.standard_normal((N, H, W, K), dtype=np.float32), dtype)
    aux = rng.standard_normal((M, Cc)).astype(np.float32)
    bits = pack_relu_bits(aux)
    res, resb = _prep(rng.standard_normal((P, Cc), dtype=np.float32), dtype)
    y, yb = _prep(rng.standard_normal((M, Cc), dtype=np.float32) + 2.0, dtype)
    fstats = np.zeros((2, 3, Cc), np.float32)
    fstats[:, 0] = y.sum(0) / 2
    g = conv_dgrad_ref(dy, w, (N, H, W, Cc), 1, 0)
    full = np.zeros((N, H, W, Cc), np.float32)
    full[:, ::2, ::2] = res.reshape(N, H // 2, W // 2, Cc)
    want = ((g + full).reshape(M, Cc)) * (aux > 0)
    out = outbuf_like(dtype, (M, Cc))
    dst = np.zeros((2, 3, Cc), np.float32)
    ep = make_ep(out, Cc, residual=resb, colsum=dst, relu_bits=bits)
    ep.colsum_replicas, ep.colsum_stride = 2, 3 * Cc
    ep.bn_y, ep.bn_stats, ep.bn_replicas, ep.bn_rstride, ep.bn_inv_count, ep.mask_after_residual = ptr(yb), ptr(fstats), 2, 3 * Cc, 1.0 / M, 1
    ep.residual_subsample = 2
    assert lib().clite_conv_dgrad_wt(ptr(dyb), ptr(wtb), C.byref(cv), C.byref(ep), None) == 0
    got = from_bf16(out) if dtype == BF16 else out
    _close(got, want, 8e-3 if dtype == BF16 else 2e-3)
    d = dst.sum(0)
    _close(d[0], got.sum(0), 1e-3)
    _close(d[1], (got * (y - fstats[:, 0].sum(0) / M)).sum(0), 5e-3)
    # refusals: the other entry points, odd extents, a windowed convolution
    assert lib().clite_conv_dgrad(ptr(dyb), ptr(wtb), C.byref(cv), C.byref(ep), None) == -1
    assert lib().clite_conv_fwd(ptr(dyb), ptr(wtb), C.byref(cv), C.byref(ep), None) == -1
    ep.mask_after_residual = 0
    assert lib().clite_conv_dgrad_wt(ptr(dyb), ptr(wtb), C.byref(cv), C.byref(ep), None) == -1


@pytest.mark.policy_independent
def test_f32_split_bf16_form():
    """clite_set_f32_split (ABI v10): the exact-f32 mode's launches with every product formed on three bf16 MFMAs (hi/lo split in registers, f32
    storage and accumulation). Against float64 numpy on f32 inputs: GEMM nt / nn / tn (k-contiguous and k-strided LDS images, ragged edges, a K
    that is not a multiple of the tile), a 3 x 3 convolution forward / input gradient / weight gradient, and the BatchNorm-backward dgrad epilogue.
    Bar: 3e-5 of max|ref| (the split form's ~2^-17 per product; the bf16 path sits at ~4e-3, the exact form at ~1e-6 on these sizes)."""
    L = lib()
    L.clite_set_f32_split.argtypes = [C.c_int]
    rng = np.random.default_rng(11)
    try:
        assert L.clite_set_f32_split(1) == 0 and L.clite_get_f32_split() == 1
        M, N, K = 200, 136, 104
        A = rng.standard_normal((M, K)).astype(np.float32)
        B = rng.standard_normal((N, K)).astype(np.float32)
        out = np.zeros((M, N), np.float32)
        assert L.clite_gemm_nt(ptr(A), K, ptr(B), K, M, N, K, F32, C.byref(make_ep(out, N, out_f32=True)), None) == 0
        ref = A.astype(np.float64) @ B.astype(np.float64).T
        err_nt = np.abs(out - ref).max() / np.abs(ref).max()
        assert err_nt < 3e-5, err_nt
        Bn = np.ascontiguousarray(B.T)
        out2 = np.zeros((M, N), np.float32)
        assert L.clite_gemm_nn(ptr(A), K, ptr(Bn), N, M, N, K, F32, C.byref(make_ep(out2, N, out_f32=True)), None) == 0
        assert np.abs(out2 - ref).max() / np.abs(ref).max() < 3e-5
        At = np.ascontiguousarray(A.T)          # [K][M]
        out3 = np.ones((M, N), np.float32)
        assert L.clite_gemm_tn(ptr(At), M, ptr(Bn), N, M, N, K, F32, C.byref(make_ep(out3, N, out_f32=True, atomic=True)), None) == 0
        assert np.abs(out3 - 1 - ref).max() / np.abs(ref).max() < 3e-5
        # convolution, all three directions, and the BatchNorm-backward epilogue (run-time form)
        Nn, H, W, Cc, Kc = 2, 9, 7, 64, 32
        cv = Conv(F32, Nn, H, W, Cc, Kc, 3, 3, 1, 1, H, W)
        x = rng.standard_normal((Nn, H, W, Cc)).astype(np.float32)
        w = (rng.standard_normal((Kc, 3, 3, Cc)) * 0.2).astype(np.float32)
        dy = rng.standard_normal((Nn, H, W, Kc)).astype(np.float32)
        y = np.zeros((Nn, H, W, Kc), np.float32)
        assert L.clite_conv_fwd(ptr(x), ptr(w), C.byref(cv), C.byref(make_ep(y, Kc, out_f32=True)), None) == 0
        r = conv_ref(x.astype(np.float64), w.astype(np.float64), 1, 1)
        assert np.abs(y - r).max() / np.abs(r).max() < 3e-5
        Mr = Nn * H * W
        aux = rng.standard_normal((Mr, Cc)).astype(np.float32)
        yb = (rng.standard_normal((Mr, Cc)) + 2).astype(np.float32)
        fst = np.zeros((2, 3, Cc), np.float32)
        fst[:, 0] = yb.sum(0) / 2
        dx = np.zeros((Mr, Cc), np.float32)
        dst = np.zeros((2, 3, Cc), np.float32)
        ep = make_ep(dx, Cc, out_f32=True, dact_aux=aux, dact=1, colsum=dst)
        ep.colsum_replicas, ep.colsum_stride = 2, 3 * Cc
        ep.bn_y, ep.bn_stats, ep.bn_replicas, ep.bn_rstride, ep.bn_inv_count = ptr(yb), ptr(fst), 2, 3 * Cc, 1.0 / Mr
        assert L.clite_conv_dgrad(ptr(dy), ptr(w), C.byref(cv), C.byref(ep), None) == 0
        g = conv_dgrad_ref(dy.astype(np.float64), w.astype(np.float64), (Nn, H, W, Cc), 1, 1).reshape(Mr, Cc) * (aux > 0)
        assert np.abs(dx - g).max() / np.abs(g).max() < 3e-5
        assert np.abs(dst.sum(0)[0] - g.sum(0)).max() / np.abs(g.sum(0)).max() < 1e-4
        dw = np.zeros((Kc, 3, 3, Cc), np.float32)
        assert L.clite_conv_wgrad(ptr(dy), ptr(x), C.byref(cv), ptr(dw), None) == 0
        rw = conv_wgrad_ref(dy.astype(np.float64), x.astype(np.float64), (Kc, 3, 3, Cc), 1, 1)
        assert np.abs(dw - rw).max() / np.abs(rw).max() < 3e-5
    finally:
        L.clite_set_f32_split(0)
    # the default (exact) form on the first problem, for scale
    out_e = np.zeros((M, N), np.float32)
    assert L.clite_gemm_nt(ptr(A), K, ptr(B), K, M, N, K, F32, C.byref(make_ep(out_e, N, out_f32=True)), None) == 0
    assert np.abs(out_e - ref).max() / np.abs(ref).max() < 2e-6
