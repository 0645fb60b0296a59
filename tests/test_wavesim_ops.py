"""CPU-side (wave simulator) checks of the HBM-bound kernels against numpy. Runs without a GPU."""
import ctypes as C

import numpy as np
import pytest

from simlib import Bn, lib, outbuf, prep, ptr, val

BF16, F32 = 0, 1


def _close(got, ref, rel):
    assert np.abs(got - ref).max() <= rel * max(np.abs(ref).max(), 1e-6), np.abs(got - ref).max()


def _tol(dtype):
    return 1e-2 if dtype == BF16 else 2e-5


@pytest.mark.parametrize("dtype", [BF16, F32])
@pytest.mark.parametrize("M,Cc,dual", [(70, 64, False), (33, 256, True), (9, 2048, False)])
def test_bn_apply_and_backward(dtype, M, Cc, dual):
    rng = np.random.default_rng(M)
    y, yb = prep(rng.standard_normal((M, Cc), dtype=np.float32) * 2 + 0.5, dtype)
    r, rb = prep(rng.standard_normal((M, Cc), dtype=np.float32), dtype)
    gamma = (1 + 0.1 * rng.standard_normal(Cc)).astype(np.float32)
    beta = (0.1 * rng.standard_normal(Cc)).astype(np.float32)
    g2 = (1 + 0.1 * rng.standard_normal(Cc)).astype(np.float32)
    b2 = (0.1 * rng.standard_normal(Cc)).astype(np.float32)
    stats = np.stack([y.sum(0), (y * y).sum(0)]).astype(np.float32)
    rstats = np.stack([r.sum(0), (r * r).sum(0)]).astype(np.float32)
    rm = np.zeros(Cc, np.float32); rv = np.ones(Cc, np.float32)
    rm2 = np.zeros(Cc, np.float32); rv2 = np.ones(Cc, np.float32)
    p = Bn(M, Cc, ptr(stats), ptr(gamma), ptr(beta), ptr(rm), ptr(rv), 1, 1, 0.1, 1e-5, 1,
           ptr(rstats) if dual else None, ptr(g2) if dual else None, ptr(b2) if dual else None,
           ptr(rm2) if dual else None, ptr(rv2) if dual else None)
    out = outbuf((M, Cc), dtype)
    assert lib().clite_bn_apply(C.byref(p), dtype, ptr(yb), ptr(rb), ptr(out), None) == 0
    mean, var = y.mean(0), y.var(0)
    xhat = (y - mean) / np.sqrt(var + 1e-5)
    z = xhat * gamma + beta
    if dual:
        z = z + (r - r.mean(0)) / np.sqrt(r.var(0) + 1e-5) * g2 + b2
    else:
        z = z + r
    ref = np.maximum(z, 0)
    _close(val(out, dtype), ref, _tol(dtype))
    _close(rm, 0.1 * mean, 1e-4)
    _close(rv, 0.9 + 0.1 * var * M / (M - 1), 1e-4)

    # backward of the main branch: dz = dout * (out > 0); dy = gamma*rstd*(dz - mean(dz) - xhat*mean(dz*xhat))
    outv = val(out, dtype)
    dout, doutb = prep(rng.standard_normal((M, Cc), dtype=np.float32), dtype)
    dstats = np.zeros((2, Cc), np.float32)
    assert lib().clite_bn_bwd_reduce(dtype, ptr(doutb), ptr(out), ptr(yb), ptr(dstats), M, Cc, None) == 0
    dz = dout * (outv > 0)
    _close(dstats[0], dz.sum(0), 1e-4)
    _close(dstats[1], (dz * y).sum(0), 1e-4)
    dy = outbuf((M, Cc), dtype); dzb = outbuf((M, Cc), dtype)
    dg = np.ones(Cc, np.float32); db = np.ones(Cc, np.float32)
    assert lib().clite_bn_bwd_apply(C.byref(p), dtype, ptr(doutb), ptr(out), ptr(yb), ptr(dstats), ptr(dy), ptr(dzb), ptr(dg), ptr(db), None) == 0
    rstd = 1 / np.sqrt(var + 1e-5)
    dyref = gamma * rstd * (dz - dz.mean(0) - xhat * (dz * xhat).mean(0))
    _close(val(dy, dtype), dyref, _tol(dtype))
    _close(val(dzb, dtype), dz, _tol(dtype))
    _close(dg, 1 + (dz * xhat).sum(0), 1e-3)
    _close(db, 1 + dz.sum(0), 1e-3)


@pytest.mark.parametrize("dtype", [BF16, F32])
def test_pools_and_image(dtype):
    rng = np.random.default_rng(3)
    N, H, W, Cc = 2, 9, 8, 16
    x, xb = prep(np.round(rng.standard_normal((N, H, W, Cc), dtype=np.float32) * 2) / 2, dtype)  # coarse values -> ties
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    out = outbuf((N, Ho, Wo, Cc), dtype); idx = np.zeros((N, Ho, Wo, Cc), np.uint8)
    assert lib().clite_maxpool3x3s2_fwd(dtype, ptr(xb), ptr(out), ptr(idx), N, H, W, Cc, None) == 0
    xp = np.full((N, H + 2, W + 2, Cc), -np.inf, np.float32); xp[:, 1:-1, 1:-1] = x
    ref = np.full((N, Ho, Wo, Cc), -np.inf, np.float32); ridx = np.zeros((N, Ho, Wo, Cc), np.int64)
    for r in range(3):
        for s in range(3):
            v = xp[:, r:r + 2 * Ho:2, s:s + 2 * Wo:2]
            better = v > ref
            ridx = np.where(better, r * 3 + s, ridx); ref = np.where(better, v, ref)
    assert np.array_equal(val(out, dtype), ref)
    assert np.array_equal(idx, ridx)
    dout, doutb = prep(rng.standard_normal((N, Ho, Wo, Cc), dtype=np.float32), dtype)
    dx = outbuf((N, H, W, Cc), dtype)
    assert lib().clite_maxpool3x3s2_bwd(dtype, ptr(doutb), ptr(idx), ptr(dx), N, H, W, Cc, None) == 0
    dxp = np.zeros((N, H + 2, W + 2, Cc), np.float32)
    for r in range(3):
        for s in range(3):
            dxp[:, r:r + 2 * Ho:2, s:s + 2 * Wo:2] += dout * (ridx == r * 3 + s)
    _close(val(dx, dtype), dxp[:, 1:-1, 1:-1], _tol(dtype))

    a = outbuf((N, Cc), dtype)
    assert lib().clite_avgpool_fwd(dtype, ptr(xb), ptr(a), N, H * W, Cc, None) == 0
    _close(val(a, dtype), x.reshape(N, -1, Cc).mean(1), _tol(dtype))
    da, dab = prep(rng.standard_normal((N, Cc), dtype=np.float32), dtype)
    dxa = outbuf((N, H * W, Cc), dtype)
    assert lib().clite_avgpool_bwd(dtype, ptr(dab), ptr(dxa), N, H * W, Cc, None) == 0
    _close(val(dxa, dtype), np.repeat(da[:, None, :], H * W, 1) / (H * W), _tol(dtype))

    img = rng.standard_normal((N, 3, 6, 5), dtype=np.float32)
    o = outbuf((N, 12, 16, 4), dtype)
    assert lib().clite_image_to_nhwc4(dtype, ptr(img), ptr(o), N, 6, 5, 3, 12, 16, None) == 0
    refi = np.zeros((N, 12, 16, 4), np.float32); refi[:, 3:9, 3:8, :3] = img.transpose(0, 2, 3, 1)
    _close(val(o, dtype), refi, _tol(dtype))

    cs = np.ones(Cc, np.float32)
    assert lib().clite_colsum(dtype, ptr(xb), ptr(cs), N * H * W, Cc, None) == 0
    _close(cs, 1 + x.reshape(-1, Cc).sum(0), 1e-4)
