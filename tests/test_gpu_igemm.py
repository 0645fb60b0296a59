"""GPU parity of the implicit-GEMM engine (through the C ABI) against torch fp32 references on the same
bf16-rounded inputs. Tolerance: the kernel accumulates in fp32 from exact bf16 products, so against an fp32
reference computed from the same bf16 inputs the error is summation-order only: <= 2e-3 relative to max|ref| for
fp32 outputs; bf16 outputs add one rounding (2^-8 relative)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _hip():
    from clip_lite_amd import hip
    return hip


def _rel(got, ref):
    return ((got.float() - ref).abs().max() / ref.abs().max().clamp_min(1e-6)).item()


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 136, 104), (300, 64, 32), (70, 1000, 200), (3840, 768, 768), (128, 2048, 2048)])
def test_gemm_nt_bias_relu(M, N, K):
    hip = _hip()
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    B = torch.randn(N, K, device="cuda", generator=g).bfloat16()
    bias = torch.randn(N, device="cuda", generator=g)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    pre = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ep = hip.epilogue(out, N, bias=bias, act=hip.ACT_RELU, preact=pre)
    hip.check(hip.lib().clite_gemm_nt(hip.p(A), hip.p(B), M, N, K, C.byref(ep), hip.stream_ptr()), "gemm_nt")
    ref = A.float() @ B.float().t() + bias
    assert _rel(pre, ref) < 6e-3
    assert _rel(out, ref.relu()) < 6e-3


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 136, 104), (300, 64, 40), (3840, 768, 3072)])
def test_gemm_nn(M, N, K):
    hip = _hip()
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    B = torch.randn(K, N, device="cuda", generator=g).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.float32)
    ep = hip.epilogue(out, N)
    hip.check(hip.lib().clite_gemm_nn(hip.p(A), hip.p(B), M, N, K, C.byref(ep), hip.stream_ptr()), "gemm_nn")
    assert _rel(out, A.float() @ B.float()) < 2e-3


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (72, 136, 300), (256, 8, 1000), (768, 3072, 3840)])
def test_gemm_tn_atomic(M, N, K):
    hip = _hip()
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = torch.randn(K, M, device="cuda", generator=g).bfloat16()
    B = torch.randn(K, N, device="cuda", generator=g).bfloat16()
    out = torch.ones(M, N, device="cuda", dtype=torch.float32)
    ep = hip.epilogue(out, N, atomic=True)
    hip.check(hip.lib().clite_gemm_tn(hip.p(A), hip.p(B), M, N, K, C.byref(ep), hip.stream_ptr()), "gemm_tn")
    assert _rel(out, 1 + A.float().t() @ B.float()) < 2e-3


CONV_CASES = [
    (2, 8, 8, 32, 64, 3, 3, 1, 1), (2, 9, 7, 64, 32, 3, 3, 2, 1), (3, 6, 6, 64, 128, 1, 1, 1, 0),
    (2, 8, 8, 32, 64, 1, 1, 2, 0), (1, 10, 10, 32, 160, 3, 3, 1, 1),
    (8, 56, 56, 64, 64, 3, 3, 1, 1), (8, 56, 56, 256, 128, 1, 1, 1, 0), (8, 28, 28, 128, 128, 3, 3, 2, 1),
    (4, 14, 14, 1024, 2048, 1, 1, 2, 0), (4, 7, 7, 512, 512, 3, 3, 1, 1),
]


@pytest.mark.parametrize("N,H,W,Cc,K,R,S,st,pad", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(N, H, W, Cc, K, R, S, st, pad):
    hip = _hip()
    Ho = (H + 2 * pad - R) // st + 1
    Wo = (W + 2 * pad - S) // st + 1
    cv = hip.Conv(N, H, W, Cc, K, R, S, st, pad, Ho, Wo)
    g = torch.Generator(device="cuda").manual_seed(N * H + Cc + K)
    x = torch.randn(N, H, W, Cc, device="cuda", generator=g).bfloat16()
    w = (torch.randn(K, R, S, Cc, device="cuda", generator=g) * 0.1).bfloat16()
    dy = torch.randn(N, Ho, Wo, K, device="cuda", generator=g).bfloat16()
    x32 = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    w32 = w.float().permute(0, 3, 1, 2).requires_grad_(True)
    with torch.backends.cudnn.flags(enabled=False):
        yref = F.conv2d(x32, w32, stride=st, padding=pad)
    yref.backward(dy.float().permute(0, 3, 1, 2))
    yref = yref.detach().permute(0, 2, 3, 1)

    y = torch.empty(N, Ho, Wo, K, device="cuda", dtype=torch.float32)
    cs = torch.zeros(2, K, device="cuda")
    ep = hip.epilogue(y, K, colsum=cs)
    hip.check(hip.lib().clite_conv_fwd(hip.p(x), hip.p(w), C.byref(cv), C.byref(ep), hip.stream_ptr()), "conv_fwd")
    assert _rel(y, yref) < 2e-3
    assert _rel(cs[0], yref.reshape(-1, K).sum(0)) < 2e-3
    assert _rel(cs[1], (yref.reshape(-1, K) ** 2).sum(0)) < 2e-3

    dx = torch.empty(N, H, W, Cc, device="cuda", dtype=torch.float32)
    ep = hip.epilogue(dx, Cc)
    hip.check(hip.lib().clite_conv_dgrad(hip.p(dy), hip.p(w), C.byref(cv), C.byref(ep), hip.stream_ptr()), "conv_dgrad")
    assert _rel(dx, x32.grad.permute(0, 2, 3, 1)) < 2e-3

    dw = torch.zeros(K, R, S, Cc, device="cuda", dtype=torch.float32)
    hip.check(hip.lib().clite_conv_wgrad(hip.p(dy), hip.p(x), C.byref(cv), hip.p(dw), hip.stream_ptr()), "conv_wgrad")
    assert _rel(dw, w32.grad.permute(0, 2, 3, 1)) < 2e-3
