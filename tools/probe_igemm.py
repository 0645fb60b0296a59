"""Timing probe for the implicit-GEMM engine at the shapes of the ResNet-50 + BERT-base step (per-GPU batch 128)."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clip_lite_amd import hip

L = hip.lib()

def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

def gemm(kind, M, N, K):
    A = torch.randn(M, K, device="cuda").bfloat16() if kind != "tn" else torch.randn(K, M, device="cuda").bfloat16()
    B = torch.randn(N, K, device="cuda").bfloat16() if kind == "nt" else torch.randn(K, N, device="cuda").bfloat16()
    out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if kind == "tn" else torch.bfloat16)
    ep = hip.epilogue(out, N, atomic=(kind == "tn"))
    f = getattr(hip, "gemm_" + kind)
    ms = timeit(lambda: f(hip.BF16, A, B, M, N, K, ep))
    print(f"gemm_{kind} M={M} N={N} K={K}: {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:8.1f} TFLOP/s")

def conv(N, H, W, Cc, K, R, S, st, pad):
    Ho = (H + 2 * pad - R) // st + 1; Wo = (W + 2 * pad - S) // st + 1
    cv = hip.conv_desc(hip.BF16, N, H, W, Cc, K, R, S, st, pad)
    x = torch.randn(N, H, W, Cc, device="cuda").bfloat16(); w = torch.randn(K, R, S, Cc, device="cuda").bfloat16()
    dy = torch.randn(N, Ho, Wo, K, device="cuda").bfloat16()
    y = torch.empty(N, Ho, Wo, K, device="cuda", dtype=torch.bfloat16); dx = torch.empty_like(x)
    dw = torch.zeros(K, R, S, Cc, device="cuda")
    cs = hip.Stats(torch.zeros(8 * 3 * K, device="cuda"), 8, K)
    epy = hip.epilogue(y, K, colsum=cs); epx = hip.epilogue(dx, Cc)
    fl = 2.0 * N * Ho * Wo * K * R * S * Cc
    t1 = timeit(lambda: hip.conv_fwd(x, w, cv, epy))
    t2 = timeit(lambda: hip.conv_dgrad(dy, w, cv, epx))
    t3 = timeit(lambda: hip.conv_wgrad(dy, x, cv, dw))
    print(f"conv N={N} {H}x{W} C={Cc} K={K} {R}x{S}/{st}: fwd {t1*1e3:7.1f} us {fl/t1/1e9:6.1f} TF | dgrad {t2*1e3:7.1f} us {fl/t2/1e9:6.1f} TF | wgrad {t3*1e3:7.1f} us {fl/t3/1e9:6.1f} TF")

if __name__ == "__main__":
    B = 128
    for k, M, N, K in [("nt", 3840, 768, 768), ("nt", 3840, 2304, 768), ("nt", 3840, 3072, 768), ("nt", 3840, 768, 3072),
                       ("nn", 3840, 768, 768), ("nn", 3840, 768, 3072), ("nn", 3840, 3072, 768),
                       ("tn", 768, 768, 3840), ("tn", 3072, 768, 3840), ("tn", 768, 3072, 3840),
                       ("nt", 128, 2048, 2048), ("nn", 128, 2048, 2048), ("tn", 2048, 2048, 128), ("nt", 8192, 8192, 8192)]:
        gemm(k, M, N, K)
    for c in [(B, 56, 56, 64, 64, 1, 1, 1, 0), (B, 56, 56, 64, 64, 3, 3, 1, 1), (B, 56, 56, 64, 256, 1, 1, 1, 0), (B, 56, 56, 256, 64, 1, 1, 1, 0),
              (B, 56, 56, 256, 128, 1, 1, 1, 0), (B, 56, 56, 128, 128, 3, 3, 2, 1), (B, 28, 28, 128, 512, 1, 1, 1, 0), (B, 28, 28, 512, 128, 1, 1, 1, 0),
              (B, 28, 28, 128, 128, 3, 3, 1, 1), (B, 56, 56, 256, 512, 1, 1, 2, 0),
              (B, 14, 14, 256, 256, 3, 3, 1, 1), (B, 14, 14, 256, 1024, 1, 1, 1, 0), (B, 14, 14, 1024, 256, 1, 1, 1, 0),
              (B, 7, 7, 512, 512, 3, 3, 1, 1), (B, 7, 7, 512, 2048, 1, 1, 1, 0), (B, 7, 7, 2048, 512, 1, 1, 1, 0)]:
        conv(*c)
