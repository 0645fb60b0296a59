"""Timing probe: the BERT-base GEMM shapes of the step (M = 128 x 30 tokens) through the C ABI. Usage: python tools/probe_bert.py [tag]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clip_lite_amd import hip


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def gemm(kind, M, N, K):
    A = torch.randn(M, K, device="cuda").bfloat16() if kind != "tn" else torch.randn(K, M, device="cuda").bfloat16()
    B = torch.randn(N, K, device="cuda").bfloat16() if kind == "nt" else torch.randn(K, N, device="cuda").bfloat16()
    out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if kind == "tn" else torch.bfloat16)
    ep = hip.epilogue(out, N, atomic=(kind == "tn"))
    f = getattr(hip, "gemm_" + kind)
    ms = timeit(lambda: f(hip.BF16, A, B, M, N, K, ep))
    return ms * 1e3, 2 * M * N * K / ms / 1e9


if __name__ == "__main__":
    tag = sys.argv[1] if len(sys.argv) > 1 else ""
    tot = 0.0
    for k, M, N, K in [("nt", 3840, 768, 768), ("nt", 3840, 2304, 768), ("nt", 3840, 3072, 768), ("nt", 3840, 768, 3072),
                       ("nn", 3840, 768, 768), ("nn", 3840, 768, 2304), ("nn", 3840, 768, 3072), ("nn", 3840, 3072, 768),
                       ("tn", 768, 768, 3840), ("tn", 2304, 768, 3840), ("tn", 3072, 768, 3840), ("tn", 768, 3072, 3840),
                       ("nt", 8192, 8192, 8192)]:
        us, tf = gemm(k, M, N, K)
        if M != 8192:
            tot += us
        print(f"{tag:10s} gemm_{k} M={M:5d} N={N:5d} K={K:5d}: {us:8.1f} us {tf:8.1f} TF/s")
    print(f"{tag:10s} sum over the 12 BERT layer shapes: {tot:.1f} us")
