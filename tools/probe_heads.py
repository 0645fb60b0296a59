"""Timing probe: the tiny-M GEMMs of the projection heads / prior discriminators (M = 128 or 256 rows) — 16-32 workgroups, latency-bound.
Usage: python tools/probe_heads.py [tag]   (CLITE_IGEMM_STAGES=4|5 selects deeper LDS rings for A/B runs)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clip_lite_amd import hip
from probe_bert import timeit


def gemm(kind, M, N, K):
    """nt / nn with a bias epilogue and a split-K workspace (as the heads call them); tn as in probe_bert"""
    A = torch.randn(M, K, device="cuda").bfloat16() if kind != "tn" else torch.randn(K, M, device="cuda").bfloat16()
    B = torch.randn(N, K, device="cuda").bfloat16() if kind == "nt" else torch.randn(K, N, device="cuda").bfloat16()
    out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if kind == "tn" else torch.bfloat16)
    bias = torch.zeros(N, device="cuda")
    ws = torch.zeros(M, N, device="cuda") if kind != "tn" else None      # never re-zeroed here: timing only
    ep = hip.epilogue(out, N, atomic=True) if kind == "tn" else hip.epilogue(out, N, bias=bias, ws=ws)
    f = getattr(hip, "gemm_" + kind)
    ms = timeit(lambda: f(hip.BF16, A, B, M, N, K, ep))
    return ms * 1e3, 2 * M * N * K / ms / 1e9

if __name__ == "__main__":
    tag = sys.argv[1] if len(sys.argv) > 1 else ""
    tot = 0.0
    for k, M, N, K, n in [("nt", 128, 2048, 2048, 4), ("nn", 128, 2048, 2048, 4), ("tn", 2048, 2048, 128, 4), ("nn", 128, 768, 2048, 2), ("nt", 128, 2048, 768, 2),
                          ("nt", 256, 1000, 2048, 1), ("nt", 256, 200, 1000, 2), ("nn", 256, 1000, 200, 2), ("nn", 128, 2048, 1000, 1), ("nt", 128, 768, 768, 1)]:
        us, tf = gemm(k, M, N, K)
        tot += n * us
        print(f"{tag:8s} gemm_{k} M={M:5d} N={N:5d} K={K:5d} x{n}: {us:7.1f} us {tf:7.1f} TF/s")
    print(f"{tag:8s} weighted sum: {tot:.1f} us per step")
