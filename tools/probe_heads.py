"""Timing probe: the tiny-M GEMMs of the projection heads / prior discriminators (M = 128 or 256 rows) — 16-32 workgroups, latency-bound.
Usage: python tools/probe_heads.py [tag]   (CLITE_IGEMM_STAGES=4|5 selects deeper LDS rings for A/B runs)"""
import sys
from probe_bert import gemm

if __name__ == "__main__":
    tag = sys.argv[1] if len(sys.argv) > 1 else ""
    tot = 0.0
    for k, M, N, K, n in [("nt", 128, 2048, 2048, 4), ("nn", 128, 2048, 2048, 4), ("tn", 2048, 2048, 128, 4), ("nn", 128, 768, 2048, 2), ("nt", 128, 2048, 768, 2),
                          ("nt", 256, 1000, 2048, 1), ("nt", 256, 200, 1000, 2), ("nn", 256, 1000, 200, 2), ("nn", 128, 2048, 1000, 1), ("nt", 128, 768, 768, 1)]:
        us, tf = gemm(k, M, N, K)
        tot += n * us
        print(f"{tag:8s} gemm_{k} M={M:5d} N={N:5d} K={K:5d} x{n}: {us:7.1f} us {tf:7.1f} TF/s")
    print(f"{tag:8s} weighted sum: {tot:.1f} us per step")
