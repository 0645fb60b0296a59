"""What the vendor GEMM (hipBLASLt behind torch.matmul) reaches on the BERT GEMM shapes of the step — a yardstick for the engine's
headroom on this box, not part of the product. Usage: python tools/probe_blaslt.py"""
import torch

SHAPES = [  # (M, N, K, layout) as clite_gemm_* sees them
    (3840, 768, 3072, "nt"), (3840, 3072, 768, "nt"), (3840, 2304, 768, "nt"), (3840, 768, 768, "nt"),
    (3840, 768, 3072, "nn"), (3840, 3072, 768, "nn"), (3840, 768, 2304, "nn"),
    (768, 3072, 3840, "tn"), (3072, 768, 3840, "tn"), (2304, 768, 3840, "tn"), (768, 768, 3840, "tn"),
    (8192, 8192, 8192, "nt"),
]


def main():
    dev = torch.device("cuda", 0)
    for M, N, K, lay in SHAPES:
        if lay == "nt":
            a = torch.randn(M, K, device=dev, dtype=torch.bfloat16); b = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
            f = lambda: a @ b.t()
        elif lay == "nn":
            a = torch.randn(M, K, device=dev, dtype=torch.bfloat16); b = torch.randn(K, N, device=dev, dtype=torch.bfloat16)
            f = lambda: a @ b
        else:
            a = torch.randn(K, M, device=dev, dtype=torch.bfloat16); b = torch.randn(K, N, device=dev, dtype=torch.bfloat16)
            f = lambda: a.t() @ b
        for _ in range(5):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 50
        e0.record()
        for _ in range(n):
            f()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        print(f"{lay} {M:5d} {N:5d} {K:5d}  {us:8.1f} us  {2.0 * M * N * K / us / 1e6:8.1f} TF/s", flush=True)


if __name__ == "__main__":
    main()
