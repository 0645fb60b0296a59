"""The BERT GEMM shapes of the step by row count, on whatever library CLITE_HIP_LIB names (default: the product's): how the time of the 8-wave kernels depends
on the number of 128 x 128 tiles (90 .. 360 for N = 768) — with `make variant VAR_EXTRA=-DCLITE_ABLATE=1|2|3` builds also the memory side, the compute side
and the bare MFMA loop of csrc/igemm_wide.h (profiles/r5_wide_ablation.txt).   python tools/probe_rows768.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from clip_lite_amd import hip
from probe_bn import timeit
for (N, K) in [(768, 3072), (768, 768), (768, 2304), (3072, 768), (2304, 768)]:
    for M in [1920, 2560, 3840, 4608, 5376, 5504, 7680]:
        A = torch.randn(M, K, device="cuda").bfloat16()
        B = torch.randn(N, K, device="cuda").bfloat16()
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        bias = torch.zeros(N, device="cuda")
        ep = hip.epilogue(out, N, bias=bias)
        t = timeit(lambda: hip.gemm_nt(hip.BF16, A, B, M, N, K, ep), iters=50)
        tiles = ((M + 127) // 128) * ((N + 127) // 128)
        print(f"M={M:5d} N={N:5d} K={K:5d}: {t:6.1f} us  {2*M*N*K/t/1e6:6.0f} TF/s   128x128 tiles {tiles}")
