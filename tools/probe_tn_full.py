import sys, os
sys.path.insert(0, '/root/repo')
import torch
from clip_lite_amd import hip
def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
hip.set_tile_policy(4)
for M, N, K in [(4096, 4096, 3840), (8192, 8192, 3840), (3072, 768, 3840), (3072 * 4, 768 * 3, 3840), (512, 4608, 6272), (512 * 4, 4608 * 4, 6272)]:
    A = torch.randn(K, M, device='cuda').bfloat16(); B = torch.randn(K, N, device='cuda').bfloat16()
    out = torch.zeros(M, N, device='cuda')
    ms = timeit(lambda: hip.gemm_tn(hip.BF16, A, B, M, N, K, hip.epilogue(out, N, atomic=True)))
    print(f"gemm_tn {M}x{N}x{K}: {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:7.1f} TF/s", flush=True)
