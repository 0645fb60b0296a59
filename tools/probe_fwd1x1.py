"""The HBM-bound 1 x 1 forward convolutions of ResNet-50's 56 x 56 / 28 x 28 stages (bf16 store + column statistics), batch 128: python tools/probe_fwd1x1.py
(CLITE_HIP_LIB=build/varnpf/libclite_hip_var.so: one tile per workgroup, the form before the row-range persistent one).  us, median of 20, and TB/s"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clip_lite_amd import hip


def timed(fn, n=20):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


for (H, Cc, K, st) in [(56, 64, 256, 1), (56, 256, 64, 1), (56, 64, 64, 1), (28, 128, 512, 1), (28, 512, 128, 1), (56, 256, 128, 1), (56, 256, 512, 2), (14, 256, 1024, 1), (14, 1024, 256, 1)]:
    N = 128
    cv = hip.conv_desc(hip.BF16, N, H, H, Cc, K, 1, 1, st, 0)
    x = torch.randn(N, H, H, Cc, device="cuda").bfloat16()
    w = torch.randn(K, 1, 1, Cc, device="cuda").bfloat16()
    y = torch.empty(N * cv.Ho * cv.Wo, K, device="cuda", dtype=torch.bfloat16)
    stt = hip.Stats(torch.zeros(8 * 3 * K, device="cuda"), 8, K)
    t = timed(lambda: hip.conv_fwd(x, w, cv, hip.epilogue(y, K, colsum=stt)))
    mb = (x.numel() / (st * st) + y.numel()) * 2 / 1e6
    print(f"conv_fwd {Cc:4d}->{K:4d} 1x1/{st} @{H:3d}: {t:7.1f} us   {mb / t:5.2f} TB/s of {mb:6.1f} MB")
