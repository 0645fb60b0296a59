"""The weight gradients of the last chain segment (layer1 + the stride-1 downsample; ResNet-50, per-GPU batch 128), one by one through
clite_conv_wgrad and all together as ONE grouped launch (hip.WgradGroup — what the captured step replays as `wgrad_s2`), with each member's HBM
ideal at 5.5 TB/s: where the 0.8 ms of that group go.   python tools/probe_wgrad_s2.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clip_lite_amd import hip


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


N, H = 128, 56
members = [("conv1 256->64 1x1", 256, 64, 1, 2), ("conv2 64->64 3x3", 64, 64, 3, 3), ("conv3 64->256 1x1", 64, 256, 1, 3),
           ("layer1.0 conv1 64->64 1x1", 64, 64, 1, 1), ("downsample 64->256 1x1", 64, 256, 1, 1)]
items = []
tot_single = tot_ideal = 0.0
for name, Cc, K, R, count in members:
    cv = hip.conv_desc(hip.BF16, N, H, H, Cc, K, R, R, 1, R // 2)
    x = torch.randn(N, H, H, Cc, device="cuda").bfloat16()
    dy = torch.randn(N, H, H, K, device="cuda").bfloat16()
    dw = torch.zeros(K, R, R, Cc, device="cuda")
    t = timeit(lambda: hip.conv_wgrad(dy, x, cv, dw))
    ideal = (x.numel() + dy.numel()) * 2 / 5.5e6
    print(f"{name:28s} x{count}: alone {t:7.1f} us   HBM ideal {ideal:6.1f} us")
    tot_single += t * count
    tot_ideal += ideal * count
    items += [(dy, x, cv, dw)] * count
ws = hip.WgradGroup.alloc_workspace(torch.device("cuda"))


def group():
    g = hip.WgradGroup(hip.BF16, ws)
    for dy, x, cv, dw in items:
        g.conv(dy, x, cv, dw)
    g.launch()


tg = timeit(group)
print(f"sum of the {len(items)} single launches {tot_single:.0f} us; ONE grouped launch {tg:.0f} us; HBM ideal {tot_ideal:.0f} us")
