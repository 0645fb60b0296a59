"""The stem's kernels one by one at the benchmark's size (batch 128, 224 x 224; bf16): python tools/probe_stem.py [batch]   (times in us, median of 20)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clip_lite_amd import hip

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
H = W = 224
Ho = Wo = 112
Hp, Wp = H + 6, W + 8
dt = hip.BF16


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


img = torch.randn(N, 3, H, W, device="cuda")
xpad = torch.empty(N, Hp, Wp, 4, device="cuda", dtype=torch.bfloat16)
print(f"image_to_nhwc4            {timed(lambda: hip.image_to_nhwc4(dt, img, xpad, N, H, W, 3, Hp, Wp)):8.1f} us   (115 MB)")
dy = (torch.randn(N, Ho, Wo, 64, device="cuda") * 0.1).bfloat16()
dwv = torch.zeros(64, 7, 8, 4, device="cuda")
dw = torch.zeros(64, 7, 7, 3, device="cuda")
print(f"stem_wgrad + unpack       {timed(lambda: (dwv.zero_(), hip.stem_wgrad(dt, dy, xpad, N, Hp, Wp, Ho, Wo, dwv), hip.stem_unpack_grad(dwv, dw))):8.1f} us   (243 MB)")
hip.patch_workspace(torch.device('cuda', 0))
print(f"stem_wgrad_patch          {timed(lambda: hip.stem_wgrad_patch(dt, dy, xpad, N, Hp, Wp, Ho, Wo, dw)):8.1f} us")

# bn1's backward + conv1's weight gradient: two kernels (the un-pooled gradient through memory) against the fused one
Hq = Wq = 56
y = (torch.randn(N * Ho * Wo, 64, device="cuda") * 2).bfloat16()
st = hip.Stats(torch.zeros(8 * 3 * 64, device="cuda"), 8, 64)
st.t.view(8, 3, 64)[0, 0] = y.float().sum(0)
st.t.view(8, 3, 64)[0, 1] = (y.float() ** 2).sum(0)
gamma, beta, rm, rv = torch.ones(64, device="cuda"), torch.zeros(64, device="cuda"), torch.zeros(64, device="cuda"), torch.ones(64, device="cuda")
desc = lambda: hip.bn_desc(N * Ho * Wo, 64, st, gamma, beta, rm, rv, True, False, 0.1, 1e-5, False)
p0 = torch.empty(N * Hq * Wq, 64, device="cuda", dtype=torch.bfloat16)
idx = torch.empty(N * Hq * Wq, 64, device="cuda", dtype=torch.uint8)
print(f"stem_bn_pool_fwd          {timed(lambda: hip.stem_bn_pool_fwd(dt, desc(), y, p0, idx, N, Ho, Wo)):8.1f} us")
dpool = (torch.randn(N * Hq * Wq, 64, device="cuda") * 0.1).bfloat16()
ds = hip.Stats(torch.zeros(8 * 3 * 64, device="cuda"), 8, 64)
ds.t.view(8, 3, 64)[0, 0] = 1.0
dy0 = torch.empty(N * Ho * Wo, 64, device="cuda", dtype=torch.bfloat16)
dg, db = torch.zeros(64, device="cuda"), torch.zeros(64, device="cuda")
print(f"stem_bn_pool_bwd_apply    {timed(lambda: hip.stem_bn_pool_bwd_apply(dt, desc(), dpool, idx, y, ds, dy0, dg, db, N, Ho, Wo)):8.1f} us")
