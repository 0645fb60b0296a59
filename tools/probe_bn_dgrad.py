"""Time the BatchNorm-backward conv dgrad (the ResNet backward's fused form) at one shape, cold caches: python tools/probe_bn_dgrad.py N H W C K R stride pad [resid]
Used with `make variant VAR_EXTRA=-DCLITE_ABLATE=1|2` builds to split a launch into its memory side, its compute side and its epilogue."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clip_lite_amd import hip

N, H, W, Cc, K, R, st, pad = [int(x) for x in sys.argv[1:9]]
resid = len(sys.argv) > 9
cv = hip.conv_desc(hip.BF16, N, H, W, Cc, K, R, R, st, pad)
M = N * H * W
wt = torch.randn(Cc, R, R, K, device="cuda").bfloat16()
dy = torch.randn(N, cv.Ho, cv.Wo, K, device="cuda").bfloat16()
y = torch.randn(M, Cc, device="cuda").bfloat16()
res = torch.randn(M, Cc, device="cuda").bfloat16()
bits = torch.randint(0, 255, (M, Cc // 8), device="cuda", dtype=torch.uint8)
dz = torch.empty(M, Cc, device="cuda", dtype=torch.bfloat16)
fst = hip.Stats(torch.zeros(8, 3, Cc, device="cuda"), 8, Cc)
dst = hip.Stats(torch.zeros(8, 3, Cc, device="cuda"), 8, Cc)
fn = lambda: hip.conv_dgrad(dy, wt, cv, hip.epilogue(dz, Cc, residual=res if resid else None, relu_bits=bits, mask_after_residual=resid, colsum=dst, bn=(y, fst, M)), wt=True)
if os.environ.get("CLITE_TILE_POLICY"):       # 1 / 2 / 3: force the 8-wave 128 x 128 / 256 x 128 / 256 x 256 tile, 4: the 4-wave kernels
    hip.set_tile_policy(int(os.environ["CLITE_TILE_POLICY"]))
junk = torch.empty(300 * 1024 * 1024, device="cuda", dtype=torch.uint8)
ts = []
for i in range(8):
    junk.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
ts = sorted(ts[2:])
print(f"bn dgrad {Cc}<-{K} {R}x{R} @{H} resid={int(resid)}: median {ts[len(ts) // 2]:.1f} us (min {ts[0]:.1f})")
if os.environ.get("CLITE_PHASES"):          # a -DCLITE_STAMP=1 variant build: accumulated phase durations of every workgroup (igemm_dma_bn_kernel PHASE)
    import ctypes as C
    import numpy as np
    n = 8 * 1024
    buf = (C.c_ulonglong * n)()
    lib = hip.lib()
    lib.clite_dbg_read.argtypes = [C.c_void_p, C.c_int]
    assert lib.clite_dbg_read(buf, n) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 8).astype(np.float64)
    t = t[t[:, 7] > 0]
    tiles = t[:, 7]
    print(f"  {len(t)} workgroups, {np.median(tiles):.1f} tiles each; phase sums per workgroup in us (median), per tile in brackets")
    for i, nm in enumerate(["launch prologue (BN means)", "tile setup + early requests + prefill issue", "wait first operand tile", "main loop", "stage accumulators + first requests",
                            "row loop", "column-sum fold + atomics"]):
        v = t[:, i] / 100.0
        per = f"[{np.median(v / tiles):6.2f}]" if 1 <= i <= 5 else ""
        print(f"    {nm:46s} {np.median(v):7.2f} {per}")
    print(f"    total {np.median(t[:, :7].sum(axis=1)) / 100.0:7.2f}")
