"""Time ONE implicit-GEMM configuration (HIP events, cold Infinity Cache between runs): python tools/probe_time.py conv_fwd N H W C K R S stride pad ...
(same arguments as tools/probe_one.py). Used with `make variant` builds (CLITE_HIP_LIB=...) for same-box A/B of kernel variants and ablations."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clip_lite_amd import hip

kind = sys.argv[1]
a = [int(x) for x in sys.argv[2:]]
N, H, W, Cc, K, R, S, st, pad = a
cv = hip.conv_desc(hip.BF16, N, H, W, Cc, K, R, S, st, pad)
x = torch.randn(N, H, W, Cc, device="cuda").bfloat16()
w = torch.randn(K, R, S, Cc, device="cuda").bfloat16()
dy = torch.randn(N, cv.Ho, cv.Wo, K, device="cuda").bfloat16()
y = torch.empty(N, cv.Ho, cv.Wo, K, device="cuda", dtype=torch.bfloat16)
dx = torch.empty_like(x)
cs = hip.Stats(torch.zeros(8 * 3 * K, device="cuda"), 8, K)
fn = {"conv_fwd": lambda: hip.conv_fwd(x, w, cv, hip.epilogue(y, K, colsum=cs)),
      "conv_dgrad": lambda: hip.conv_dgrad(dy, w, cv, hip.epilogue(dx, Cc))}[kind]
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
print(f"{kind} {' '.join(map(str, a))}: median {ts[len(ts) // 2]:.1f} us (min {ts[0]:.1f})")
