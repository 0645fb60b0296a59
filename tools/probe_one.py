"""Run ONE implicit-GEMM configuration a few times (for rocprofv3 --pmc passes). Usage:
   python tools/probe_one.py conv_fwd N H W C K R S stride pad | conv_dgrad ... | conv_wgrad ... | gemm_nt M N K | gemm_nn M N K | gemm_tn M N K"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clip_lite_amd import hip

kind = sys.argv[1]
a = [int(x) for x in sys.argv[2:]]
if kind.startswith("conv"):
    N, H, W, Cc, K, R, S, st, pad = a
    cv = hip.conv_desc(hip.BF16, N, H, W, Cc, K, R, S, st, pad)
    x = torch.randn(N, H, W, Cc, device="cuda").bfloat16()
    w = torch.randn(K, R, S, Cc, device="cuda").bfloat16()
    dy = torch.randn(N, cv.Ho, cv.Wo, K, device="cuda").bfloat16()
    y = torch.empty(N, cv.Ho, cv.Wo, K, device="cuda", dtype=torch.bfloat16)
    dx = torch.empty_like(x)
    dw = torch.zeros(K, R, S, Cc, device="cuda")
    cs = hip.Stats(torch.zeros(8 * 3 * K, device="cuda"), 8, K)
    fn = {"conv_fwd": lambda: hip.conv_fwd(x, w, cv, hip.epilogue(y, K, colsum=cs)),
          "conv_dgrad": lambda: hip.conv_dgrad(dy, w, cv, hip.epilogue(dx, Cc)),
          "conv_wgrad": lambda: hip.conv_wgrad(dy, x, cv, dw)}[kind]
else:
    M, N, K = a
    k = kind[-2:]
    A = torch.randn(M, K, device="cuda").bfloat16() if k != "tn" else torch.randn(K, M, device="cuda").bfloat16()
    B = torch.randn(N, K, device="cuda").bfloat16() if k == "nt" else torch.randn(K, N, device="cuda").bfloat16()
    out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if k == "tn" else torch.bfloat16)
    ep = hip.epilogue(out, N, atomic=(k == "tn"))
    f = getattr(hip, kind)
    fn = lambda: f(hip.BF16, A, B, M, N, K, ep)
junk = torch.empty(300 * 1024 * 1024, device="cuda", dtype=torch.uint8)
for _ in range(5):
    junk.zero_()          # push the operands out of the 256 MB Infinity Cache between runs
    fn()
torch.cuda.synchronize()
