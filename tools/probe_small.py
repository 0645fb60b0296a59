"""Few representative igemm launches for counter collection."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clip_lite_amd import hip
B = 128
def gemm(kind, M, N, K):
    A = torch.randn(M, K, device="cuda").bfloat16() if kind != "tn" else torch.randn(K, M, device="cuda").bfloat16()
    Bm = torch.randn(N, K, device="cuda").bfloat16() if kind == "nt" else torch.randn(K, N, device="cuda").bfloat16()
    out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if kind == "tn" else torch.bfloat16)
    ep = hip.epilogue(out, N, atomic=(kind == "tn"))
    for _ in range(3): getattr(hip, "gemm_" + kind)(hip.BF16, A, Bm, M, N, K, ep)
def conv(N, H, W, Cc, K, R, S, st, pad):
    cv = hip.conv_desc(hip.BF16, N, H, W, Cc, K, R, S, st, pad)
    x = torch.randn(N, H, W, Cc, device="cuda").bfloat16(); w = torch.randn(K, R, S, Cc, device="cuda").bfloat16()
    dy = torch.randn(N, cv.Ho, cv.Wo, K, device="cuda").bfloat16()
    y = torch.empty(N, cv.Ho, cv.Wo, K, device="cuda", dtype=torch.bfloat16); dx = torch.empty_like(x); dw = torch.zeros(K, R, S, Cc, device="cuda")
    for _ in range(3):
        hip.conv_fwd(x, w, cv, hip.epilogue(y, K)); hip.conv_dgrad(dy, w, cv, hip.epilogue(dx, Cc)); hip.conv_wgrad(dy, x, cv, dw)
gemm("nt", 8192, 8192, 8192); gemm("nt", 3840, 3072, 768); gemm("nn", 3840, 768, 3072); gemm("tn", 3072, 768, 3840)
conv(B, 56, 56, 64, 64, 3, 3, 1, 1); conv(B, 28, 28, 128, 128, 3, 3, 1, 1); conv(B, 14, 14, 256, 1024, 1, 1, 1, 0)
torch.cuda.synchronize()
