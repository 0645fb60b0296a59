"""Per-workgroup phase timeline of one igemm launch (diagnostic build -DCLITE_STAMP=1, CLITE_HIP_LIB=.../libclite_stamp.so).
Usage: python tools/probe_stamps.py conv_fwd N H W C K R S stride pad | gemm_nt M N K ..."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
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
    cs = hip.Stats(torch.zeros(8 * 3 * K, device="cuda"), 8, K)
    # conv_dgrad_bn: the ResNet backward's fused form (ReLU mask + residual + the two BatchNorm-backward reductions in the epilogue)
    prev_out = torch.randn(N, H, W, Cc, device="cuda").bfloat16()
    prev_y = torch.randn(N, H, W, Cc, device="cuda").bfloat16()
    res = torch.randn(N, H, W, Cc, device="cuda").bfloat16()
    pst = hip.Stats(torch.zeros(8 * 3 * Cc, device="cuda"), 8, Cc)
    dst = hip.Stats(torch.zeros(8 * 3 * Cc, device="cuda"), 8, Cc)
    fn = {"conv_fwd": lambda: hip.conv_fwd(x, w, cv, hip.epilogue(y, K, colsum=cs)),
          "conv_dgrad": lambda: hip.conv_dgrad(dy, w, cv, hip.epilogue(dx, Cc)),
          "conv_dgrad_bn": lambda: hip.conv_dgrad(dy, w, cv, hip.epilogue(dx, Cc, dact_aux=prev_out, dact=hip.DACT_RELU, colsum=dst,
                                                                           bn=(prev_y.view(-1, Cc), pst, N * H * W))),
          "conv_dgrad_bnres": lambda: hip.conv_dgrad(dy, w, cv, hip.epilogue(dx, Cc, residual=res, dact_aux=prev_out, dact=hip.DACT_RELU,
                                                                              mask_after_residual=True, colsum=dst,
                                                                              bn=(prev_y.view(-1, Cc), pst, N * H * W)))}[kind]
else:
    M, N, K = a[:3]
    k = kind.partition(":")[0][-2:]
    A = torch.randn(M, K, device="cuda").bfloat16()
    B = torch.randn(N, K, device="cuda").bfloat16() if k == "nt" else torch.randn(K, N, device="cuda").bfloat16()
    out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    # epilogue form after the layout: gemm_nt, gemm_nt:gelu (bias + GELU + pre-activation store), gemm_nt:bdr (bias + dropout + residual),
    # gemm_nn:dgelu (x GELU'(pre-activation)), gemm_nn:res (+ residual)
    kind, _, form = kind.partition(":")
    k = kind[-2:]
    bias = torch.randn(N, device="cuda")
    pre = torch.randn(M, N, device="cuda").bfloat16()
    res = torch.randn(M, N, device="cuda").bfloat16()
    ep = {"": lambda: hip.epilogue(out, N),
          "bias": lambda: hip.epilogue(out, N, bias=bias),
          "gelu": lambda: hip.epilogue(out, N, bias=bias, act=hip.ACT_GELU, preact=pre),
          "bdr": lambda: hip.epilogue(out, N, bias=bias, drop=(0.1, 1234, 7), residual=res),
          "dgelu": lambda: hip.epilogue(out, N, dact_aux=pre, dact=hip.DACT_GELU),
          "res": lambda: hip.epilogue(out, N, residual=res)}[form]()
    if len(sys.argv) > 5:
        hip.set_tile_policy(int(sys.argv[5]))
        a = a[:3]
    f = getattr(hip, kind)
    fn = lambda: f(hip.BF16, A, B, M, N, K, ep)
os.environ.setdefault("CLITE_IGEMM_G2", "0")
for _ in range(3):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); fn(); e1.record(); torch.cuda.synchronize()
n = 8 * 8192
buf = (C.c_ulonglong * n)()
lib = hip.lib()
lib.clite_dbg_read.argtypes = [C.c_void_p, C.c_int]
assert lib.clite_dbg_read(buf, n) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 8).astype(np.int64)
t = t[t[:, 0] > 0]
t0 = t[:, 0].min()
us = (t[:, :8] - t0) / 100.0
print(f"{kind} {a}: kernel {e0.elapsed_time(e1) * 1e3:.1f} us, {len(us)} workgroups stamped")
names = ["start", "init done", "prologue issued", "first tile landed", "main loop done", "epilogue done"]
d = np.diff(us, axis=1)
print("phase durations per workgroup (us): median / p90")
for i, nm in enumerate(["init (addresses, divisions)", "prologue DMA issue", "wait first tile", "main loop", "epilogue"]):
    print(f"  {nm:30s} {np.median(d[:, i]):7.2f} / {np.percentile(d[:, i], 90):7.2f}")
print(f"  epilogue: staged+barrier (pass 0)  {np.median(us[:, 6] - us[:, 4]):7.2f}   rows of pass 0 written {np.median(us[:, 7] - us[:, 6]):7.2f}   rest {np.median(us[:, 5] - us[:, 7]):7.2f}")
print(f"  workgroup lifetime             {np.median(us[:, 5] - us[:, 0]):7.2f} / {np.percentile(us[:, 5] - us[:, 0], 90):7.2f}")
print(f"  start times: min {us[:, 0].min():.1f} median {np.median(us[:, 0]):.1f} max {us[:, 0].max():.1f};  last end {us[:, 5].max():.1f}")
