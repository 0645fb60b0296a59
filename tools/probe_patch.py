"""The patch-resident 3 x 3 conv (csrc/conv_patch.hip) at the step's shape, stand-alone: forward and BatchNorm-backward input gradient, median of 20
launches with a 512 MB cache flush in front of each. With ablation variants of the library (CLITE_HIP_LIB=build/varN/libclite_hip_var.so built by
tools/build_patch_variants.sh: -DCLITE_PATCH_ABLATE=1 no MFMA loop, 2 no global stores, 4 no patch DMA after the first) the differences give the
phases' shares. Usage: python tools/probe_patch.py [batch]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from clip_lite_amd import hip


def timed(fn, flush, n=20):
    ts = []
    for _ in range(n):
        flush.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    H = W = 56
    C = 64
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(N, H, W, C, device="cuda", generator=g).bfloat16()
    dy = torch.randn(N, H, W, C, device="cuda", generator=g).bfloat16()
    w = (torch.randn(C, 3, 3, C, device="cuda", generator=g) * 0.1).bfloat16()
    wt = w.permute(3, 1, 2, 0).contiguous()
    cv = hip.conv_desc(hip.BF16, N, H, W, C, C, 3, 3, 1, 1)
    M = N * H * W
    y = torch.empty(M, C, device="cuda", dtype=torch.bfloat16)
    st = hip.Stats(torch.zeros(4 * 3 * C, device="cuda"), 4, C)
    bits = torch.from_numpy(np.packbits((torch.randn(M, C) > 0).numpy(), axis=-1, bitorder="little")).cuda()
    by = torch.randn(M, C, device="cuda", generator=g).bfloat16()
    fst = hip.Stats(torch.zeros(4 * 3 * C, device="cuda"), 4, C)
    dst = hip.Stats(torch.zeros(4 * 3 * C, device="cuda"), 4, C)
    flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
    ep_f = hip.epilogue(y, C, colsum=st)
    ep_b = hip.epilogue(y, C, relu_bits=bits, colsum=dst, bn=(by, fst, M))
    dw = torch.zeros(C, 3, 3, C, device="cuda")

    def grouped():
        grp = hip.WgradGroup(hip.BF16)
        grp.conv(dy, x, cv, dw)
        grp.launch()
    for pol, name in ((0, "patch-resident"), (4, "implicit GEMM (4-wave)")):
        hip.set_tile_policy(pol)
        tf = timed(lambda: hip.conv_fwd(x, w, cv, ep_f), flush)
        tb = timed(lambda: hip.conv_dgrad(dy, wt, cv, ep_b, wt=True), flush)
        hip.WgradGroup.patch = pol == 0
        tw = timed(grouped, flush)
        print(f"{name:24s} batch {N}: forward {tf:7.1f} us   BatchNorm-backward dgrad {tb:7.1f} us   weight gradient {tw:7.1f} us   (lib {os.path.basename(hip.LIB_PATH)})")
    hip.WgradGroup.patch = False
    hip.set_tile_policy(0)


if __name__ == "__main__":
    main()
