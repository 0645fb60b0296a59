"""Timing probe: the BatchNorm kernels (apply / backward reduce / backward apply) at the ResNet-50 shapes of the step (batch 128)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clip_lite_amd import hip


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


if __name__ == "__main__":
    tot = [0, 0, 0]
    # (M, C, count in ResNet-50 incl. downsample units)
    for M, C, n in [(1605632, 64, 1), (401408, 64, 6), (401408, 256, 4), (100352, 128, 8), (100352, 512, 5), (25088, 256, 12), (25088, 1024, 7),
                    (6272, 512, 6), (6272, 2048, 4)]:
        y = torch.randn(M, C, device="cuda").bfloat16()
        res = torch.randn(M, C, device="cuda").bfloat16()
        out = torch.empty_like(y)
        dout = torch.randn(M, C, device="cuda").bfloat16()
        dy = torch.empty_like(y)
        st = hip.Stats(torch.zeros(8 * 3 * C, device="cuda"), 8, C)
        st.t[:C] = y.float().sum(0)
        st.t[C:2 * C] = (y.float() ** 2).sum(0)
        dst = hip.Stats(torch.zeros(8 * 3 * C, device="cuda"), 8, C)
        g, b, rm, rv = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
        dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        d1 = hip.bn_desc(M, C, st, g, b, rm, rv, True, True, 0.1, 1e-5, True)
        d2 = hip.bn_desc(M, C, st, g, b, rm, rv, True, False, 0.1, 1e-5, False)
        t1 = timeit(lambda: hip.bn_apply(hip.BF16, d1, y, res, out))
        bits = torch.empty(M, C // 8, device="cuda", dtype=torch.uint8)
        d1b = hip.bn_desc(M, C, st, g, b, rm, rv, True, True, 0.1, 1e-5, True, relu_bits=bits)
        t1b = timeit(lambda: hip.bn_apply(hip.BF16, d1b, y, res, out))
        t2b = timeit(lambda: hip.bn_bwd_reduce(hip.BF16, dout, bits, y, st, dst, M, C))
        t3b = timeit(lambda: hip.bn_bwd_apply(hip.BF16, d2, dout, bits, y, dst, dy, None, dg, db))
        t2 = timeit(lambda: hip.bn_bwd_reduce(hip.BF16, dout, out, y, st, dst, M, C))
        t3 = timeit(lambda: hip.bn_bwd_apply(hip.BF16, d2, dout, out, y, dst, dy, None, dg, db))
        e = M * C * 2
        print(f"M={M:8d} C={C:5d} x{n:2d}: apply(+res) {t1:7.1f} us {3 * e / t1 / 1e3:6.0f} GB/s | bwd_reduce {t2:7.1f} us {3 * e / t2 / 1e3:6.0f} GB/s | bwd_apply {t3:7.1f} us {4 * e / t3 / 1e3:6.0f} GB/s")
        print(f"                       with packed relu bits: apply {t1b:7.1f} us | bwd_reduce {t2b:7.1f} us | bwd_apply {t3b:7.1f} us")
        tot[0] += n * t1; tot[1] += n * t2; tot[2] += n * t3
    print(f"weighted totals per step: apply {tot[0] / 1e3:.2f} ms, bwd_reduce {tot[1] / 1e3:.2f} ms, bwd_apply {tot[2] / 1e3:.2f} ms")
