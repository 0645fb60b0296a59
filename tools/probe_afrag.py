"""BatchNorm + ReLU applied on the A FRAGMENT of the 1 x 1 convolution that consumes it (bn2 -> conv3 of a Bottleneck, reference model_zoo/resnet.py:60-100),
against the two launches the step runs today (clite_bn_apply writing a, then clite_conv_fwd reading it), at the four conv3 shapes of ResNet-50, batch 128.
Probe build only:   make variant VAR_EXTRA=-DCLITE_PROBE_AFRAG && CLITE_HIP_LIB=build/var/libclite_hip_var.so python tools/probe_afrag.py
Prints, per shape: bn_apply us | conv_fwd on a us | conv_fwd on the raw tensor with the fragment-side apply us | max error of the fused output against
the two-launch one (both bf16) relative to its max."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clip_lite_amd import hip
from probe_fwd1x1 import timed

if __name__ == "__main__":
    L = hip.lib()
    L.clite_probe_set_affine.argtypes = [C.c_void_p, C.c_void_p]
    L.clite_probe_set_affine.restype = None
    tot = [0.0, 0.0, 0.0]
    for (H, Cc, K, n) in [(56, 64, 256, 3), (28, 128, 512, 4), (14, 256, 1024, 6), (7, 512, 2048, 3)]:
        N = 128
        M = N * H * H
        cv = hip.conv_desc(hip.BF16, N, H, H, Cc, K, 1, 1, 1, 0)
        z = (torch.randn(M, Cc, device="cuda") * 1.5 + 0.2).bfloat16()
        w = (torch.randn(K, Cc, device="cuda") * 0.05).bfloat16()
        g, b = torch.rand(Cc, device="cuda") + 0.5, torch.randn(Cc, device="cuda") * 0.2
        rm, rv = torch.zeros(Cc, device="cuda"), torch.ones(Cc, device="cuda")
        st = hip.Stats(torch.zeros(8 * 3 * Cc, device="cuda"), 8, Cc)
        st.t[:Cc] = z.float().sum(0)
        st.t[Cc:2 * Cc] = (z.float() ** 2).sum(0)
        a = torch.empty_like(z)
        bits = torch.empty(M, Cc // 8, device="cuda", dtype=torch.uint8)
        d = hip.bn_desc(M, Cc, st, g, b, rm, rv, True, False, 0.1, 1e-5, True, relu_bits=bits)
        y1 = torch.empty(M, K, device="cuda", dtype=torch.bfloat16)
        y2 = torch.empty_like(y1)
        so = hip.Stats(torch.zeros(8 * 3 * K, device="cuda"), 8, K)
        L.clite_probe_set_affine(None, None)
        t_bn = timed(lambda: hip.bn_apply(hip.BF16, d, z, None, a))
        t_cv = timed(lambda: hip.conv_fwd(a, w, cv, hip.epilogue(y1, K, colsum=so)))
        mean = z.float().mean(0)
        var = z.float().var(0, unbiased=False)
        scale = (g * torch.rsqrt(var + 1e-5)).contiguous()
        shift = (b - mean * scale).contiguous()
        L.clite_probe_set_affine(scale.data_ptr(), shift.data_ptr())
        t_fu = timed(lambda: hip.conv_fwd(z, w, cv, hip.epilogue(y2, K, colsum=so)))
        L.clite_probe_set_affine(None, None)
        torch.cuda.synchronize()
        err = ((y2.float() - y1.float()).abs().max() / y1.float().abs().max()).item()
        print(f"conv3 {Cc:4d}->{K:4d} @{H:3d} x{n}: bn_apply {t_bn:6.1f} us | conv on a {t_cv:6.1f} us | fused {t_fu:6.1f} us  ({t_bn + t_cv - t_fu:+6.1f} us per block)   err {err:.1e}")
        tot[0] += n * t_bn; tot[1] += n * t_cv; tot[2] += n * t_fu
    print(f"per step (16 blocks): bn_apply {tot[0]:.0f} us + conv {tot[1]:.0f} us = {tot[0] + tot[1]:.0f} us  ->  fused {tot[2]:.0f} us")
