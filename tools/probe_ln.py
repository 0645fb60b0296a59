"""Timing probe: LayerNorm forward / backward at the BERT shape of the step (M = 3840, C = 768), by feature: with / without the dgamma,dbeta
atomics, with / without the recomputed dropout masks. Usage: python tools/probe_ln.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clip_lite_amd import hip
from probe_bert import timeit

if __name__ == "__main__":
    M, C = 3840, 768
    x = torch.randn(M, C, device="cuda").bfloat16()
    dy = torch.randn(M, C, device="cuda").bfloat16()
    out, dx, dxm = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    st = torch.zeros(M, 2, device="cuda")
    hip.layernorm_fwd(hip.BF16, x, g, b, 1e-12, out, st, M, C)
    print(f"fwd                         {timeit(lambda: hip.layernorm_fwd(hip.BF16, x, g, b, 1e-12, out, st, M, C)) * 1e3:7.1f} us")
    for name, kw in [("bwd plain, no dgamma", dict(dgamma=None, dbeta=None)),
                     ("bwd plain", dict(dgamma=dg, dbeta=db)),
                     ("bwd drop_in", dict(dgamma=dg, dbeta=db, drop_in=(0.1, 1234, 7))),
                     ("bwd drop_in + masked out", dict(dgamma=dg, dbeta=db, drop_in=(0.1, 1234, 7), drop_out=(0.1, 1234, 8), dxm=dxm))]:
        dgamma, dbeta = kw.get("dgamma"), kw.get("dbeta")
        f = lambda: hip.layernorm_bwd(hip.BF16, dy, x, st, g, dx, kw.get("dxm"), dgamma, dbeta, M, C, drop_in=kw.get("drop_in", hip.NO_DROP),
                                      drop_out=kw.get("drop_out", hip.NO_DROP))
        print(f"{name:27s} {timeit(f) * 1e3:7.1f} us")
