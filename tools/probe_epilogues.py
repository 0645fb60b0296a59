"""What the fused epilogue forms of the BERT GEMMs cost on top of a plain store, per tile policy (0 = the shape rule, 1 W128, 2 W256x128, 3 W256,
4 narrow): python tools/probe_epilogues.py   (bf16, MI355X; times in us, median of 20 launches after warm-up)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clip_lite_amd import hip

SHAPES = [("nt", 3840, 3072, 768, ("", "bias", "gelu")), ("nt", 3840, 768, 3072, ("", "bias", "bdr")), ("nt", 3840, 768, 768, ("", "bias", "bdr")),
          ("nt", 3840, 2304, 768, ("", "bias")),
          # input gradients: on the transposed weight copies they are gemm_nt launches too (round 3); the nn rows = the strided-weight form
          ("nt", 3840, 3072, 768, ("dgelu", "dgelu+b")), ("nt", 3840, 768, 3072, ("res",)), ("nt", 3840, 768, 2304, ("res",)), ("nt", 3840, 768, 768, ("res",)),
          ("nn", 3840, 3072, 768, ("", "dgelu")), ("nn", 3840, 768, 3072, ("", "res"))]


def timed(fn, n=20):
    for _ in range(5):
        fn()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


def main():
    print(f"{'launch':28s} {'form':6s} " + " ".join(f"{'p' + str(p):>7s}" for p in range(5)))
    for k, M, N, K, forms in SHAPES:
        A = torch.randn(M, K, device="cuda").bfloat16()
        B = (torch.randn(N, K, device="cuda") if k == "nt" else torch.randn(K, N, device="cuda")).bfloat16()
        out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
        bias = torch.randn(N, device="cuda")
        pre = torch.randn(M, N, device="cuda").bfloat16()
        res = torch.randn(M, N, device="cuda").bfloat16()
        f = getattr(hip, "gemm_" + k)
        for form in forms:
            ep = {"": lambda: hip.epilogue(out, N), "bias": lambda: hip.epilogue(out, N, bias=bias),
                  "gelu": lambda: hip.epilogue(out, N, bias=bias, act=hip.ACT_GELU, preact=pre),
                  "bdr": lambda: hip.epilogue(out, N, bias=bias, drop=(0.1, 1234, 7), residual=res),
                  "dgelu": lambda: hip.epilogue(out, N, dact_aux=pre, dact=hip.DACT_GELU),
                  "dgelu+b": lambda: hip.epilogue(out, N, dact_aux=pre, dact=hip.DACT_GELU, colsum=torch.zeros(N, device="cuda"), colsum_rows=1),
                  "res": lambda: hip.epilogue(out, N, residual=res)}[form]()
            row = []
            for pol in range(5):
                hip.set_tile_policy(pol)
                row.append(timed(lambda: f(hip.BF16, A, B, M, N, K, ep)))
            hip.set_tile_policy(0)
            print(f"gemm_{k} {M}x{N}x{K:<12d} {form or 'plain':6s} " + " ".join(f"{t:7.1f}" for t in row))


if __name__ == "__main__":
    main()
