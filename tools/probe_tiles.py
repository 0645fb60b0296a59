"""Tile-policy A/B: the step's representative GEMM / conv launches timed under every tile policy of include/clite.h (clite_set_tile_policy:
1 = 128x128 wide, 2 = 256x128, 3 = 256x256, 4 = round-1 4-wave 128x128x32 kernels), interleaved in ONE process (rule 24 of the CDNA guide),
with the result of each policy checked against policy 4. Output feeds the automatic rule in csrc/gemm_wide.hip.
Usage (GPU box): python tools/probe_tiles.py [rounds] > gpurun_out/tiles.txt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from clip_lite_amd import hip  # noqa: E402

POLICIES = (4, 1, 2, 3)
B = 128


def bf(*shape):
    return torch.randn(*shape, device="cuda").bfloat16()


def gemm_case(kind, M, N, K):
    A = bf(M, K) if kind != "tn" else bf(K, M)
    Bm = bf(N, K) if kind == "nt" else bf(K, N)
    out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if kind == "tn" else torch.bfloat16)
    f = getattr(hip, "gemm_" + kind)

    def run():
        if kind == "tn":
            out.zero_()
        f(hip.BF16, A, Bm, M, N, K, hip.epilogue(out, N, atomic=(kind == "tn")))
        return out
    return f"gemm_{kind} {M}x{N}x{K}", 2.0 * M * N * K, run


def conv_cases(Cin, K, R, stride, H):
    pad = R // 2
    cv = hip.conv_desc(hip.BF16, B, H, H, Cin, K, R, R, stride, pad)
    x, w = bf(B * H * H, Cin), bf(K, R, R, Cin) * 0.05
    y = torch.empty(B * cv.Ho * cv.Wo, K, device="cuda", dtype=torch.bfloat16)
    dy = bf(B * cv.Ho * cv.Wo, K)
    dx = torch.empty(B * H * H, Cin, device="cuda", dtype=torch.bfloat16)
    dw = torch.zeros(K, R, R, Cin, device="cuda")
    st = hip.Stats(torch.zeros(8, 3, K, device="cuda"), 8, K)
    flops = 2.0 * B * cv.Ho * cv.Wo * K * R * R * Cin
    lab = f"{Cin}->{K} {R}x{R}/{stride} @{H}"

    def fwd():
        hip.conv_fwd(x, w, cv, hip.epilogue(y, K, colsum=st))
        return y

    def dgrad():
        hip.conv_dgrad(dy, w, cv, hip.epilogue(dx, Cin))
        return dx

    def wgrad():
        dw.zero_()
        hip.conv_wgrad(dy, x, cv, dw)
        return dw
    return [("conv_fwd   " + lab, flops, fwd), ("conv_dgrad " + lab, flops, dgrad), ("conv_wgrad " + lab, flops, wgrad)]


def main(rounds=5):
    cases = []
    for k, M, N, K in [("nt", 3840, 768, 768), ("nt", 3840, 2304, 768), ("nt", 3840, 3072, 768), ("nt", 3840, 768, 3072),
                       ("nn", 3840, 768, 768), ("nn", 3840, 768, 2304), ("nn", 3840, 768, 3072), ("nn", 3840, 3072, 768),
                       ("tn", 768, 768, 3840), ("tn", 2304, 768, 3840), ("tn", 3072, 768, 3840), ("tn", 768, 3072, 3840), ("nt", 8192, 8192, 8192)]:
        cases.append(gemm_case(k, M, N, K))
    for c in [(64, 64, 3, 1, 56), (64, 256, 1, 1, 56), (256, 64, 1, 1, 56), (256, 128, 1, 1, 56), (128, 128, 3, 1, 28), (128, 512, 1, 1, 28), (512, 128, 1, 1, 28),
              (256, 256, 3, 1, 14), (256, 1024, 1, 1, 14), (1024, 256, 1, 1, 14), (512, 512, 3, 1, 7), (512, 2048, 1, 1, 7), (2048, 512, 1, 1, 7),
              (128, 128, 3, 2, 56), (256, 512, 1, 2, 56)]:
        cases.extend(conv_cases(*c))
    ev = lambda: torch.cuda.Event(enable_timing=True)
    print(f"{'launch':34s} " + " ".join(f"{'p' + str(p) + ' us':>9s}" for p in POLICIES) + "   best   TF/s(best)  maxdiff vs p4")
    for name, flops, run in cases:
        ref, times, diff = None, {p: [] for p in POLICIES}, 0.0
        for p in POLICIES:
            hip.set_tile_policy(p)
            o = run().float().clone()
            if p == 4:
                ref = o
            else:
                diff = max(diff, ((o - ref).abs().max() / ref.abs().max().clamp_min(1e-6)).item())
        for _ in range(rounds):
            for p in POLICIES:
                hip.set_tile_policy(p)
                run()
                e0, e1 = ev(), ev()
                e0.record()
                for _ in range(4):
                    run()
                e1.record()
                torch.cuda.synchronize()
                times[p].append(e0.elapsed_time(e1) / 4 * 1e3)
        med = {p: sorted(t)[len(t) // 2] for p, t in times.items()}
        best = min(med, key=med.get)
        print(f"{name:34s} " + " ".join(f"{med[p]:9.1f}" for p in POLICIES) + f"   p{best}   {flops / med[best] / 1e6:9.1f}   {diff:.2e}", flush=True)
    hip.set_tile_policy(0)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
