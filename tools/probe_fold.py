"""Stand-alone timing of the folded BatchNorm backward against the unfolded pair of launches, on conv3 -> bn3 of the four ResNet-50 stages at batch 128:
old = clite_bn_bwd_apply + clite_conv_dgrad_wt (BatchNorm-backward epilogue), new = clite_bn_fold_prepare + clite_conv_dgrad_bnfold; and the weight-gradient
side's extra launches. python tools/probe_fold.py [policy]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from clip_lite_amd import hip  # noqa: E402

BF16 = hip.BF16


def timed(fn, n=20):
    flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        flush.zero_()          # cold L2 / Infinity Cache, as inside the step
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts))


def main():
    if len(sys.argv) > 1:
        hip.set_tile_policy(int(sys.argv[1]))
    g = torch.Generator(device="cuda").manual_seed(0)
    for M, K, Cin in [(401408, 256, 64), (100352, 512, 128), (25088, 1024, 256), (6272, 2048, 512)]:
        R = 4 if M > 262144 else 2 if M > 65536 else 1
        a = torch.relu(torch.randn(M, Cin, device="cuda", generator=g)).bfloat16()
        W = (torch.randn(K, Cin, device="cuda", generator=g) * 0.05).bfloat16()
        Wt = W.t().contiguous()
        pair = torch.randn(2, M, K, device="cuda", generator=g).bfloat16()
        dz, y = pair[0], pair[1]
        gamma = torch.ones(K, device="cuda")
        zeros = torch.zeros(K, device="cuda")
        st = hip.Stats(torch.zeros(R, 3, K, device="cuda"), R, K)
        st.t[:, 1] = M / R
        pre = hip.Stats(torch.randn(R, 3, K, device="cuda", generator=g), R, K)
        y2 = torch.randn(M, Cin, device="cuda", generator=g).bfloat16()
        st2 = hip.Stats(torch.zeros(R, 3, Cin, device="cuda"), R, Cin)
        bits = torch.randint(0, 255, (M, Cin // 8), device="cuda", dtype=torch.uint8, generator=g)
        desc = hip.bn_desc(M, K, st, gamma, zeros, zeros, gamma, True, False, 0.1, 1e-5, False)
        dy = torch.empty(M, K, device="cuda", dtype=torch.bfloat16)
        dz2 = torch.empty(M, Cin, device="cuda", dtype=torch.bfloat16)
        d2 = hip.Stats(torch.zeros(R, 3, Cin, device="cuda"), R, Cin)
        cv = hip.conv_desc(BF16, 1, 1, M, Cin, K, 1, 1, 1, 0)
        mk = lambda bias=None: hip.epilogue(dz2, Cin, relu_bits=bits, colsum=d2, bn=(y2, st2, M), bias=bias)
        t_apply = timed(lambda: hip.bn_bwd_apply(BF16, desc, dz, None, y, pre, dy, None, zeros, zeros))
        t_dgrad = timed(lambda: hip.conv_dgrad(dy, Wt, cv, mk(), wt=True))
        f = hip.bn_fold_prepare(desc, pre, Wt, Cin, zeros, zeros)
        t_prep = timed(lambda: hip.bn_fold_prepare(desc, pre, Wt, Cin, zeros, zeros))
        t_fold = timed(lambda: hip.conv_dgrad_bnfold(pair, f.w2, M, K, Cin, mk(f.bias)))
        G = torch.zeros(Cin, Cin, device="cuda")
        dw = torch.zeros(K, Cin, device="cuda")
        asum = hip.Stats(torch.zeros(R, 3, Cin, device="cuda"), R, Cin)
        t_cs = 0.0
        t_fin = timed(lambda: hip.bn_fold_wgrad_finish(G, asum, f.coef, Wt, M, K, Cin, dw))
        t_gemm = 0.0

        def gram():
            grp = hip.WgradGroup(BF16)
            grp.conv(a, a, hip.conv_desc(BF16, 1, 1, M, Cin, Cin, 1, 1, 1, 0), G)
            grp.launch()
        t_gram = timed(gram)
        mb = M * K * 2 / 1e6
        print(f"M {M:6d} K {K:4d} Cin {Cin:3d} ({mb:5.0f} MB): apply {t_apply:6.1f} + dgrad {t_dgrad:6.1f} = {t_apply + t_dgrad:6.1f} us | prepare {t_prep:5.1f} + fold-dgrad {t_fold:6.1f} = "
              f"{t_prep + t_fold:6.1f} us | wgrad side: colsum {t_cs:5.1f} finish {t_fin:5.1f} gemm {t_gemm:5.1f} gram(alone) {t_gram:6.1f}")


if __name__ == "__main__":
    main()
