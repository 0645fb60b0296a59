import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from clip_lite_amd import hip
BF16 = hip.BF16
M, K, Cin = 401408, 256, 64
R = 4
g = torch.Generator(device="cuda").manual_seed(0)
W = (torch.randn(K, Cin, device="cuda", generator=g) * 0.05).bfloat16()
Wt = W.t().contiguous()
pair = torch.randn(2, M, K, device="cuda", generator=g).bfloat16()
src = torch.randn(M, K, device="cuda", generator=g).bfloat16()
gamma = torch.ones(K, device="cuda"); zeros = torch.zeros(K, device="cuda")
st = hip.Stats(torch.zeros(R, 3, K, device="cuda"), R, K); st.t[:, 1] = M / R
pre = hip.Stats(torch.randn(R, 3, K, device="cuda", generator=g), R, K)
y2 = torch.randn(M, Cin, device="cuda", generator=g).bfloat16()
st2 = hip.Stats(torch.zeros(R, 3, Cin, device="cuda"), R, Cin)
bits = torch.randint(0, 255, (M, Cin // 8), device="cuda", dtype=torch.uint8, generator=g)
desc = hip.bn_desc(M, K, st, gamma, zeros, zeros, gamma, True, False, 0.1, 1e-5, False)
dz2 = torch.empty(M, Cin, device="cuda", dtype=torch.bfloat16)
d2 = hip.Stats(torch.zeros(R, 3, Cin, device="cuda"), R, Cin)
f = hip.bn_fold_prepare(desc, pre, Wt, Cin, zeros, zeros)
mk = lambda: hip.epilogue(dz2, Cin, relu_bits=bits, colsum=d2, bn=(y2, st2, M), bias=f.bias)
flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
def t(pre_fn, n=15):
    ts = []
    for _ in range(n):
        pre_fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); hip.conv_dgrad_bnfold(pair, f.w2, M, K, Cin, mk()); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return np.median(ts), min(ts)
print("after flush.zero_()          ", t(lambda: flush.zero_()))
print("after nothing (warm)         ", t(lambda: None))
print("after pair[0].copy_(src)     ", t(lambda: pair[0].copy_(src)))
print("after flush + copy           ", t(lambda: (flush.zero_(), pair[0].copy_(src))))
print("after flush + sync + idle    ", t(lambda: (flush.zero_(), torch.cuda.synchronize(), torch.cuda._sleep(2000000))))
pairs = [torch.randn(2, M, K, device="cuda", generator=g).bfloat16() for _ in range(3)]
def cyc():
    ts=[]
    for i in range(15):
        p = pairs[i % 3]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); hip.conv_dgrad_bnfold(p, f.w2, M, K, Cin, mk()); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1)*1e3)
    return np.median(ts), min(ts)
print("cycling 3 pair buffers (1.2 GB)", cyc())
