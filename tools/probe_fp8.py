"""fp8 (e4m3) forward GEMM / conv against the bf16 launch of the same shape, operands PRE-quantised (the quantiser's cost excluded): what a
producer-fused quantiser could gain at best; the fp8 launch under tile policies 0 (shape rule) / 1 (128 x 128) / 2 (256 x 128).
python tools/probe_fp8.py [batch]   (CLITE_HIP_LIB=build/varf8/libclite_hip_var.so for the non-scaled v_mfma_f32_32x32x16_fp8_fp8 build)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clip_lite_amd import hip


BATCH = int(sys.argv[1]) if len(sys.argv) > 1 else 128


def pols(fn):
    ts = []
    for pol in (0, 1, 2):
        hip.set_tile_policy(pol)
        ts.append(timeit(fn))
    hip.set_tile_policy(0)
    return " / ".join(f"{t:6.1f}" for t in ts)


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


for M, N, K in [(3840, 3072, 768), (3840, 768, 3072), (3840, 2304, 768), (3840, 768, 768), (7680, 3072, 768), (7680, 768, 3072)]:
    A, B = torch.randn(M, K, device="cuda").bfloat16(), torch.randn(N, K, device="cuda").bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    A8, B8 = hip.Fp8Tensor(A, hip.BF16), hip.Fp8Tensor(B, hip.BF16)
    t0 = timeit(lambda: hip.gemm_nt(hip.BF16, A, B, M, N, K, hip.epilogue(out, N)))
    t1 = pols(lambda: hip.gemm_nt_fp8(A8, B8, M, N, K, hip.epilogue(out, N)))
    tq = timeit(lambda: hip.Fp8Tensor(A, hip.BF16))
    print(f"gemm_nt {M}x{N}x{K}: bf16 {t0:6.1f} us  fp8 {t1} us  (quantising A separately: {tq:5.1f} us)")
for (N, H, W, Cc, K, R, st, pad) in [(BATCH, 56, 56, 64, 64, 3, 1, 1), (BATCH, 56, 56, 64, 256, 1, 1, 0), (BATCH, 56, 56, 256, 64, 1, 1, 0), (BATCH, 28, 28, 128, 128, 3, 1, 1),
                                      (BATCH, 28, 28, 512, 128, 1, 1, 0), (BATCH, 14, 14, 256, 256, 3, 1, 1), (BATCH, 14, 14, 1024, 256, 1, 1, 0), (BATCH, 14, 14, 256, 1024, 1, 1, 0),
                                      (BATCH, 7, 7, 512, 512, 3, 1, 1), (BATCH, 7, 7, 2048, 512, 1, 1, 0), (BATCH, 7, 7, 512, 2048, 1, 1, 0)]:
    cv = hip.conv_desc(hip.BF16, N, H, W, Cc, K, R, R, st, pad)
    x, w = torch.randn(N, H, W, Cc, device="cuda").bfloat16(), torch.randn(K, R, R, Cc, device="cuda").bfloat16()
    y = torch.empty(N * cv.Ho * cv.Wo, K, device="cuda", dtype=torch.bfloat16)
    st_ = hip.Stats(torch.zeros(8 * 3 * K, device="cuda"), 8, K)
    x8, w8 = hip.Fp8Tensor(x, hip.BF16), hip.Fp8Tensor(w, hip.BF16)
    t0 = timeit(lambda: hip.conv_fwd(x, w, cv, hip.epilogue(y, K, colsum=st_)))
    t1 = pols(lambda: hip.conv_fwd_fp8(x8, w8, cv, hip.epilogue(y, K, colsum=st_)))
    tq = timeit(lambda: hip.Fp8Tensor(x, hip.BF16))
    print(f"conv_fwd {Cc:4d}->{K:4d} {R}x{R} @{H:3d}: bf16 {t0:6.1f} us  fp8 {t1} us  (quantising x separately: {tq:5.1f} us)")
