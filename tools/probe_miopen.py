"""Yardstick only (never on the product path): MIOpen's convolution kernels, through torch.nn.functional.conv2d (bf16, channels_last), on the
ResNet-50 layer shapes at batch 128 — forward, and backward (data + weight gradient together) — next to this engine's launches
(tools/layer_profile.py).  python tools/probe_miopen.py"""
import torch
import torch.nn.functional as F

SHAPES = [  # Cin, Cout, k, stride, H
    (64, 64, 1, 1, 56), (64, 64, 3, 1, 56), (64, 256, 1, 1, 56), (256, 64, 1, 1, 56), (256, 128, 1, 1, 56), (128, 128, 3, 2, 56),
    (128, 128, 3, 1, 28), (128, 512, 1, 1, 28), (512, 128, 1, 1, 28), (256, 512, 1, 2, 56), (512, 256, 1, 1, 28), (256, 256, 3, 2, 28),
    (256, 256, 3, 1, 14), (256, 1024, 1, 1, 14), (1024, 256, 1, 1, 14), (512, 1024, 1, 2, 28), (1024, 512, 1, 1, 14), (512, 512, 3, 2, 14),
    (512, 512, 3, 1, 7), (512, 2048, 1, 1, 7), (2048, 512, 1, 1, 7), (1024, 2048, 1, 2, 14)]


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def main():
    torch.backends.cudnn.benchmark = True
    B = 128
    print(f"{'conv':28s} {'fwd us':>8s} {'TF/s':>7s} {'bwd us':>8s} {'TF/s':>7s}")
    for ci, co, k, st, H in SHAPES:
        x = torch.randn(B, ci, H, H, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        w = torch.randn(co, ci, k, k, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        pad = k // 2
        y = F.conv2d(x, w, None, st, pad)
        dy = torch.randn_like(y)
        fl = 2.0 * y.numel() * ci * k * k
        tf = timed(lambda: F.conv2d(x, w, None, st, pad))

        def bwd():
            x.grad = w.grad = None
            yy = F.conv2d(x, w, None, st, pad)
            yy.backward(dy)
        tb = timed(bwd) - tf
        print(f"{ci:4d}->{co:4d} {k}x{k}/{st} @{H:<3d}          {tf:8.1f} {fl / tf / 1e6:7.1f} {tb:8.1f} {2 * fl / tb / 1e6:7.1f}")


if __name__ == "__main__":
    main()
