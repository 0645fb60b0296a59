"""Global gradient norm of the benchmarked step over its first steps (is clip_grad_norm_(10) active? — decides whether a speculative
un-clipped update of the text encoder ahead of the norm can pay). Usage: python tools/probe_gradnorm.py [--steps 12]"""
import argparse
import contextlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--batch", type=int, default=128)
    a = ap.parse_args()
    args = argparse.Namespace(batch=a.batch, visual="resnet50", layers=12, f32=False, loss="jsd", fp8=False)
    from clip_lite_amd.train_loop import TrainStep
    from clip_lite_amd.utils.common import GradScaler
    device = torch.device("cuda", 0)
    with contextlib.redirect_stdout(sys.stderr):
        model, opt, sched = bench.build(args, device)
    step = TrainStep(model, opt, sched, GradScaler(True), 10.0, None)
    batches = bench.synthetic_batches(args, device, 0)
    A = model.runtime.arena
    regs = {k: A.region(k + ".") for k in ("text_encoder", "image_encoder", "loss")}
    for i in range(a.steps):
        out = step(batches[i % 2])
        torch.cuda.synchronize()
        # the update kernel zeroes the gradients: the norm it used is what clip_grad_norm left in opt.sumsq
        print(f"step {i}: loss {out['loss'].item():.4f}  global grad norm {opt.optimizer.sumsq.sqrt().item():.3f}")


if __name__ == "__main__":
    main()
