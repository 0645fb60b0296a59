"""Host-side profile (cProfile) of the eager train step — where the Python launch path spends its time:  python tools/profile_eager.py [--steps 5]"""
import argparse
import contextlib
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--visual", default="resnet50")
    ap.add_argument("--layers", type=int, default=12)
    ap.add_argument("--f32", action="store_true")
    ap.add_argument("--loss", default="jsd")
    args = ap.parse_args()
    from clip_lite_amd.train_loop import TrainStep
    from clip_lite_amd.utils.common import GradScaler
    device = torch.device("cuda", 0)
    with contextlib.redirect_stdout(sys.stderr):
        model, opt, sched = bench.build(args, device)
    step = TrainStep(model, opt, sched, GradScaler(True), 10.0, None, graph=False)
    batches = bench.synthetic_batches(args, device, 0)
    for i in range(3):
        step(batches[i % 2])
    torch.cuda.synchronize()
    # backward runs on the autograd engine's thread, which cProfile (per thread) does not see: wrap the backward executors with a profiler of their own
    import clip_lite_amd.encoder as E
    import clip_lite_amd.loss as Lm
    bpr = cProfile.Profile()

    def wrap(mod, name):
        fn = getattr(mod, name)

        def inner(*a, **k):
            bpr.enable()
            try:
                return fn(*a, **k)
            finally:
                bpr.disable()
        setattr(mod, name, inner)
    for mod, name in ((E, "resnet_backward"), (E, "bert_backward"), (Lm, "jsd_backward")):
        if hasattr(mod, name):
            wrap(mod, name)
    pr = cProfile.Profile()
    pr.enable()
    for i in range(args.steps):
        step(batches[i % 2])
    pr.disable()
    torch.cuda.synchronize()
    print("==== main thread")
    pstats.Stats(pr).sort_stats("tottime").print_stats(12)
    print("==== backward executors (autograd thread)")
    pstats.Stats(bpr).sort_stats("tottime").print_stats(25)


if __name__ == "__main__":
    main()
