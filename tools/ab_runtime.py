"""A/B timing of executor switches on the captured train step: python tools/ab_runtime.py [attr=value ...] [--steps 50]
`text.attr=value` sets an attribute of the BERT module (dropout probabilities), `hip.path=value` one of clip_lite_amd.hip, `step.attr=value` one of the
TrainStep. Each other `attr=value` sets an attribute of the model's DeviceRuntime (fuse_bn_backward, s2_classes, group_wgrad, overlap_wgrad ...) before the
step is captured; prints ms/step (HIP events around `steps` replays). The baseline is the same command without arguments."""
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
    ap.add_argument("sets", nargs="*")
    ap.add_argument("--steps", type=int, default=50)
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
    rt = model.runtime
    for s in [x for x in args.sets if x.startswith("hip.")]:        # e.g. hip.WgradGroup.wide=0
        from clip_lite_amd import hip
        path, v = s[4:].split("=")
        obj = hip
        *parents, leaf = path.split(".")
        for name in parents:
            obj = getattr(obj, name)
        setattr(obj, leaf, type(getattr(obj, leaf))(int(v)))
    for s in [x for x in args.sets if x.startswith("text.")]:        # e.g. text.hidden_dropout_prob=0 (a sensitivity probe: what the dropout machinery costs)
        k, v = s[5:].split("=")
        setattr(model.text_encoder.strans, k, float(v))
    for s in [x for x in args.sets if not x.startswith("step.") and not x.startswith("hip.") and not x.startswith("text.")]:
        k, v = s.split("=")
        old = getattr(rt, k)
        setattr(rt, k, type(old)(int(v)) if isinstance(old, (bool, int)) else type(old)(v))
    step = TrainStep(model, opt, sched, GradScaler(True), 10.0, None, graph=True)
    for s_ in [x for x in args.sets if x.startswith("step.")]:
        k, v = s_[5:].split("=")
        setattr(step, k, int(v))
    batches = bench.synthetic_batches(args, device, 0)
    for i in range(6):
        step(batches[i % 2])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(args.steps):
        step(batches[i % 2])
    e1.record()
    torch.cuda.synchronize()
    print(f"{' '.join(args.sets) or 'defaults':40s} {e0.elapsed_time(e1) / args.steps:8.3f} ms/step")


if __name__ == "__main__":
    main()
