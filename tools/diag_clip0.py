"""The captured per-phase step with gradient clipping DISABLED (clip_grad_norm = 0: no norm phases are captured): six steps of a small model, prints the
losses and the replay count.   python tools/diag_clip0.py"""
import sys, contextlib, argparse
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from clip_lite_amd.train_loop import TrainStep
from clip_lite_amd.utils.common import GradScaler
args = argparse.Namespace(batch=16, visual="resnet18", layers=2, f32=False, loss="jsd", fp8=False, gpus=1)
dev = torch.device("cuda", 0)
with contextlib.redirect_stdout(sys.stderr):
    model, opt, sched = bench.build(args, dev)
step = TrainStep(model, opt, sched, GradScaler(True), 0.0, None, graph=True)
bs = bench.synthetic_batches(args, dev, 0)
ls = [step(bs[i % 2])["loss"].item() for i in range(6)]
torch.cuda.synchronize()
print("clip disabled, graph:", step.graph, "replays", getattr(step, "replays", None), ["%.4f" % l for l in ls])
