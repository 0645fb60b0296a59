"""When does each span of the gradient arena reach the exchange in the captured data-parallel step? One rank (the exchange object exists, its
collectives are no-ops at world size 1), the benchmark's model and batch: prints, for one replayed step, every span handed to
GradientExchange.reduce_span with its size and the time (ms from the step's first event) at which its gradients were final on the stream
that produced them. VERDICT r4 next 4: >= 60 % of the 625 MB before 9 ms.      python tools/timeline_handover.py [step.text_segments=1]"""
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    args = types.SimpleNamespace(visual="resnet50", layers=12, f32=False, loss="jsd", batch=128, fp8=False)
    from clip_lite_amd.train_loop import TrainStep
    from clip_lite_amd.utils import distributed as cdist
    from clip_lite_amd.utils.common import GradScaler
    device = torch.device("cuda", 0)
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):
        model, opt, sched = bench.build(args, device)
    ex = cdist.GradientExchange(model.runtime.arena)
    model.runtime.exchange = ex
    step = TrainStep(model, opt, sched, GradScaler(True), 10.0, ex, graph=True)
    for kv in sys.argv[1:]:
        k, v = kv.split("=")
        if k.startswith("step."):
            setattr(step, k[5:], int(v))
        else:
            setattr(model.runtime, k, type(getattr(model.runtime, k))(int(v)))
    step.track_handover = True
    batches = bench.synthetic_batches(args, device, 0)
    for i in range(8):
        step(batches[i % 2])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(20):
        step(batches[i % 2])
    e1.record()
    torch.cuda.synchronize()
    total = model.runtime.arena.total
    print(f"captured data-parallel step, one rank, text_segments = {len(step._tsegs)}: {e0.elapsed_time(e1) / 20:.3f} ms per step")
    acc = 0
    for (lo, hi), ev in sorted(step.handover, key=lambda x: step._t0.elapsed_time(x[1])):
        acc += hi - lo
        print(f"  {step._t0.elapsed_time(ev):7.3f} ms  span [{lo:>10d}, {hi:>10d})  {(hi - lo) * 4 / 1e6:7.1f} MB   cumulative {acc * 4 / 1e6:7.1f} MB = {100.0 * acc / total:5.1f} %")


if __name__ == "__main__":
    main()
