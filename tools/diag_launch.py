"""Host-side cost of replaying the captured step: wall time of each hipGraph launch call (the host enqueues every node's packet inside the call),
host time per step and GPU time per step. If the host time per step approaches the GPU time, the order in which the graphs are launched decides
which stream starves. Usage (GPU box): python tools/diag_launch.py"""
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    args = types.SimpleNamespace(visual="resnet50", layers=12, f32=False, loss="jsd", batch=128, fp8=False)
    from clip_lite_amd.train_loop import TrainStep
    from clip_lite_amd.utils.common import GradScaler
    device = torch.device("cuda", 0)
    model, opt, sched = bench.build(args, device)
    for kv in sys.argv[1:]:          # runtime switches, as tools/ab_runtime.py: attr=value on the DeviceRuntime
        k_, v_ = kv.split("=")
        old = getattr(model.runtime, k_)
        setattr(model.runtime, k_, type(old)(int(v_)))
    step = TrainStep(model, opt, sched, GradScaler(True), 10.0, None, graph=True)
    batches = bench.synthetic_batches(args, device, 0)
    for i in range(6):
        step(batches[i % 2])
    torch.cuda.synchronize()
    G = step._graphs
    times = {k: [] for k in G}
    marks = []
    orig = {}
    for k, g in G.items():
        orig[k] = g.replay

        def timed(k=k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()                      # on the stream the graph is launched on (the current one)
            t0 = time.perf_counter()
            orig[k]()
            times[k].append((time.perf_counter() - t0) * 1e3)
            e1.record()
            marks.append((k, e0, e1))
        g.replay = timed
    host = []
    torch.cuda.synchronize()
    t_all = time.perf_counter()
    starts = []
    for i in range(10):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        starts.append(e)
        t0 = time.perf_counter()
        step(batches[i % 2])
        host.append((time.perf_counter() - t0) * 1e3)
    torch.cuda.synchronize()
    total = (time.perf_counter() - t_all) * 1e3 / 10
    print(f"GPU-paced step {total:.2f} ms; host time inside step(): median {sorted(host)[5]:.2f} ms (min {min(host):.2f}, max {max(host):.2f})")
    for k, v in times.items():
        print(f"  {k:14s} host launch {sorted(v)[len(v) // 2]:7.3f} ms")
    # GPU-side timeline of the 8th step: when each graph's first / last node ran, relative to the step's first event
    n = len(G)
    print("GPU-side timeline of one replayed step (ms from its start; events on the launch streams, no profiler):")
    for k, e0, e1 in marks[7 * n:8 * n]:
        print(f"  {k:14s} {starts[7].elapsed_time(e0):7.3f} -> {starts[7].elapsed_time(e1):7.3f}")


if __name__ == "__main__":
    main()
