"""Per-launch profile of the implicit-GEMM engine inside one train step: every conv / linear forward, dgrad and wgrad launch with its
GEMM shape, duration (HIP events on the launch stream), TFLOP/s and minimum-HBM-traffic GB/s. Usage (on the GPU box):
    python tools/layer_profile.py [--batch 128] [--visual resnet50] [--layers 12] > gpurun_out/layers.txt"""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--visual", default="resnet50")
    ap.add_argument("--layers", type=int, default=12)
    ap.add_argument("--f32", action="store_true")
    ap.add_argument("--loss", default="jsd", choices=["jsd", "infonce"])
    ap.add_argument("--fp8", action="store_true", help="the image encoder's fp8 forward (clip-lite_amd/fp8.py)")
    ap.add_argument("--back-to-back", action="store_true", help="issue every launch twice and time the second (profiling only: accumulating outputs double)")
    args = ap.parse_args()
    from clip_lite_amd import hip
    from clip_lite_amd.train_loop import TrainStep
    from clip_lite_amd.utils.common import GradScaler
    device = torch.device("cuda", 0)
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):
        model, opt, sched = bench.build(args, device)
    model.runtime.fp8 = bool(args.fp8)
    model.overlap_encoders = False      # per-launch timing: nothing else may share the chip
    step = TrainStep(model, opt, sched, GradScaler(True), 10.0, None)
    batches = bench.synthetic_batches(args, device, 0)
    for i in range(3):
        step(batches[i % 2])
    torch.cuda.synchronize()
    names = bench.IGEMM_ENTRY_POINTS
    lib = hip.lib()
    recs = []

    def wrap(name):
        fn = getattr(lib, name)

        def timed(*a):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if args.back_to_back:       # the same launch once more right in front of the timed one: no idle gap, warm caches (upper bound of what host gaps cost)
                fn(*a)
            e0.record()
            rc = fn(*a)
            e1.record()
            if name == "clite_wgrad_group":      # one launch set, many members: reported as one row (sum of the members' work)
                ms = bench.describe_group(a, 4 if args.f32 else 2)
                d = (f"group of {len(ms)} wgrads", sum(m[1] for m in ms), 1, 1, sum(m[4] for m in ms))
                recs.append((name, d + (sum(2.0 * m[1] * m[2] * m[3] for m in ms),), e0, e1))
            else:
                d = bench.describe_launch(name, a, 4 if args.f32 else 2)
                # + the fused epilogue's own operands (ReLU-mask source, BatchNorm input, residual: one M x N read each; saved pre-activation:
                # one write), which SURVEY §8(d)'s per-launch figure leaves out but which bound the HBM-limited launches
                extra = 0
                for x in a:
                    ep = getattr(x, "_obj", None)
                    if isinstance(ep, hip.Epilogue):
                        extra = sum(1 for f in ("dact_aux", "bn_y", "residual", "preact") if getattr(ep, f)) * d[1] * d[2] * (4 if args.f32 else 2)
                        if ep.relu_bits:
                            extra += d[1] * d[2] // 8
                recs.append((name, d[:4] + (d[4] + extra, None), e0, e1))
            return rc
        return timed
    wrapped = {n: wrap(n) for n in names}

    class Proxy:
        def __getattr__(self, k):
            return wrapped.get(k) or getattr(lib, k)
    hip._lib = Proxy()
    REP = 3
    for i in range(REP):
        step(batches[i % 2])
    torch.cuda.synchronize()
    hip._lib = lib
    agg = collections.OrderedDict()
    for name, d, e0, e1 in recs:
        key = (name, d)
        agg.setdefault(key, []).append(e0.elapsed_time(e1))
    rows = []
    for (name, (lab, M, N, K, nbytes, flops)), ts in agg.items():
        n = len(ts) // REP
        us = sum(ts) / len(ts) * 1e3
        fl = flops if flops is not None else 2.0 * M * N * K
        # "ideal": the larger of the traffic at 5.5 TB/s (what the BatchNorm streaming kernels reach) and the MACs at 1.6 PFLOP/s (65 % of the dense peak)
        ideal = max(nbytes / 5.5e12, fl / 1.6e15) * 1e6
        rows.append((name[6:], lab, M, N, K, n, us, fl / us / 1e6, nbytes / us / 1e3, ideal))
    tot = sum(r[5] * r[6] for r in rows)
    print(f"{'op':12s} {'layer':28s} {'M':>8s} {'N':>6s} {'K':>8s} {'n':>3s} {'us':>8s} {'TF/s':>7s} {'GB/s':>7s} {'%':>5s} {'ideal us':>8s} {'lost us/step':>12s}")
    for r in sorted(rows, key=lambda r: -r[5] * (r[6] - r[9])):
        print(f"{r[0]:12s} {r[1]:28s} {r[2]:8d} {r[3]:6d} {r[4]:8d} {r[5]:3d} {r[6]:8.1f} {r[7]:7.1f} {r[8]:7.0f} {100 * r[5] * r[6] / tot:5.1f} {r[9]:8.1f} {r[5] * (r[6] - r[9]):12.1f}")
    print(f"total igemm time per step: {tot / 1e3:.3f} ms over {sum(r[5] for r in rows)} launches; ideal {sum(r[5] * r[9] for r in rows) / 1e3:.3f} ms")
    by = collections.defaultdict(float)
    for r in rows:
        by[r[0]] += r[5] * r[6]
    for k, v in sorted(by.items(), key=lambda kv: -kv[1]):
        print(f"  {k:12s} {v / 1e3:8.3f} ms")


if __name__ == "__main__":
    main()
