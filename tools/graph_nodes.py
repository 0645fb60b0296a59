"""Node list of the captured train step: every per-phase hipGraph dumped with hipGraphDebugDotPrint (TrainStep.debug_graph_dir) and tallied by node type —
kernels by name, memcpy and memset nodes with their sizes. VERDICT r4 next 8: which nodes of the step are not this library's kernels, and why.
    python tools/graph_nodes.py [attr=value ...]  > profiles/r5_graph_nodes.txt"""
import collections
import os
import re
import sys
import tempfile
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
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):
        model, opt, sched = bench.build(args, device)
    for kv in sys.argv[1:]:
        k_, v_ = kv.split("=")
        old = getattr(model.runtime, k_)
        setattr(model.runtime, k_, type(old)(int(v_)))
    step = TrainStep(model, opt, sched, GradScaler(True), 10.0, None, graph=True)
    d = step.debug_graph_dir = tempfile.mkdtemp()
    batches = bench.synthetic_batches(args, device, 0)
    for i in range(4):
        step(batches[i % 2])
    torch.cuda.synchronize()
    total = collections.Counter()
    for name in step._graphs:
        path = os.path.join(d, name + ".dot")
        if not os.path.exists(path):
            print(f"{name}: no dump at {path}; directory holds {os.listdir(d)[:5]}")
            continue
        txt = open(path).read()
        if os.environ.get("GRAPH_NODES_RAW"):
            print(txt[:1500])
        labels = re.findall(r'label="([^"]*)"', txt)
        kinds = collections.Counter()
        others = []
        for lab in labels:
            flat = lab.replace("\\n", " ").replace("\n", " ")
            up = flat.upper()
            if "MEMCPY" in up:
                kinds["memcpy"] += 1
                others.append(flat[:160])
            elif "MEMSET" in up:
                kinds["memset"] += 1
                others.append(flat[:160])
            elif "KERNEL" in up or "clite" in flat or "kernel" in flat:
                m = re.search(r"([A-Za-z_][\w:<>, ]*kernel\w*)", flat)
                short = (m.group(1) if m else flat)[:70]
                aten = "at::native" in flat or "elementwise" in flat or "vectorized" in flat
                kinds["ATen kernel" if aten else "library kernel"] += 1
                if aten:
                    others.append(flat[:160])
            else:
                kinds["other"] += 1
        total.update(kinds)
        print(f"{name:18s} " + "  ".join(f"{k} {v}" for k, v in sorted(kinds.items())))
        for o, n in collections.Counter(others).most_common():
            print(f"      {n:3d} x {o}")
    print("whole step: " + "  ".join(f"{k} {v}" for k, v in sorted(total.items())))


if __name__ == "__main__":
    main()
