"""Headline benchmark: image-caption pairs/sec of the full CLIP-Lite pretraining step (BASELINE.json configs[1]/[2]):
ResNet-50 + BERT-base (12 layers), JSD-MI loss with both priors, per-GPU batch 128, bf16 storage + bf16 MFMA, synthetic
224x224 images and 30-token captions, dropout and prior noise ON; step = zero_grad, forward, backward, gradient mean over
ranks, global-norm clip, SGD(momentum, wd, per-tensor lr) + Lookahead, LR schedule (reference train.py:211-226).

`python bench.py --gpus N --steps K --warmup W`; for N > 1 run under `python -m torch.distributed.run --nproc-per-node N`.
Rank 0 prints ONE JSON line. Weak scaling: per-GPU batch fixed at 128, `value` = N * 128 * K / time (max over ranks).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as tdist  # noqa: E402

FLOP_PER_PAIR = 40.19e9            # 6.699 GMAC fwd x 2 x 3 (fwd + dgrad + wgrad), ResNet-50 + BERT-base L=30 + heads
FLOP_PER_PAIR_BY_VISUAL = {"resnet50": 40.19e9, "resnet101": 62.47e9}      # SURVEY.md §8d (C2 / C5), with BERT-base 12 layers
MFMA_PEAK_TFLOPS = 2500.0          # dense bf16, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def build(args, device):
    from clip_lite_amd.encoder import ImageEncoder, TextEncoder
    from clip_lite_amd.loss import InfoNCELoss, JSDInfoMaxLoss
    from clip_lite_amd.model import VLInfoModel
    from clip_lite_amd.optim import FusedSGD, Lookahead
    from clip_lite_amd.optim.lr_scheduler import LinearWarmupCosineAnnealingLR
    torch.manual_seed(1234)
    ie = ImageEncoder(args.visual)
    te = TextEncoder(mode="train_sbert", num_hidden_layers=args.layers)
    loss_cls = InfoNCELoss if args.loss == "infonce" else JSDInfoMaxLoss
    loss = loss_cls(ie.img_encoder.out_dim, 768, "dot", 0.1, True, True)
    model = VLInfoModel(te, ie, loss, "train_sbert", is_amp=not args.f32).to(device).train()
    groups = []
    for name, p in model.named_parameters():      # reference factories.py:464-482
        lr = 0.2 if "image_encoder" in name else 1e-3
        groups.append({"params": [p], "lr": lr, "weight_decay": 1e-4})
    opt = Lookahead(FusedSGD(groups, momentum=0.9), k=5, alpha=0.5)
    sched = LinearWarmupCosineAnnealingLR(opt, total_steps=250000, warmup_steps=10000)
    return model, opt, sched


def synthetic_batches(args, device, rank, n=2):
    g = torch.Generator(device=device).manual_seed(1234 + rank)
    out = []
    for _ in range(n):
        img = torch.randn(args.batch, 3, 224, 224, device=device, generator=g)
        ids = torch.randint(1000, 30522, (args.batch, 30), device=device, generator=g)
        ids[:, 0], ids[:, -1] = 101, 102
        out.append({"image": img, "input_ids": ids, "attention_mask": torch.ones(args.batch, 30, dtype=torch.long, device=device)})
    return out


CPU_BASELINE_SNIPPET = """
import os, sys, time, json, torch
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
from oracle import ref_model as O
threads = min(os.cpu_count() or 1, {threads})
torch.set_num_threads(threads)
torch.manual_seed(1234)
B = 8
M = O.build_oracle_model({visual!r}, "train_sbert", {layers}).train()
opt = O.build_optimizer(M.named_parameters())
batch = {{"image": torch.randn(B, 3, 224, 224), "input_ids": torch.randint(1000, 30522, (B, 30)), "attention_mask": torch.ones(B, 30, dtype=torch.long)}}
O.train_step(M, opt, batch, 0)
n, t0 = 0, time.time()
while n < 2 or (time.time() - t0 < 15 and n < 16):        # ~10-15 s of CPU work
    O.train_step(M, opt, batch, n + 1); n += 1
dt = time.time() - t0
print(json.dumps({{"value": B * n / dt, "unit": "pairs/s", "cores": threads, "kind": "port",
                  "sample": f"{{n}} fp32 train steps of the same model at batch {{B}} (oracle/ref_model.py, torch CPU, {{threads}} threads)"}}))
"""


def cpu_baseline(args):
    """The oracle's fp32 CPU train step (the restatement of the reference's own step) on a bounded sample of the same workload,
    in a child process with a hard time limit (many-core hosts can be pathologically slow at batch 8 if over-threaded)."""
    import subprocess
    code = CPU_BASELINE_SNIPPET.format(root=ROOT, threads=32, visual=args.visual, layers=args.layers)
    try:
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=240, env={**os.environ, "HIP_VISIBLE_DEVICES": ""})
        return json.loads(out.stdout.strip().splitlines()[-1])
    except Exception as e:      # noqa: BLE001
        return {"value": None, "unit": "pairs/s", "cores": None, "kind": "port", "sample": f"cpu baseline did not finish within 240 s ({type(e).__name__})"}


def kernel_source_hash():
    """sha256 over everything that decides what the kernels do and how they are launched: csrc/, include/clite.h and the Python executors.
    tools/pmc_traffic.py stores it next to the PMC byte counts; bench.py reports those counts only while the hash still matches."""
    import hashlib
    h = hashlib.sha256()
    pkg = os.path.join(ROOT, "clip-lite_amd")
    files = [os.path.join(ROOT, "include", "clite.h")]
    for d, _, fs in sorted(os.walk(pkg)):
        if "__pycache__" in d or os.sep + "lib" in d:
            continue
        files += [os.path.join(d, f) for f in sorted(fs) if f.endswith((".hip", ".h", ".py"))]
    for f in sorted(files):
        h.update(os.path.relpath(f, ROOT).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


TRAFFIC_JSON = os.path.join(ROOT, "profiles", "r5_hbm_traffic.json")
KERNEL_STATS_JSON = os.path.join(ROOT, "profiles", "r5_kernel_stats.json")
# SURVEY.md section 8(d)'s algorithmic HBM bytes of one step at per-GPU batch 128: ~170 MB per image of conv-output traffic (write once, read once with
# BatchNorm / ReLU fused on load, residual re-reads; forward + ~2x backward) + the update's 156.2 M x (3 reads + 2 writes) x 4 B. BERT's activations
# are not in that model (MFMA-bound by its arithmetic intensity).
ALGORITHMIC_BYTES_PER_IMAGE = 170e6
ALGORITHMIC_UPDATE_BYTES = 156.2e6 * 5 * 4


def pmc_traffic(args):
    """HBM-side bytes of the igemm family per step (all its launches) from the PMC passes committed under profiles/ (rocprofv3 --pmc
    FETCH_SIZE and --pmc WRITE_SIZE in separate runs, FETCH_SIZE doubled per MI355X_MICROARCH.md: counters cannot be collected inside a timed
    run). Reported only for the configuration they were taken on AND only while the kernel sources are the ones they were taken with
    (kernel_source_hash); otherwise null — a stale number is worse than none."""
    if args.visual != "resnet50" or args.layers != 12 or args.batch != 128 or args.f32 or args.loss != "jsd" or getattr(args, "fp8", False):
        return None, "no PMC pass for this configuration", None
    try:
        with open(TRAFFIC_JSON) as f:
            d = json.load(f)
    except Exception:      # noqa: BLE001
        return None, "profiles/r5_hbm_traffic.json absent", None
    if d.get("kernel_source_hash") != kernel_source_hash():
        return None, f"PMC passes of {d.get('date')} were taken on other kernel sources (hash {d.get('kernel_source_hash')}): re-run tools/pmc_traffic.py", None
    moved = (d["igemm_hbm_GB_per_step"] + d["bn_kernels_hbm_GB_per_step"] + d["other_kernels_hbm_GB_per_step"]) * 1e9
    return (d["igemm_hbm_GB_per_step"] * 1e9, f"rocprofv3 PMC passes of {d.get('date')}, kernel sources {d['kernel_source_hash']} (profiles/r5_hbm_traffic.json)",
            {"bytes_moved": moved, "by_family": {"igemm": d["igemm_hbm_GB_per_step"] * 1e9, "batchnorm": d["bn_kernels_hbm_GB_per_step"] * 1e9,
                                                  "other": d["other_kernels_hbm_GB_per_step"] * 1e9}})


def in_step_kernel_time(args):
    """The implicit-GEMM family's kernel time per step INSIDE the two-stream captured step, from the rocprofv3 kernel trace committed under profiles/
    (tools/rocpd_stats.py side-car; same hash guard as the PMC figures). roofline.frac is an isolated figure (per-launch events, the second stream
    off: nothing shares the chip with a timed kernel); this one is what the kernels take while the step actually runs."""
    if args.visual != "resnet50" or args.layers != 12 or args.batch != 128 or args.f32 or args.loss != "jsd" or getattr(args, "fp8", False):
        return None
    try:
        with open(KERNEL_STATS_JSON) as f:
            d = json.load(f)
    except Exception:      # noqa: BLE001
        return None
    return d if d.get("kernel_source_hash") == kernel_source_hash() else None


def describe_launch(name, a, esz):
    """(label, M, N, K, algorithmic bytes) of one implicit-GEMM launch from its C-ABI arguments: every operand read once and the output
    written once (weight gradients: + the f32 accumulator read-modify-write) — SURVEY.md §8(d)'s per-launch figure."""
    if name == "clite_conv_dgrad_bnfold":          # (pair, w2, M, K, Cin, ep, stream): the folded BatchNorm backward's input gradient, K-concatenation [dz | y]
        M, K, Cin = a[2], a[3], a[4]
        return f"{Cin:4d}->{K:4d} 1x1/1 bnfold", M, Cin, 2 * K, esz * (2 * M * K + M * Cin + Cin * 2 * K)
    if name.startswith("clite_conv"):
        cv = a[2]._obj
        P, Q = cv.N * cv.Ho * cv.Wo, cv.N * cv.H * cv.W
        kk = cv.R * cv.S * cv.C
        lab = f"{cv.C:4d}->{cv.K:4d} {cv.R}x{cv.S}/{cv.stride} {cv.H:3d}->{cv.Ho:3d}"
        if name == "clite_conv_fwd":
            return lab, P, cv.K, kk, esz * (Q * cv.C + P * cv.K + cv.K * kk)
        if name == "clite_conv_fwd_fp8":           # e4m3 operands (1 byte), bf16 output
            return lab + " fp8", P, cv.K, kk, Q * cv.C + cv.K * kk + esz * P * cv.K
        if name in ("clite_conv_dgrad_s2class", "clite_conv_dgrad_s2class_wt"):     # one input-parity class of a 3x3 / stride-2 dgrad: a quarter of the pixels, its own taps
            ph, pw = a[3], a[4]
            taps = (1 if (ph + 1) & 1 else 2) * (1 if (pw + 1) & 1 else 2)
            return lab + f" class {ph}{pw}", Q // 4, cv.C, taps * cv.K, esz * (Q * cv.C // 4 + P * cv.K + cv.K * taps * cv.C)
        if name in ("clite_conv_dgrad", "clite_conv_dgrad_wt"):
            return lab, Q, cv.C, cv.R * cv.S * cv.K, esz * (Q * cv.C + P * cv.K + cv.K * kk)
        return lab, cv.K, kk, P, esz * (Q * cv.C + P * cv.K) + 4 * cv.K * kk
    if name.startswith("clite_stem"):
        N, Ho, Wo = a[3], a[6], a[7]
        P = N * Ho * Wo
        if name == "clite_stem_fwd":
            return "stem 7x7", P, 64, 224, esz * (N * a[4] * a[5] * 4 + P * 64)
        return "stem 7x7", 64, 224, P, esz * (N * a[4] * a[5] * 4 + P * 64)
    M, N, K = a[4], a[5], a[6]
    if name == "clite_gemm_nt_fp8":
        return "linear fp8", M, N, K, M * K + N * K + esz * M * N
    if name == "clite_gemm_tn":
        return "linear", M, N, K, esz * (K * M + K * N) + 4 * M * N
    return "linear", M, N, K, esz * (M * K + N * K + M * N)


IGEMM_ENTRY_POINTS = ("clite_gemm_nt", "clite_gemm_nn", "clite_gemm_tn", "clite_conv_fwd", "clite_conv_dgrad", "clite_conv_dgrad_s2class", "clite_conv_dgrad_wt",
                      "clite_conv_dgrad_s2class_wt", "clite_conv_wgrad", "clite_conv_dgrad_bnfold",
                      "clite_stem_fwd", "clite_stem_wgrad", "clite_wgrad_group", "clite_gemm_nt_fp8", "clite_conv_fwd_fp8")


def describe_group(a, esz):
    """Members of one clite_wgrad_group call -> [(label, M, N, K, algorithmic bytes)]"""
    out = []
    for it in a[1][:a[2]]:
        if it.kind == 0:
            cv = it.cv
            P, Q, kk = cv.N * cv.Ho * cv.Wo, cv.N * cv.H * cv.W, cv.R * cv.S * cv.C
            out.append((f"{cv.C:4d}->{cv.K:4d} {cv.R}x{cv.S}/{cv.stride} {cv.H:3d}->{cv.Ho:3d}", cv.K, kk, P, esz * (Q * cv.C + P * cv.K) + 4 * cv.K * kk))
        else:
            out.append(("linear", it.M, it.N, it.K, esz * (it.K * it.M + it.K * it.N) + 4 * it.M * it.N))
    return out


def kernel_roofline(step_fn, batches, steps=3, esz=2):
    """Average duration of the dominant kernel family (the implicit-GEMM engine: every conv / linear forward, dgrad and wgrad launch),
    measured with events recorded on the stream the kernels are launched on, with the algorithmic bytes and FLOPs of the same launches
    (describe_launch) summed live. Returns (ms per step, launches per step, algorithmic bytes per step, launched FLOPs per step)."""
    from clip_lite_amd import hip
    lib = hip.lib()
    events, tally = [], [0.0, 0.0]

    def wrap(name):
        fn = getattr(lib, name)

        def timed(*a):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*a)
            e1.record()
            events.append((e0, e1))
            members = describe_group(a, esz) if name == "clite_wgrad_group" else [describe_launch(name, a, esz)]
            for _, M, N, K, nbytes in members:
                tally[0] += nbytes
                tally[1] += 2.0 * M * N * K
            return rc
        return timed

    class Proxy:
        def __getattr__(self, k):
            return wrapped.get(k) or getattr(lib, k)
    wrapped = {n: wrap(n) for n in IGEMM_ENTRY_POINTS}
    hip._lib = Proxy()
    try:
        for i in range(steps):
            step_fn(batches[i % len(batches)])
        torch.cuda.synchronize()
    finally:
        hip._lib = lib
    total_ms = sum(e0.elapsed_time(e1) for e0, e1 in events)
    return total_ms / steps, len(events) // steps, tally[0] / steps, tally[1] / steps


def side_record(base_args, device, steps, **override):
    """One more configuration of the same step, built, captured, timed and freed in turn (N = 1 only, after the headline timing): VERDICT r3
    item 6 — the metric names "bs1024 ... at 1 GPU" and the north star's 1e-4 loss bar is the exact-f32 mode's, so the driver's line carries
    both numbers beside the headline (per-GPU batch 128, bf16). Same TrainStep(graph=True) path as the headline; 3 untimed steps (2 eager +
    the capture), then `steps` timed replays between synchronisations."""
    import contextlib
    import gc
    from clip_lite_amd.train_loop import TrainStep
    from clip_lite_amd.utils.common import GradScaler
    args = argparse.Namespace(**{**vars(base_args), **override})
    from clip_lite_amd import hip
    hip.set_f32_split(bool(getattr(args, "f32_split", False)))
    with contextlib.redirect_stdout(sys.stderr):
        model, opt, sched = build(args, device)
    step = TrainStep(model, opt, sched, GradScaler(True), 10.0, None, graph=True, defer_update=True)
    batches = synthetic_batches(args, device, 0)
    for i in range(3):
        step(batches[i % len(batches)])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        out = step(batches[i % len(batches)])
    step.finish()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    loss = out["loss"].item()
    ms = dt / steps * 1e3
    flop_pair = FLOP_PER_PAIR_BY_VISUAL.get(args.visual) if args.layers == 12 else None
    rec = {"ms_per_step": ms, "value": args.batch * steps / dt, "unit": "pairs/s", "steps": steps, "batch": args.batch,
           "dtype": "f32" if args.f32 else "bf16", "loss": loss, "launch": "hipGraph replay" if step.replays else "eager"}
    if args.f32:
        rec["matrix_products"] = ("split-bf16: f32 storage, three bf16 MFMAs per product (clite_set_f32_split)" if getattr(args, "f32_split", False)
                                  else "exact: v_mfma_f32_32x32x2_f32, k-ordered fmaf chain")
        # the bars this form is held to at full size against the oracle fixture (tests/test_gpu_model.py::test_full_size_config2_f32_*): shared by both
        # forms except ONE — the stem BatchNorm gain's gradient, where the split form's 2^-17-per-product noise arrives amplified through 50 train-mode
        # BatchNorms (observed 0.12 of max|g| against 0.018 for the exact form). The footnote travels with the number.
        rec["parity_bars"] = {"loss": 1e-4, "gradient_norms": 2e-3, "head_level_gradients": 2e-3,
                              "stem_bn1_gradient": 2e-1 if getattr(args, "f32_split", False) else 4e-2}
    hip.set_f32_split(False)
    if flop_pair and not args.f32:
        rec["whole_step_frac"] = flop_pair * args.batch / (ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS
    del step, model, opt, sched, batches, out
    gc.collect()
    torch.cuda.empty_cache()
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=128, help="per-GPU batch")
    ap.add_argument("--visual", default="resnet50")
    ap.add_argument("--layers", type=int, default=12)
    ap.add_argument("--f32", action="store_true", help="exact-f32 parity kernels instead of bf16")
    ap.add_argument("--f32-split", action="store_true", help="with --f32: the split-bf16 form of the f32 matrix products (clite_set_f32_split: f32 storage, "
                    "three bf16 MFMAs per product; held to the same 1e-4 full-size loss bar as the exact form, tests/test_gpu_model.py)")
    ap.add_argument("--loss", default="jsd", choices=["jsd", "infonce"], help="cross-modal term: the reference's JSD estimator or the InfoNCE all-pairs variant (BASELINE config 4)")
    ap.add_argument("--fp8", action="store_true", help="BASELINE configs[4]: the image encoder's L2-bound forward convs (3x3, and 1x1 at <= 14 x 14) and BERT's QKV / FFN1 / FFN2 "
                    "forward linears on OCP e4m3 operands (v_mfma_scale_f32_32x32x64_f8f6f4 at unit block scales, f32 accumulate; activations quantised by their producers "
                    "with delayed scaling, weights per step with current scaling: clip-lite_amd/fp8.py); everything else, and every backward GEMM, stays bf16")
    ap.add_argument("--no-fp8-text", action="store_true", help="with --fp8: leave the text encoder in bf16 (A/B)")
    ap.add_argument("--no-fp8-dgrad", action="store_true", help="with --fp8: leave the image encoder's input gradients in bf16 (A/B)")
    ap.add_argument("--no-fp8-wgrad", action="store_true", help="with --fp8: leave the weight gradients in bf16 (A/B)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side-records", action="store_true", help="skip the two extra single-GPU records of the default run (the whole bs = 1024 step on "
                    "one GPU; the exact-f32 parity mode): profiling runs and A/B loops")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel from Python each step instead of replaying the captured hipGraph of the step")
    ap.add_argument("--no-defer-update", action="store_true", help="keep the whole update at the end of the step. Default (round 4): the text encoder's and "
                    "heads' share of the update runs at the start of the NEXT step on the text encoder's stream, beside the image forward "
                    "(TrainStep defer_update; -0.1 ms per step, same-box A/B; step.finish() completes the last one inside the timed region)")
    ap.add_argument("--backend", default="nccl", help="process-group backend (nccl = RCCL over xGMI); gloo is for single-GPU logic tests")
    ap.add_argument("--single-device", action="store_true", help="logic test only: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--host-input", action="store_true", help="batches start in pinned host memory and cross PCIe every step through the train loop's "
                    "copy-stream prefetcher (utils/common.cycle): the PCIe-inclusive rate quoted in DESIGN.md, never the headline value")
    ap.add_argument("--exchange", default="allreduce", choices=["allreduce", "mesh"], help="gradient exchange: RCCL all-reduce per region (default), or the "
                    "direct mesh form for a fully connected xGMI node: all-to-all, local sum, all-gather (utils/distributed.py)")
    ap.add_argument("--force-exchange", action="store_true", help="run the gradient exchange (process group, RCCL all-reduces between the graphs) even with "
                    "one rank: exercises the data-parallel launch path on a one-GPU box (launch under torch.distributed.run)")
    args = ap.parse_args()

    # stdout carries exactly one line, the JSON result: RCCL prints a version banner to fd 1 when the process group comes up (and the encoders
    # print construction banners), so fd 1 is pointed at stderr for the whole run and the result goes to the original stdout
    sys.stdout.flush()
    result_stream = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus} (WORLD_SIZE={world})")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist_on = world > 1 or args.force_exchange
    rccl_ranks = None
    if dist_on:
        tdist.init_process_group(backend=args.backend, init_method="env://")
        # pre-flight of the collective path: a 1-element SUM all-reduce of ones must come back as the number of ranks the
        # launcher promised — a process group that silently formed with fewer ranks (or a backend that is not reducing) fails here, not in the timing
        probe = torch.ones(1, device=device, dtype=torch.float32)
        tdist.all_reduce(probe, op=tdist.ReduceOp.SUM)
        rccl_ranks = int(round(probe.item()))
        if rccl_ranks != tdist.get_world_size() or tdist.get_world_size() != world:
            raise SystemExit(f"collective pre-flight failed: all-reduce of ones = {rccl_ranks}, process group {tdist.get_world_size()}, WORLD_SIZE {world}")

    from clip_lite_amd import hip as _hip
    _hip.set_f32_split(bool(args.f32 and args.f32_split))
    from clip_lite_amd.train_loop import TrainStep
    from clip_lite_amd.utils import distributed as cdist
    from clip_lite_amd.utils.common import GradScaler
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):      # the encoders print a construction banner like the reference's; keep stdout = the JSON line
        model, opt, sched = build(args, device)
    cdist.broadcast_parameters(model)
    model.runtime.fp8 = bool(args.fp8)
    model.runtime.fp8_text = bool(args.fp8) and not args.no_fp8_text
    model.runtime.fp8_dgrad = bool(args.fp8) and not args.no_fp8_dgrad
    model.runtime.fp8_wgrad = bool(args.fp8) and not getattr(args, "no_fp8_wgrad", False)
    exchange = None
    if dist_on:
        exchange = cdist.GradientExchange(model.runtime.arena, algorithm=args.exchange)
        model.runtime.exchange = exchange
    # defer_update: the step's last kernel (the update of the text encoder and the heads) runs at the start of the next step, beside the image
    # forward; step.finish() below then completes the last timed step's update INSIDE the timed region. Round 3 measured it neutral (17.32 vs
    # 17.35 ms); with round 4's shorter side stream it is worth 0.1 ms (15.14 -> 15.04, same box) and is on by default
    step = TrainStep(model, opt, sched, GradScaler(True), 10.0, exchange, graph=not args.no_graph, defer_update=not args.no_defer_update)
    eager_step = TrainStep(model, opt, sched, GradScaler(True), 10.0, exchange)      # per-launch timing needs eager launches
    batches = synthetic_batches(args, device, rank)
    if args.host_input:
        from clip_lite_amd.utils.common import cycle

        class _HostLoader:          # what a DataLoader(pin_memory=True) hands out
            sampler = None
            data = [{k: v.cpu().pin_memory() for k, v in b.items()} for b in batches]

            def __iter__(self):
                return iter(self.data)
        feed = cycle(_HostLoader(), device)
        next_batch = lambda i: next(feed)
    else:
        next_batch = lambda i: batches[i % len(batches)]

    for i in range(max(args.warmup, 3 if step.graph else 0)):      # graph mode: 2 eager steps, then the capture + first replay
        step(next_batch(i))
    torch.cuda.synchronize()
    if dist_on:
        tdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(next_batch(i))
    step.finish()
    torch.cuda.synchronize()
    if dist_on:
        tdist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
        dt = t.item()
    loss = out["loss"].item()
    replicas_identical = None
    if dist_on:
        # data parallel keeps every rank's parameters bit-identical (same mean gradient, same update). A BITWISE comparison: integer sums of the
        # parameters' 32-bit patterns (wrapping int64 arithmetic is exact and order-independent), whole arena plus two strided sub-samples
        bits = model.runtime.arena.flat_p.view(torch.int32).to(torch.int64)
        mine = torch.stack([bits.sum(), (bits * (1 + torch.arange(bits.numel(), device=bits.device) % 8191)).sum(), bits[::97].sum()])
        every = [torch.empty_like(mine) for _ in range(tdist.get_world_size())]
        tdist.all_gather(every, mine)
        replicas_identical = all(torch.equal(every[0], e) for e in every)
    # every rank runs the instrumented steps (they contain the gradient exchange, a collective); rank 0 reports its timings
    model.overlap_encoders = False          # per-launch durations: nothing else may share the chip with the timed kernel
    gemm_ms, n_launch, alg_bytes, launch_flops = kernel_roofline(eager_step, batches, esz=4 if args.f32 else 2)
    model.overlap_encoders = True

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * args.batch * args.steps / dt
        flop_pair = FLOP_PER_PAIR_BY_VISUAL.get(args.visual) if args.layers == 12 else None
        gemm_tflops = flop_pair * args.batch / (gemm_ms * 1e-3) / 1e12 if flop_pair else None
        traffic, traffic_note, hbm = pmc_traffic(args)
        if hbm is not None:          # the HBM side of the same step: what crosses the memory interface against SURVEY 8(d)'s algorithmic bytes and the 8 TB/s peak
            hbm["bytes_algorithmic"] = ALGORITHMIC_BYTES_PER_IMAGE * args.batch + ALGORITHMIC_UPDATE_BYTES
            hbm["peak_TBps"] = 8.0
            hbm["achieved_TBps"] = hbm["bytes_moved"] / (ms * 1e-3) / 1e12
            hbm["frac_of_peak"] = hbm["achieved_TBps"] / 8.0
            hbm["moved_over_algorithmic"] = hbm["bytes_moved"] / hbm["bytes_algorithmic"]
        kst = in_step_kernel_time(args)
        res = {
            "metric": "image-caption pairs/sec (global batch) — ResNet-50+BERT bs1024, 1/2/4/8 MI355X",
            "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.f32 else ("fp8 (e4m3 operands of the image encoder's 3x3 / late 1x1 forward convs" + ("" if args.no_fp8_text else " and BERT's QKV / FFN forward linears") + ("" if args.no_fp8_dgrad else "; e5m2 x e4m3 input gradients" + ("" if getattr(args, "no_fp8_wgrad", False) else " and weight gradients") + " of the 3x3 / late 1x1 convs") + ") + bf16" if args.fp8 else "bf16"), "data": "synthetic" + (", fed from pinned host memory every step" if args.host_input else ""),
            "config": {"workload": f"{args.visual} + BERT-base({args.layers}L) + JSD-MI heads/priors, per-GPU batch {args.batch}, 224x224 images, "
                                   f"30-token captions, dropout 0.1 + prior noise on, clip 10 + SGD(0.9, wd 1e-4) + Lookahead(5, 0.5)" + ("" if args.loss == "jsd" else ", InfoNCE all-pairs loss"),
                       "global_batch": world * args.batch, "parallelism": f"dp{world}"},
            "loss": loss, "launch": "hipGraph replay" if step.graph else "eager", "replicas_identical": replicas_identical,
            "rccl_ranks": rccl_ranks, "collective_backend": (args.backend if dist_on else None),
            # which numerical bar this line's kernels are held to (VERDICT r2 weak point 3): the 1e-4 loss bar of the north star is the exact-f32
            # mode's; bf16 storage cannot meet it (neither would the reference's fp16 AMP) and is tested at its own bars
            "precision_note": ("exact-f32 kernels: loss within 1e-4 of the fp32 oracle (tests/test_gpu_model.py, fixtures from the reference)" if args.f32 else
                               "bf16 kernels with f32 accumulation: full-size loss within 2e-2 of the fp32 oracle, conditioned-problem gradient cosine >= 0.93 "
                               "(tests/test_gpu_model.py, tests/test_gpu_ops.py); the 1e-4 bar is met by `--f32` (this line's f32_parity_mode record)"),
            "roofline": {"bound": "mfma", "kernel": "clite::igemm_dma_kernel / igemm_wide_kernel family (all conv/linear fwd+dgrad+wgrad launches of one step)",
                         "achieved": gemm_tflops, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": (gemm_tflops / MFMA_PEAK_TFLOPS) if gemm_tflops else None, "traffic": traffic,
                         "launches_per_step": n_launch, "kernel_ms_per_step": gemm_ms,
                         "algorithmic_bytes": alg_bytes, "launched_flops": launch_flops, "traffic_note": traffic_note,
                         "kernel_source_hash": kernel_source_hash(),
                         # the same family's kernel time inside the two-stream step (committed rocprofv3 kernel trace): what bounds the step is bytes, not
                         # the matrix pipe - `hbm` says how many
                         "frac_in_step": (flop_pair * args.batch / (kst["igemm_ms_per_step"] * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS) if (kst and flop_pair) else None,
                         "kernel_ms_per_step_in_step": kst["igemm_ms_per_step"] if kst else None,
                         "hbm": hbm,
                         "whole_step_frac": flop_pair * args.batch / (ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS if gemm_tflops else None},
        }
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(args)
        default_cfg = args.visual == "resnet50" and args.layers == 12 and args.batch == 128 and not args.f32 and args.loss == "jsd" and not args.fp8
        if world == 1 and not dist_on and default_cfg and not args.no_side_records and step.graph and not args.host_input:
            # the headline's model, optimizer and captured graphs are freed first: each record owns the GPU while it is timed
            import gc
            del step, eager_step, model, opt, sched, exchange, batches, out
            gc.collect()
            torch.cuda.empty_cache()
            try:
                res["bs1024_single_gpu"] = side_record(args, device, 10, batch=1024)
            except Exception as e:      # noqa: BLE001  (the headline must survive a failure of an extra record)
                res["bs1024_single_gpu"] = {"error": f"{type(e).__name__}: {e}"}
            try:
                # the 1e-4-parity mode in its faster form (split-bf16 products: same full-size parity test, tests/test_gpu_model.py f32_form) ...
                res["f32_parity_mode"] = side_record(args, device, 5, f32=True, f32_split=True)
            except Exception as e:      # noqa: BLE001
                res["f32_parity_mode"] = {"error": f"{type(e).__name__}: {e}"}
            try:
                # ... and in the exact form (one k-ordered fmaf chain per output)
                res["f32_exact_mode"] = side_record(args, device, 3, f32=True, f32_split=False)
            except Exception as e:      # noqa: BLE001
                res["f32_exact_mode"] = {"error": f"{type(e).__name__}: {e}"}
        print(json.dumps(res), file=result_stream, flush=True)
    if dist_on:
        tdist.barrier()
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()
