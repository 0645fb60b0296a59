"""Diagnostic: loss trajectories of the small train-step problem under eager / graph and with / without encoder overlap."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import contextlib
import torch
from detfill import det_fill, det_tensor
from clip_lite_amd.encoder import ImageEncoder, TextEncoder
from clip_lite_amd.loss import JSDInfoMaxLoss
from clip_lite_amd.model import VLInfoModel
from clip_lite_amd.optim import FusedSGD, Lookahead
from clip_lite_amd.optim.lr_scheduler import LinearWarmupCosineAnnealingLR
from clip_lite_amd.train_loop import TrainStep
from clip_lite_amd.utils.common import GradScaler

B, L = 8, 12
batches = []
for i in range(3):
    ids = torch.randint(1000, 30522, (B, L), generator=torch.Generator().manual_seed(i))
    batches.append({"image": det_tensor(f"gimg{i}", (B, 3, 64, 64), "normal").cuda(), "input_ids": ids.cuda(), "attention_mask": torch.ones(B, L, dtype=torch.long).cuda()})


def run(graph, overlap, drop=0.1, lowp=True, k=3, clip=10.0):
    torch.manual_seed(7)
    with contextlib.redirect_stdout(sys.stderr):
        te = TextEncoder(mode="train_sbert", num_hidden_layers=2)
    te.strans.hidden_dropout_prob = te.strans.attention_probs_dropout_prob = drop
    M = det_fill(VLInfoModel(te, ImageEncoder("resnet18"), JSDInfoMaxLoss(512, 768, "dot", 0.1, True, True), "train_sbert", is_amp=lowp)).to("cuda").train()
    M.overlap_encoders = overlap
    groups = [{"params": [p], "lr": 0.01 if "image_encoder" in n else 1e-3, "weight_decay": 1e-4} for n, p in M.named_parameters()]
    opt = Lookahead(FusedSGD(groups, momentum=0.9), k=k, alpha=0.5)
    sched = LinearWarmupCosineAnnealingLR(opt, total_steps=40, warmup_steps=3)
    step = TrainStep(M, opt, sched, GradScaler(True), clip, None, graph=graph, graph_warmup=2)
    losses = [step(batches[s % 3])["loss"].item() for s in range(7)]
    torch.cuda.synchronize()
    return losses, M.runtime.arena.flat_p.clone()


for name, kw in [("eager ov", dict(graph=False, overlap=True)), ("eager no-ov", dict(graph=False, overlap=False)),
                 ("graph ov", dict(graph=True, overlap=True)), ("graph no-ov", dict(graph=True, overlap=False)),
                 ("eager ov f32", dict(graph=False, overlap=True, lowp=False)), ("eager no-ov f32", dict(graph=False, overlap=False, lowp=False))]:
    l, p = run(**kw)
    print(f"{name:20s}", " ".join(f"{x:.5f}" for x in l), f"|p|={p.norm().item():.6f}")
