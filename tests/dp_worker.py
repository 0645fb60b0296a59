"""Worker of tests/test_gpu_train_step.py::test_two_rank_gpu_steps_match_shardwise_oracle — test infrastructure, launched by
`python -m torch.distributed.run --nproc-per-node 2 tests/dp_worker.py OUT.npz STEPS [graph|eager] [bf16]`. Every rank builds the same deterministically
filled ResNet-18 + 1-layer BERT + JSD heads in the exact-f32 deterministic-reduction mode, trains STEPS data-parallel steps on ITS OWN
shard (images, captions and prior noise are functions of (step, rank): detfill.det_tensor) through TrainStep + GradientExchange — the
product's data-parallel path (reference train.py:174-178) — and rank 0 writes the per-step losses of both ranks, the final parameters and its
BatchNorm buffers. Both ranks use cuda:0 (one GPU on the test box), so the process group is gloo."""
import os
import sys

import numpy as np
import torch
import torch.distributed as tdist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from detfill import det_fill, det_tensor      # noqa: E402

B, S, L = 4, 64, 9
CNN_LR = 0.01


def shard(step, rank):
    ids = torch.randint(1000, 30522, (B, L), generator=torch.Generator().manual_seed(1000 * step + rank))
    ids[:, 0], ids[:, -1] = 101, 102
    return {"image": det_tensor(f"img{step}r{rank}", (B, 3, S, S), "normal"), "input_ids": ids, "attention_mask": torch.ones(B, L, dtype=torch.long)}


def noise(step, rank):
    return det_tensor(f"u1{step}r{rank}", (B, 512), "uniform"), det_tensor(f"u2{step}r{rank}", (B, 768), "uniform")


def main():
    out_path, steps = sys.argv[1], int(sys.argv[2])
    graph = len(sys.argv) > 3 and sys.argv[3] == "graph"
    lowp = len(sys.argv) > 4 and sys.argv[4] == "bf16"          # bf16 kernels, fast reductions: the grouped weight gradients' staged hand-over
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    tdist.init_process_group(backend="gloo", init_method="env://")
    from clip_lite_amd import hip
    from clip_lite_amd.encoder import ImageEncoder, TextEncoder
    from clip_lite_amd.loss import JSDInfoMaxLoss
    from clip_lite_amd.model import VLInfoModel
    from clip_lite_amd.optim import FusedSGD, Lookahead
    from clip_lite_amd.optim.lr_scheduler import LinearWarmupCosineAnnealingLR
    from clip_lite_amd.train_loop import TrainStep
    from clip_lite_amd.utils import distributed as cdist
    from clip_lite_amd.utils.common import GradScaler
    hip.set_deterministic(not lowp)
    te = TextEncoder(mode="train_sbert", num_hidden_layers=1)
    te.strans.hidden_dropout_prob = te.strans.attention_probs_dropout_prob = 0.0
    M = det_fill(VLInfoModel(te, ImageEncoder("resnet18"), JSDInfoMaxLoss(512, 768, "dot", 0.1, True, True), "train_sbert", is_amp=lowp)).to("cuda").train()
    cdist.broadcast_parameters(M)
    groups = [{"params": [p], "lr": CNN_LR if "image_encoder" in n else 1e-3, "weight_decay": 1e-4} for n, p in M.named_parameters()]
    opt = Lookahead(FusedSGD(groups, momentum=0.9), k=5, alpha=0.5)
    sched = LinearWarmupCosineAnnealingLR(opt, total_steps=40, warmup_steps=1)
    ex = cdist.GradientExchange(M.runtime.arena)
    M.runtime.exchange = ex
    step_fn = TrainStep(M, opt, sched, GradScaler(False), 10.0, ex, graph=graph, graph_warmup=1)
    losses = []
    # the pinned prior noise lives in two persistent device tensors refilled before every step: a captured step keeps the ADDRESSES it recorded
    n1, n2 = (torch.empty(B, d, device="cuda", dtype=torch.bfloat16 if lowp else torch.float32) for d in (512, 768))
    M.loss.set_prior_noise(n1, n2)
    for s in range(steps):
        u1, u2 = noise(s, rank)
        n1.copy_(u1)
        n2.copy_(u2)
        out = step_fn({k: v.cuda() for k, v in shard(s, rank).items()})
        mine = torch.tensor([out["loss"].item()], dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        tdist.all_gather(every, mine)
        losses.append([e.item() for e in every])
    torch.cuda.synchronize()
    bits = M.runtime.arena.flat_p.view(torch.int32).to(torch.int64).cpu()
    mine = torch.stack([bits.sum(), bits[::7].sum()])
    every = [torch.zeros_like(mine) for _ in range(world)]
    tdist.all_gather(every, mine)
    if rank == 0:
        state = {k: v.detach().float().cpu().numpy() for k, v in M.state_dict().items() if v.dtype.is_floating_point}
        np.savez(out_path, __losses__=np.array(losses), __identical__=np.array(all(torch.equal(every[0], e) for e in every)),
                 __replays__=np.array(getattr(step_fn, "replays", 0)), **state)
    tdist.barrier()
    tdist.destroy_process_group()


if __name__ == "__main__":
    main()
