"""The fused update kernels against torch.optim on the same gradients (GPU; through the C ABI): FusedAdamW (reference factories.py:439: OPTIMIZER_NAME "adamw",
torch.optim.AdamW with its defaults) incl. global-norm clipping, per-parameter lr / weight decay groups, the Lookahead wrapper, the checkpoint layout and the
factory. (FusedSGD's parity runs through the train-step tests against the oracle's torch.optim.SGD + Lookahead.)"""
import copy

import numpy as np
import pytest
import torch

from detfill import det_fill

pytestmark = pytest.mark.gpu


def _model(lowp=False, mode="sbert"):
    from clip_lite_amd.encoder import ImageEncoder, TextEncoder
    from clip_lite_amd.loss import JSDInfoMaxLoss
    from clip_lite_amd.model import VLInfoModel
    te = TextEncoder(mode=mode, num_hidden_layers=1)
    return det_fill(VLInfoModel(te, ImageEncoder("resnet18"), JSDInfoMaxLoss(512, 768, "dot", 0.1, True, True), mode, is_amp=lowp)).to("cuda").train()


def _groups(named):
    return [{"params": [p], "lr": 3e-3 if "image_encoder" in n else 1e-3, "weight_decay": 0.0 if n.endswith("bias") else 1e-2} for n, p in named]


@pytest.mark.parametrize("lookahead", [False, True])
def test_fused_adamw_matches_torch_adamw(lookahead):
    """Five steps on the same random gradients (scaled so that the global norm is clipped on some steps and not on others): parameters within 2e-6 of
    max |p| of torch.optim.AdamW + clip_grad_norm_ (+ the reference's Lookahead arithmetic, k = 2, alpha = 0.5), moments within 1e-5; the gradient arena
    is zero after every step; state_dict() has torch.optim.AdamW's layout and round-trips."""
    from clip_lite_amd.optim import FusedAdamW, Lookahead
    M = _model()
    named = list(M.named_parameters())
    ref = [torch.nn.Parameter(p.detach().clone()) for _, p in named]
    opt = FusedAdamW(_groups(named))
    wrapped = Lookahead(opt, k=2, alpha=0.5) if lookahead else opt
    topt = torch.optim.AdamW([{"params": [r], "lr": g["lr"], "weight_decay": g["weight_decay"]} for r, g in zip(ref, _groups(named))])
    slow = [r.detach().clone() for r in ref]
    gen = torch.Generator(device="cuda").manual_seed(0)
    for step in range(5):
        scale = 3e-3 if step % 2 else 3e-5          # global norm above / below the clip value of 1
        for (_, p), r in zip(named, ref):
            g = torch.randn(p.shape, device="cuda", generator=gen) * scale
            p.grad.copy_(g)
            r.grad = g.clone()
        wrapped.clip_grad_norm(1.0)
        torch.nn.utils.clip_grad_norm_(ref, 1.0)
        wrapped.step()
        topt.step()
        if lookahead and step % 2 == 1:          # reference optim/lookahead.py:88-101: every k-th step the fast weights move half way back to the slow ones
            with torch.no_grad():
                for r, s_ in zip(ref, slow):
                    r.mul_(0.5).add_(s_, alpha=0.5)
                    s_.copy_(r)
        torch.cuda.synchronize()
        assert not M.runtime.arena.flat_g.any()
        for (n, p), r in zip(named, ref):
            assert (p.detach() - r.detach()).abs().max().item() <= 2e-6 * max(r.detach().abs().max().item(), 1.0), (step, n)
    sd = opt.state_dict()
    tsd = topt.state_dict()
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"} and float(sd["state"][0]["step"]) == 5.0
    for i in (0, 7, len(named) - 1):
        for k in ("exp_avg", "exp_avg_sq"):
            a, b = sd["state"][i][k], tsd["state"][i][k]
            assert (a - b).abs().max().item() <= 1e-5 * max(b.abs().max().item(), 1e-12), (i, k)
    opt2 = FusedAdamW(_groups(named))
    opt2.load_state_dict(copy.deepcopy(sd))
    assert opt2.steps == 5 and torch.equal(opt2.flat_v, opt.flat_v) and torch.equal(opt2.flat_v2, opt.flat_v2)


def test_optimizer_factory_builds_adamw_and_it_trains_through_the_captured_step():
    """OPTIM.OPTIMIZER_NAME = adamw through the factory (reference factories.py:438-487: no momentum keyword for it) with Lookahead, then six steps of the
    captured train step (per-phase graphs, deferred update of the text encoder and the heads): every loss finite, the step really replays, and the loss
    falls on a repeated batch."""
    from clip_lite_amd.config import Config
    from clip_lite_amd.factories import OptimizerFactory
    from clip_lite_amd.optim import FusedAdamW, Lookahead
    from clip_lite_amd.optim.lr_scheduler import LinearWarmupCosineAnnealingLR
    from clip_lite_amd.train_loop import TrainStep
    from clip_lite_amd.utils.common import GradScaler
    from detfill import det_tensor
    M = _model(lowp=True)
    cfg = Config(override_list=["OPTIM.OPTIMIZER_NAME", "adamw", "OPTIM.LR", 1e-3, "OPTIM.CNN_LR", 1e-3, "OPTIM.TRANS_LR", 1e-4, "OPTIM.WEIGHT_DECAY", 1e-2])
    opt = OptimizerFactory.from_config(cfg, M.named_parameters())
    assert isinstance(opt, Lookahead) and isinstance(opt.optimizer, FusedAdamW)
    sched = LinearWarmupCosineAnnealingLR(opt, total_steps=40, warmup_steps=2)
    step = TrainStep(M, opt, sched, GradScaler(True), 10.0, None, graph=True, graph_warmup=2, defer_update=True)
    batch = {"image": det_tensor("adamw_img", (8, 3, 64, 64), "normal").cuda(), "caption_encodings": det_tensor("adamw_cap", (8, 768), "normal").cuda()}
    losses = [step(batch)["loss"].item() for _ in range(6)]
    step.finish()
    torch.cuda.synchronize()
    assert step.graph and step.replays >= 1 and all(np.isfinite(l) for l in losses)
    assert losses[-1] < losses[0], losses


def test_adamw_per_phase_graphs_with_deferred_update_match_the_eager_step_bitwise():
    """the test above builds an `sbert` model, which TrainStep captures as ONE graph (no per-phase graphs, defer_update without effect). Here
    the model is `train_sbert` (token ids, a BERT layer in the loop), i.e. the path train_loop.main takes: per-phase graphs, FusedAdamW.launch(span=...) split
    into update_img / update_rest, the text encoder's and the heads' share deferred to the start of the next step — incl. the ordering of the
    bias-correction hyper-parameter upload against the deferred share. In the deterministic-reduction mode the parameters and both moments after
    finish() must equal the eager launches of the same six steps bit for bit (dropout and prior noise on: eager and replayed steps draw the same masks)."""
    from clip_lite_amd import hip
    from clip_lite_amd.optim import FusedAdamW, Lookahead
    from clip_lite_amd.optim.lr_scheduler import LinearWarmupCosineAnnealingLR
    from clip_lite_amd.train_loop import TrainStep
    from clip_lite_amd.utils.common import GradScaler
    from detfill import det_tensor
    B, L = 8, 10
    batches = []
    for i in range(3):
        ids = torch.randint(1000, 30522, (B, L), generator=torch.Generator().manual_seed(40 + i))
        batches.append({"image": det_tensor(f"aw_img{i}", (B, 3, 64, 64), "normal").cuda(), "input_ids": ids.cuda(),
                        "attention_mask": torch.ones(B, L, dtype=torch.long).cuda()})
    hip.set_deterministic(True)
    try:
        res = []
        for graph in (False, True):
            torch.manual_seed(11)
            M = _model(lowp=True, mode="train_sbert")
            opt = Lookahead(FusedAdamW(_groups(list(M.named_parameters()))), k=3, alpha=0.5)
            sched = LinearWarmupCosineAnnealingLR(opt, total_steps=40, warmup_steps=2)
            step = TrainStep(M, opt, sched, GradScaler(True), 1.0, None, graph=graph, graph_warmup=2, defer_update=graph)
            losses = [step(batches[s % 3])["loss"].item() for s in range(6)]
            if graph:
                assert step._graphs is not None and step.replays == 4 and step._pending_rest
            step.finish()
            torch.cuda.synchronize()
            A = M.runtime.arena
            assert not A.flat_g.any()
            res.append((losses, A.flat_p.clone(), opt.optimizer.flat_v.clone(), opt.optimizer.flat_v2.clone(), A.flat_lp.clone()))
    finally:
        hip.set_deterministic(False)
    (l0, p0, v0, w0, lp0), (l1, p1, v1, w1, lp1) = res
    assert l0 == l1
    assert torch.equal(p0, p1) and torch.equal(v0, v1) and torch.equal(w0, w1) and torch.equal(lp0, lp1)
