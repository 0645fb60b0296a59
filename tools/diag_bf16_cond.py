"""How well-conditioned is the whole-model bf16 backward comparison? ResNet-50 + 2-layer BERT + JSD heads at batch 64 (128 x 128 images) with
the residual-branch BatchNorm gains (bn3 / BasicBlock bn2) scaled by s: cosine between the bf16 HIP gradients and the fp32 oracle's, per
top-level module, and the worst relative L2 error over large weight tensors. At s = 1 (default init) a randomly initialised 50-layer
BatchNorm network amplifies 2^-9 roundings into decorrelated gradients (the fp32 oracle with emulated bf16 storage does the same:
tests/test_gpu_model.py); smaller s shows what the kernels do on a conditioned problem. Feeds the bound of
tests/test_gpu_ops.py::test_resnet50_bert_bf16_backward_against_fp32_oracle. Usage (GPU box): python tools/diag_bf16_cond.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402


def run(scale, B=64, S=128, L=30, layers=2):
    from detfill import det_tensor
    from oracle import ref_model as O
    from clip_lite_amd.encoder import ImageEncoder, TextEncoder
    from clip_lite_amd.loss import JSDInfoMaxLoss
    from clip_lite_amd.model import VLInfoModel
    torch.manual_seed(21)
    Mo = O.build_oracle_model("resnet50", "train_sbert", layers, dropout=0.0).train()
    with torch.no_grad():
        for n, p in Mo.named_parameters():
            if n.endswith("bn3.weight"):
                p.mul_(scale)
    te = TextEncoder(mode="train_sbert", num_hidden_layers=layers)
    te.strans.hidden_dropout_prob = te.strans.attention_probs_dropout_prob = 0.0
    M = VLInfoModel(te, ImageEncoder("resnet50"), JSDInfoMaxLoss(2048, 768, "dot", 0.1, True, True), "train_sbert", is_amp=True)
    M.load_state_dict(Mo.state_dict())
    M = M.to("cuda").train()
    ids = torch.randint(1000, 30522, (B, L), generator=torch.Generator().manual_seed(2))
    ids[:, 0], ids[:, -1] = 101, 102
    batch = {"image": det_tensor("bimg", (B, 3, S, S), "normal"), "input_ids": ids, "attention_mask": torch.ones(B, L, dtype=torch.long)}
    u = (det_tensor("bu1", (B, 2048), "uniform"), det_tensor("bu2", (B, 768), "uniform"))
    M.loss.set_prior_noise(u[0].cuda(), u[1].cuda())
    Mo.loss.noise = u
    out = M({k: v.cuda() for k, v in batch.items()})
    out["loss"].backward()
    torch.cuda.synchronize()
    ref = Mo(batch)
    ref["loss"].backward()
    go = dict(Mo.named_parameters())
    worst, per_top = ("", 0.0), {}
    for n, p in M.named_parameters():
        a, b = p.grad.detach().float().cpu(), go[n].grad
        acc = per_top.setdefault(n.split(".")[0], [0.0, 0.0, 0.0])
        acc[0] += (a * b).sum().item(); acc[1] += (a * a).sum().item(); acc[2] += (b * b).sum().item()
        if p.dim() >= 2 and p.numel() > 4096:
            rel = ((a - b).norm() / b.norm().clamp_min(1e-12)).item()
            if rel > worst[1]:
                worst = (n, rel)
    cos = {k: round(v[0] / (v[1] * v[2]) ** 0.5, 4) for k, v in per_top.items()}
    print(f"residual gain x{scale}: loss hip {out['loss'].item():.5f} oracle {ref['loss'].item():.5f}  cosines {cos}  worst weight rel-L2 {worst}", flush=True)


if __name__ == "__main__":
    torch.set_num_threads(16)
    for s in (1.0, 0.5, 0.25, 0.1, 0.0):
        run(s)
