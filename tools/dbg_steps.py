import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import torch
from test_gpu_train_step import _models, _optim, _batch
from detfill import det_tensor
from oracle import ref_model as O
from clip_lite_amd.optim.lr_scheduler import LinearWarmupCosineAnnealingLR
M, Mo = _models()
opt = _optim(M)
sched = LinearWarmupCosineAnnealingLR(opt, total_steps=40, warmup_steps=3)
opt_o = O.build_optimizer(Mo.named_parameters(), cnn_lr=0.2, trans_lr=1e-3, lr=1e-3, k=5, alpha=0.5)
for step in range(6):
    b = _batch(step)
    u = (det_tensor(f"u1{step}", (4, 512), "uniform"), det_tensor(f"u2{step}", (4, 768), "uniform"))
    M.loss.set_prior_noise(u[0].cuda(), u[1].cuda()); Mo.loss.noise = u
    opt.zero_grad()
    out = M({k: v.cuda() for k, v in b.items()}); out["loss"].backward()
    gn = opt.clip_grad_norm(0.5 if step == 2 else 10.0)
    lrs = [g["lr"] for g in opt.param_groups][:2]
    opt.step(); sched.step()
    ref, gno = O.train_step(Mo, opt_o, b, step, sched=("cosine", 40, 3, 0.0), clip=0.5 if step == 2 else 10.0)
    so = Mo.state_dict(); worst = (0, "")
    for k, v in M.state_dict().items():
        if v.dtype.is_floating_point and "running" not in k:
            e = (v.float().cpu() - so[k]).abs().max().item() / max(so[k].abs().max().item(), 1.0)
            if e > worst[0]: worst = (e, k)
    print(f"step {step}: loss {out['loss'].item():.6f} vs {ref['loss'].item():.6f}  gradnorm {gn.item():.4f} vs {gno.item():.4f}  lr {lrs}  worst param diff {worst}")
