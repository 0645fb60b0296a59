"""Run-to-run spread of the fast (float-atomic) reduction mode on the checkpoint-resume test problem: N uninterrupted 4-step runs of
tests/test_gpu_train_step.py's problem from identical state, per-tensor relative L2 difference against the first run. The tolerance of
test_checkpoint_resume_equivalence_fast_mode is derived from this output (run on MI355X: python tools/diag_spread.py [runs])."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(runs=8):
    import test_gpu_train_step as T
    from clip_lite_amd import hip
    from clip_lite_amd.optim.lr_scheduler import LinearWarmupCosineAnnealingLR
    from detfill import det_tensor

    def one():
        M, _ = T._models()
        opt = T._optim(M, k=50)
        sched = LinearWarmupCosineAnnealingLR(opt, total_steps=40, warmup_steps=3)
        for s in range(4):
            M.loss.set_prior_noise(det_tensor(f"u1{s}", (4, 512), "uniform").cuda(), det_tensor(f"u2{s}", (4, 768), "uniform").cuda())
            opt.zero_grad()
            M({k: v.cuda() for k, v in T._batch(s).items()})["loss"].backward()
            opt.clip_grad_norm(10.0)
            opt.step()
            sched.step()
        return {k: v.float().cpu().clone() for k, v in M.state_dict().items()}

    for det in (False, True):
        hip.set_deterministic(det)
        ref = one()
        worst = {}
        for _ in range(runs - 1):
            cur = one()
            for k in ref:
                rel = (cur[k] - ref[k]).norm().item() / max(ref[k].norm().item(), 1e-3)
                worst[k] = max(worst.get(k, 0.0), rel)
        vals = sorted(worst.values())
        top = sorted(worst.items(), key=lambda kv: -kv[1])[:5]
        print(f"deterministic={det}: {runs} runs, per-tensor relative L2 spread: max {vals[-1]:.3e}, median {vals[len(vals) // 2]:.3e}")
        for k, v in top:
            print(f"    {v:.3e}  {k}")
    hip.set_deterministic(False)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 8)
