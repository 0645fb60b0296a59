"""Repeat the ragged / odd-batch / non-square f32 case (tests/test_gpu_model.py) in one process and print, per repetition, the worst parameter-gradient
error against the fp64 oracle — fast (float-atomic) reductions vs the deterministic mode:  python tools/diag_ragged.py [reps]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import contextlib
import torch
import test_gpu_model as T
from clip_lite_amd import hip

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
import clip_lite_amd.model as MM
variant = sys.argv[2] if len(sys.argv) > 2 else ''
_orig_build = T.build
def _build(*a, **k):
    M = _orig_build(*a, **k)
    if variant == 'nooverlap':
        M.overlap_encoders = False
    if variant == 'nofuse':
        M.runtime.fuse_bn_backward = False
    return M
T.build = _build
for det in (False,):
    hip.set_deterministic(det)
    for i in range(reps):
        with contextlib.redirect_stdout(sys.stderr):
            M, Mo, Md, out, ref = T.run_case("resnet18", "train_sbert", 1, False, 6, 96, 13, 512, W=160, ragged=True)
        rows = T.grad_report(M, Mo, Md)
        gmax = max(r[2] for r in rows)
        worst = max(rows, key=lambda r: r[0] / max(r[2], 1e-3 * gmax))
        nbad = sum(1 for e, eo, s, k in rows if e > max(2e-3 * max(s, 1e-3 * gmax), 16 * eo))
        if nbad:
            gd = {k: p_.grad for k, p_ in Md.named_parameters()}
            for name in ("image_encoder.img_encoder.layer3.1.bn1.bias", "image_encoder.img_encoder.layer3.1.bn1.weight", "image_encoder.img_encoder.layer3.1.conv1.weight"):
                g = dict(M.named_parameters())[name].grad.detach().double().cpu()
                t = gd[name]
                d = (g - t).abs()
                if d.dim() > 1:
                    d = d.flatten(1).max(1).values
                bad_idx = (d > 1e-3 * t.abs().max()).nonzero().flatten().tolist()
                print(f"      {name}: {len(bad_idx)} bad output channels of {d.numel()}: {bad_idx[:24]}  max err {d.max():.3e}")
            for e, eo, sc, k in rows:
                if e > max(2e-3 * max(sc, 1e-3 * gmax), 16 * eo):
                    print(f"      {k:60s} err {e:.3e} scale {sc:.3e} rel {e / max(sc, 1e-30):.3f}")
        print(f"det={int(det)} rep {i}: loss diff {abs(out['loss'].item() - ref['loss'].item()):.2e}  worst rel err {worst[0] / max(worst[2], 1e-3 * gmax):.2e} ({worst[3]})  tensors over the bar: {nbad}")
hip.set_deterministic(False)
