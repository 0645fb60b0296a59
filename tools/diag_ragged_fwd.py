"""Forward-only repetition of the ragged f32 case's image encoder: are the saved activations of layer3.1 (unit 0) identical across runs?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import contextlib
import torch
import test_gpu_model as T
from detfill import det_tensor
from clip_lite_amd.resnet import resnet_forward

with contextlib.redirect_stdout(sys.stderr):
    M = T.build("resnet18", "train_sbert", 1, False, 512)
rt, net = M.runtime, M.image_encoder.img_encoder
img = det_tensor("image", (6, 3, 96, 160), "normal").cuda()
ref = None
for i in range(12):
    feat, saved = resnet_forward(rt, net, img, True)
    torch.cuda.synchronize()
    blocks = saved["recs"]
    # block index of layer3.1 in resnet18: layer1 (2) + layer2 (2) + layer3.0 -> index 5
    units = blocks[5][0]
    u = units[0]
    y, out = u.y.float().cpu(), u.out.float().cpu()
    st = u.stats.t.float().cpu().view(u.stats.R, 3, -1).sum(0)
    cur = (y, out, st)
    if ref is None:
        ref = cur
        print("rows", y.shape, "channel 137: positives", int((out[:, 137] > 0).sum()), " channel 4:", int((out[:, 4] > 0).sum()))
    else:
        dy = (y - ref[0]).abs().max().item()
        flips = ((out > 0) != (ref[1] > 0)).sum(0)
        fl = flips.nonzero().flatten().tolist()
        print(f"rep {i}: max |y - y0| {dy:.2e}; channels with mask flips: {[(c, int(flips[c])) for c in fl][:10]}; stats diff {(st - ref[2]).abs().max().item():.2e}")
