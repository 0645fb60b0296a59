"""Generate the golden fixtures in tests/golden/ by running the REFERENCE itself (/root/reference) in the build container.

Run here only (`python tests/golden/make_golden.py`); the GPU box has no /root/reference and only reads the committed .npz
files. Nothing of the reference's source is copied: fixtures hold seeded inputs and the reference's numeric outputs.

Import recipe: import transformers first, stub the absent third-party modules the reference imports at
module scope (torchvision, sentence_transformers, nltk), and make Tensor.cuda a no-op for loss.py:186,257,280.
The reference's ImageEncoder cannot be constructed (torchvision absent), so for the whole-model case the reference
VLInfoModel / TextEncoder / JSDInfoMaxLoss are wrapped around the oracle's torchvision-topology ResNet."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)

import transformers  # noqa: F401  (must precede the stubs)

for name, attrs in {
    "torchvision": {}, "torchvision.models": {n: None for n in ("resnet18", "resnet34", "vgg19", "resnet50", "resnet101", "resnet152")},
    "sentence_transformers": {"SentenceTransformer": None},
    "nltk": {}, "nltk.tokenize": {"word_tokenize": None}, "nltk.corpus": {"wordnet": None},
}.items():
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
sys.modules["torchvision"].models = sys.modules["torchvision.models"]
torch.Tensor.cuda = lambda self, *a, **k: self
if not hasattr(np, "bool"):
    np.bool = bool

sys.path.insert(0, "/root/reference")
import loss as ref_loss            # noqa: E402
import encoder as ref_encoder      # noqa: E402
import model as ref_model          # noqa: E402
from optim.lookahead import Lookahead as RefLookahead                      # noqa: E402
from optim import lr_scheduler as ref_sched                                 # noqa: E402

from detfill import det_fill, det_tensor                                    # noqa: E402
from oracle.ref_model import OracleImageEncoder                             # noqa: E402

torch.set_num_threads(8)


def grads(module):
    return {k: p.grad.detach().clone() for k, p in module.named_parameters() if p.grad is not None}


def pin_noise(fn, tensors):
    """Run fn with torch.rand_like returning the fixture tensors in draw order (image, text)."""
    orig = torch.rand_like
    it = iter(tensors)
    torch.rand_like = lambda t, *a, **k: next(it).to(t.dtype)
    try:
        return fn()
    finally:
        torch.rand_like = orig


def save(name, **arrs):
    out = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()}
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, {k: v.shape for k, v in out.items()})


# ---------------------------------------------------------------- G1: loss.py heads + JSD + priors
def g_loss(tag, B, idim, tdim):
    L = det_fill(ref_loss.JSDInfoMaxLoss(image_dim=idim, text_dim=tdim, type="dot", prior_weight=0.1, image_prior=True, text_prior=True))
    L.train()
    img = det_tensor(tag + "img", (B, idim), "normal").abs().requires_grad_(True)     # post-ReLU-like image features
    txt = torch.tanh(det_tensor(tag + "txt", (B, tdim), "normal")).requires_grad_(True)
    u_img, u_txt = det_tensor(tag + "u_img", (B, idim), "uniform"), det_tensor(tag + "u_txt", (B, tdim), "uniform")
    d = pin_noise(lambda: L(img, txt), [u_img, u_txt])
    d["total_loss"].backward()
    g = grads(L)
    with torch.no_grad():
        L.eval()
        p_img = L.global_d.img_block(img)
        p_txt = L.global_d.text_block(txt)
    sd = L.state_dict()
    save(tag, img=img, txt=txt, u_img=u_img, u_txt=u_txt, total=d["total_loss"], cross=d["cross_modal_loss"],
         d_img=img.grad, d_txt=txt.grad,
         gnames=np.array(sorted(g)), gnorms=np.array([g[k].norm().item() for k in sorted(g)]),
         g_temperature=g["global_d.temperature"], g_prior_l2_w=g["prior_d.l2.weight"], g_img_ln_w=g["global_d.img_block.feature_block_ln.weight"],
         bn_img_rm=sd["global_d.img_block.feature_nonlinear.1.running_mean"], bn_img_rv=sd["global_d.img_block.feature_nonlinear.1.running_var"],
         bn_img_nbt=sd["global_d.img_block.feature_nonlinear.1.num_batches_tracked"],
         eval_proj_img=p_img, eval_proj_txt=p_txt)


# ---------------------------------------------------------------- G1b: loss.py variants (critic types, cluster negatives, SSL terms)
VARIANTS = {   # tag -> (type, cluster, visual_ssl, textual_ssl)
    "lossvar_dot_cluster": ("dot", True, False, False),
    "lossvar_concat": ("concat", False, False, False),
    "lossvar_dot_ssl": ("dot", False, True, True),
    "lossvar_condot_cluster_ssl": ("condot", True, True, True),
    "lossvar_dotcon_ssl": ("dotcon", False, True, True),
}


def g_loss_variant(tag, B=8, idim=512, tdim=768):
    ctype, cluster, vssl, tssl = VARIANTS[tag]
    L = det_fill(ref_loss.JSDInfoMaxLoss(image_dim=idim, text_dim=tdim, type=ctype, prior_weight=0.1, image_prior=True, text_prior=True,
                                         visual_self_supervised=vssl, textual_self_supervised=tssl))
    L.train()
    feats = {"img": det_tensor(tag + "img", (B, idim), "normal").abs(), "txt": torch.tanh(det_tensor(tag + "txt", (B, tdim), "normal"))}
    if cluster:
        feats["nimg"] = det_tensor(tag + "nimg", (B, idim), "normal").abs()
        feats["ntxt"] = torch.tanh(det_tensor(tag + "ntxt", (B, tdim), "normal"))
    if vssl:
        feats["aimg"] = det_tensor(tag + "aimg", (B, idim), "normal").abs()
    if tssl:
        feats["atxt"] = torch.tanh(det_tensor(tag + "atxt", (B, tdim), "normal"))
    for v in feats.values():
        v.requires_grad_(True)
    u_img, u_txt = det_tensor(tag + "u_img", (B, idim), "uniform"), det_tensor(tag + "u_txt", (B, tdim), "uniform")
    d = pin_noise(lambda: L(feats["img"], feats["txt"], neg_image_features=feats.get("nimg"), neg_text_features=feats.get("ntxt"),
                            aug_image_features=feats.get("aimg"), aug_text_features=feats.get("atxt")), [u_img, u_txt])
    d["total_loss"].backward()
    g = grads(L)
    out = {k: v for k, v in feats.items()}
    out.update({"d_" + k: v.grad for k, v in feats.items()})
    save(tag, u_img=u_img, u_txt=u_txt, total=d["total_loss"], cross=d["cross_modal_loss"], visual=d["visual_loss"], textual=d["textual_loss"],
         gnames=np.array(sorted(g)), gnorms=np.array([g[k].norm().item() for k in sorted(g)]), **out)


# ---------------------------------------------------------------- G2: encoder.TextEncoder (HF BertModel), dropout pinned to 0
def g_text(tag, B, Ls, layers, ragged):
    te = ref_encoder.TextEncoder(word_dict={}, mode="train_sbert", num_hidden_layers=layers)
    for m in te.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    det_fill(te)
    te.train()
    g = torch.Generator().manual_seed(1234)
    ids = torch.randint(1000, 30522, (B, Ls), generator=g)
    ids[:, 0] = 101
    mask = torch.ones(B, Ls, dtype=torch.long)
    if ragged:
        for b in range(B):
            n = Ls - (b % 3)
            mask[b, n:] = 0
            ids[b, n:] = 0
            ids[b, n - 1] = 102
    else:
        ids[:, -1] = 102
    out = te({"input_ids": ids, "attention_mask": mask})
    w = det_tensor(tag + "w", tuple(out.shape), "normal")
    (out * w).sum().backward()
    gr = grads(te)
    save(tag, ids=ids, mask=mask, out=out, w=w, gnames=np.array(sorted(gr)), gnorms=np.array([gr[k].norm().item() for k in sorted(gr)]),
         g_pooler_b=gr["strans.pooler.dense.bias"], g_ln0_w=gr["strans.embeddings.LayerNorm.weight"],
         g_pos=gr["strans.embeddings.position_embeddings.weight"][:Ls], g_q0_b=gr["strans.encoder.layer.0.attention.self.query.bias"])


# ---------------------------------------------------------------- G3: model.VLInfoModel (reference wrapper + loss + text encoder)
def g_model(tag, visual, mode, B, S, Ls, layers):
    ie = OracleImageEncoder(visual)
    te = ref_encoder.TextEncoder(word_dict={}, mode=mode, num_hidden_layers=layers)
    for m in te.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    idim = ie.img_encoder.out_dim
    L = ref_loss.JSDInfoMaxLoss(image_dim=idim, text_dim=768, type="dot", prior_weight=0.1, image_prior=True, text_prior=True)
    M = det_fill(ref_model.VLInfoModel(te, ie, L, mode, is_amp=False))
    M.train()
    batch = {"image": det_tensor(tag + "image", (B, 3, S, S), "normal")}
    if mode == "sbert":
        batch["caption_encodings"] = det_tensor(tag + "cap", (B, 768), "normal")
    else:
        g = torch.Generator().manual_seed(77)
        ids = torch.randint(1000, 30522, (B, Ls), generator=g)
        ids[:, 0] = 101
        ids[:, -1] = 102
        batch["input_ids"] = ids
        batch["attention_mask"] = torch.ones(B, Ls, dtype=torch.long)
    u_img, u_txt = det_tensor(tag + "u_img", (B, idim), "uniform"), det_tensor(tag + "u_txt", (B, 768), "uniform")
    out = pin_noise(lambda: M(batch), [u_img, u_txt])
    out["loss"].backward()
    gr = grads(M)
    save(tag, **{k: v for k, v in batch.items()}, u_img=u_img, u_txt=u_txt, total=out["loss"],
         cross=out["loss_components"]["cross_modal_loss"], gnames=np.array(sorted(gr)),
         gnorms=np.array([gr[k].norm().item() for k in sorted(gr)]),
         g_conv1=gr["image_encoder.img_encoder.conv1.weight"], g_temperature=gr["loss.global_d.temperature"])


# ---------------------------------------------------------------- G4: optim/lookahead.py + lr_scheduler.py
def g_optim():
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(det_tensor(f"p{i}", s, "normal")) for i, s in enumerate([(7, 5), (12,), (3, 4, 2)])]
    groups = [{"params": [p], "lr": lr, "weight_decay": wd} for p, lr, wd in zip(ps, (0.2, 1e-3, 1e-3), (1e-4, 1e-4, 0.0))]
    opt = RefLookahead(torch.optim.SGD(groups, momentum=0.9), k=5, alpha=0.5)
    sched = ref_sched.LinearWarmupCosineAnnealingLR(opt, total_steps=20, warmup_steps=4, min_mult=0.0)
    snaps, lrs = {}, []
    for step in range(1, 8):
        opt.zero_grad()
        for i, p in enumerate(ps):
            p.grad = det_tensor(f"g{i}_{step}", tuple(p.shape), "normal") * (3.0 if step == 3 else 1.0)
        torch.nn.utils.clip_grad_norm_(ps, 10.0)
        lrs.append([g["lr"] for g in opt.param_groups])
        opt.step()
        sched.step()
        if step in (1, 5, 6, 7):
            for i, p in enumerate(ps):
                snaps[f"p{i}_step{step}"] = p.detach().clone()
    mults = {}
    for cls, kw in (("LinearWarmupCosineAnnealingLR", {"min_mult": 0.0}), ("LinearWarmupLinearDecayLR", {}), ("LinearWarmupNoDecayLR", {})):
        o = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
        s = getattr(ref_sched, cls)(o, total_steps=500000, warmup_steps=10000, **kw)
        mults[cls] = np.array([s._lr_multiplier(t) for t in (0, 1, 9999, 10000, 255000, 499999, 500000)])
    o = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
    s = ref_sched.LinearWarmupMultiStepLR(o, total_steps=100, warmup_steps=10, milestones=[30, 60], gamma=0.1)
    mults["LinearWarmupMultiStepLR"] = np.array([s._lr_multiplier(t) for t in (0, 5, 10, 29, 30, 59, 60, 99)])
    save("optim", lrs=np.array(lrs), **snaps, **{"mult_" + k: v for k, v in mults.items()})


if __name__ == "__main__":
    g_loss("loss_b8_rn18", 8, 512, 768)
    g_loss("loss_b6_rn50", 6, 2048, 768)
    for tag in VARIANTS:
        g_loss_variant(tag)
    g_text("text_l2_b4_len7_ragged", 4, 7, 2, True)
    g_text("text_l1_b3_len30", 3, 30, 1, False)
    g_model("model_rn18_sbert_b4", "resnet18", "sbert", 4, 64, 0, 0)
    g_model("model_rn18_bert1_b4", "resnet18", "train_sbert", 4, 64, 9, 1)
    g_optim()
