"""Embedding extraction and image-text retrieval on the trained model (reference retrieval.py:66-210), on the HIP kernels.

The reference puts the two encoders and the two mutual-information projection heads in eval mode, L2-normalises the projected
features and scores every image against every caption with one matrix product (`image_embeds @ text_embeds.t()`,
retrieval.py:143), then reports recall@1/5/10 in both directions (`itm_eval`, retrieval.py:150-207). Here the encoders and the heads
are the same executors as in training (BatchNorm on running statistics, no dropout), the normalisation is `clite_l2_normalize` and the
N x M similarity is one `clite_gemm_nt` on the MFMA engine. Tokenisation / datasets stay outside (the caller passes tensors).
"""
from typing import Dict, Sequence

import numpy as np
import torch

from . import hip


@torch.no_grad()
def embed_images(model, images: torch.Tensor, batch_size: int = 128) -> torch.Tensor:
    """images: f32 NCHW on the model's device -> L2-normalised projected embeddings [N][2048] (compute dtype).
    Reference retrieval.py:122-127: image_projector(image_encoder(image)) then F.normalize."""
    rt = model.runtime
    enc, proj = model.image_encoder, model.loss.global_d.img_block
    was = (enc.training, proj.training)
    enc.eval(), proj.eval()
    outs = []
    for i in range(0, images.shape[0], batch_size):
        f = proj(enc(images[i:i + batch_size]))
        o = torch.empty_like(f)
        hip.l2_normalize(rt.dt, f.contiguous(), o, f.shape[0], f.shape[1])
        outs.append(o)
    enc.train(was[0]), proj.train(was[1])
    return torch.cat(outs, 0)


@torch.no_grad()
def embed_texts(model, input_ids: torch.Tensor, attention_mask: torch.Tensor, batch_size: int = 128) -> torch.Tensor:
    """input_ids / attention_mask: int64 [N][L <= 32] -> L2-normalised projected embeddings [N][2048].
    Reference retrieval.py:90-108: text_projector(text_encoder({...})) then F.normalize."""
    rt = model.runtime
    enc, proj = model.text_encoder, model.loss.global_d.text_block
    was = (enc.training, proj.training)
    enc.eval(), proj.eval()
    outs = []
    for i in range(0, input_ids.shape[0], batch_size):
        f = proj(enc({"input_ids": input_ids[i:i + batch_size], "attention_mask": attention_mask[i:i + batch_size]}))
        o = torch.empty_like(f)
        hip.l2_normalize(rt.dt, f.contiguous(), o, f.shape[0], f.shape[1])
        outs.append(o)
    enc.train(was[0]), proj.train(was[1])
    return torch.cat(outs, 0)


@torch.no_grad()
def similarity(model, image_embeds: torch.Tensor, text_embeds: torch.Tensor) -> torch.Tensor:
    """sims[i][t] = <image_embeds[i], text_embeds[t]> as f32 [Ni][Nt] (reference retrieval.py:143). The text count is padded to a
    multiple of 8 internally (GEMM column granularity)."""
    rt = model.runtime
    Ni, D = image_embeds.shape
    Nt = text_embeds.shape[0]
    Np = (Nt + 7) // 8 * 8
    te = text_embeds.contiguous()
    if Np != Nt:
        te = torch.cat([te, torch.zeros(Np - Nt, D, device=te.device, dtype=te.dtype)], 0)
    out = torch.empty(Ni, Np, device=image_embeds.device, dtype=torch.float32)
    hip.gemm_nt(rt.dt, image_embeds.contiguous(), te, Ni, Np, D, hip.epilogue(out, Np, out_f32=True))
    return out[:, :Nt]


def itm_eval(scores_i2t: np.ndarray, scores_t2i: np.ndarray, txt2img: Sequence[int], img2txt: Dict[int, Sequence[int]],
             image_ids: Sequence[int]) -> Dict[str, float]:
    """Recall@1/5/10 for image->text and text->image retrieval; same definitions and result keys as reference retrieval.py:150-207
    (an image's rank is the best rank among its own captions; a caption's rank is the rank of its image)."""
    image_ids = [int(i) for i in image_ids]
    img2idx = {img_id: idx for idx, img_id in enumerate(image_ids)}
    order = np.argsort(-scores_i2t, axis=1, kind="stable")
    pos = np.empty_like(order)
    rows = np.arange(order.shape[0])[:, None]
    pos[rows, order] = np.arange(order.shape[1])[None, :]          # pos[i][t] = rank of caption t for image i
    ranks = np.array([min(pos[index, t] for t in img2txt[image_ids[index]]) for index in range(scores_i2t.shape[0])])
    tr1, tr5, tr10 = (100.0 * float(np.mean(ranks < k)) for k in (1, 5, 10))
    order_t = np.argsort(-scores_t2i, axis=1, kind="stable")
    pos_t = np.empty_like(order_t)
    rows = np.arange(order_t.shape[0])[:, None]
    pos_t[rows, order_t] = np.arange(order_t.shape[1])[None, :]
    ranks_t = np.array([pos_t[index, img2idx[int(txt2img[index])]] for index in range(scores_t2i.shape[0])])
    ir1, ir5, ir10 = (100.0 * float(np.mean(ranks_t < k)) for k in (1, 5, 10))
    tr_mean, ir_mean = (tr1 + tr5 + tr10) / 3, (ir1 + ir5 + ir10) / 3
    return {"txt_r1": tr1, "txt_r5": tr5, "txt_r10": tr10, "txt_r_mean": tr_mean, "img_r1": ir1, "img_r5": ir5, "img_r10": ir10,
            "img_r_mean": ir_mean, "r_mean": (tr_mean + ir_mean) / 2}
