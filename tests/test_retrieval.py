"""Retrieval path (reference retrieval.py:66-207): recall bookkeeping on the host (CPU test) and embedding extraction +
similarity through the C ABI against the oracle's modules in eval mode (GPU test)."""
import numpy as np
import pytest
import torch


def _itm_eval_restated(scores_i2t, scores_t2i, txt2img, img2txt, image_ids):
    """The reference's loops (retrieval.py:150-207) restated literally; ties are avoided in the inputs."""
    img2idx = {int(i): k for k, i in enumerate(image_ids)}
    ranks = np.zeros(scores_i2t.shape[0])
    for index, score in enumerate(scores_i2t):
        inds = np.argsort(score)[::-1]
        ranks[index] = min(np.where(inds == i)[0][0] for i in img2txt[int(image_ids[index])])
    tr = [100.0 * len(np.where(ranks < k)[0]) / len(ranks) for k in (1, 5, 10)]
    ranks = np.zeros(scores_t2i.shape[0])
    for index, score in enumerate(scores_t2i):
        inds = np.argsort(score)[::-1]
        ranks[index] = np.where(inds == img2idx[int(txt2img[index])])[0][0]
    ir = [100.0 * len(np.where(ranks < k)[0]) / len(ranks) for k in (1, 5, 10)]
    return tr, ir


def test_itm_eval_matches_reference_definition():
    from clip_lite_amd.retrieval import itm_eval
    rng = np.random.default_rng(0)
    n_img, per = 23, 3
    image_ids = rng.permutation(1000)[:n_img]
    txt2img, img2txt = [], {}
    for i in image_ids:
        img2txt[int(i)] = list(range(len(txt2img), len(txt2img) + per))
        txt2img += [int(i)] * per
    sims = rng.standard_normal((n_img, n_img * per))
    for k, i in enumerate(image_ids):          # make the right captions likely, not certain
        sims[k, img2txt[int(i)]] += 1.5
    got = itm_eval(sims, sims.T, txt2img, img2txt, image_ids)
    tr, ir = _itm_eval_restated(sims, sims.T, txt2img, img2txt, image_ids)
    assert [got["txt_r1"], got["txt_r5"], got["txt_r10"]] == pytest.approx(tr)
    assert [got["img_r1"], got["img_r5"], got["img_r10"]] == pytest.approx(ir)
    assert got["r_mean"] == pytest.approx((sum(tr) / 3 + sum(ir) / 3) / 2)
    # known answer: a perfect score matrix
    perfect = np.full((2, 4), -1.0)
    perfect[0, :2], perfect[1, 2:] = [3, 2], [2, 3]
    r = itm_eval(perfect, perfect.T, [7, 7, 9, 9], {7: [0, 1], 9: [2, 3]}, [7, 9])
    assert r["txt_r1"] == r["img_r1"] == 100.0


@pytest.mark.gpu
def test_embeddings_and_similarity_match_oracle():
    """image/text encoders + projection heads in eval mode (BatchNorm running statistics, no dropout), L2-normalised, and the N x M
    similarity matrix: exact-f32 kernels vs the oracle's modules, 5e-4 absolute on unit-norm embeddings and on the cosines."""
    from detfill import det_fill, det_tensor
    from oracle import ref_model as O
    from clip_lite_amd import retrieval as R
    from clip_lite_amd.encoder import ImageEncoder, TextEncoder
    from clip_lite_amd.loss import JSDInfoMaxLoss
    from clip_lite_amd.model import VLInfoModel
    te = TextEncoder(mode="train_sbert", num_hidden_layers=2)
    M = det_fill(VLInfoModel(te, ImageEncoder("resnet18"), JSDInfoMaxLoss(512, 768, "dot", 0.1, True, True), "train_sbert", is_amp=False)).to("cuda").train()
    Mo = det_fill(O.build_oracle_model("resnet18", "train_sbert", 2, dropout=0.0)).eval()
    Ni, Nt, L = 5, 11, 9
    img = det_tensor("rimg", (Ni, 3, 64, 64), "normal")
    ids = torch.randint(1000, 30522, (Nt, L), generator=torch.Generator().manual_seed(5))
    mask = torch.ones(Nt, L, dtype=torch.long)
    mask[3, 6:] = 0
    ie = R.embed_images(M, img.cuda(), batch_size=2)
    tt = R.embed_texts(M, ids.cuda(), mask.cuda(), batch_size=4)
    assert M.training and M.image_encoder.img_encoder.training          # modes restored
    sims = R.similarity(M, ie, tt)
    with torch.no_grad():
        ie_o = torch.nn.functional.normalize(Mo.loss.global_d.img_block(Mo.image_encoder(img)), p=2, dim=-1)
        tt_o = torch.nn.functional.normalize(Mo.loss.global_d.text_block(Mo.text_encoder({"input_ids": ids, "attention_mask": mask})), p=2, dim=-1)
    assert (ie.float().cpu() - ie_o).abs().max().item() < 5e-4
    assert (tt.float().cpu() - tt_o).abs().max().item() < 5e-4
    assert sims.shape == (Ni, Nt) and (sims.cpu() - ie_o @ tt_o.t()).abs().max().item() < 5e-4


def _infonce_torch(f1, f2, tau):
    """The InfoNCE variant as defined in SURVEY §8f N4 / BASELINE config 4 (there is no reference code): symmetric cross-entropy with
    diagonal targets over exp(tau) * normalize(f1) @ normalize(f2).T."""
    import torch.nn.functional as F
    S = torch.exp(tau) * F.normalize(f1, dim=-1) @ F.normalize(f2, dim=-1).t()
    tgt = torch.arange(S.shape[0])
    return 0.5 * (F.cross_entropy(S, tgt) + F.cross_entropy(S.t(), tgt))


@pytest.mark.gpu
@pytest.mark.usefixtures("deterministic_reductions")
def test_infonce_loss_matches_torch_definition():
    """InfoNCELoss (MFMA similarity GEMM + lse / gradient kernels) against a plain-torch evaluation of the same definition on the
    oracle's modules: loss within 1e-4, every parameter gradient of the heads, the temperature and both encoders' last layers within
    2e-3 of max|reference| (exact-f32 kernels; dropout off, prior noise pinned)."""
    from detfill import det_fill, det_tensor
    from oracle import ref_model as O
    from clip_lite_amd.encoder import ImageEncoder, TextEncoder
    from clip_lite_amd.loss import InfoNCELoss
    from clip_lite_amd.model import VLInfoModel
    B, L = 8, 9
    te = TextEncoder(mode="train_sbert", num_hidden_layers=1)
    te.strans.hidden_dropout_prob = te.strans.attention_probs_dropout_prob = 0.0
    M = det_fill(VLInfoModel(te, ImageEncoder("resnet18"), InfoNCELoss(512, 768, "dot", 0.1, True, True), "train_sbert", is_amp=False)).to("cuda").train()
    Mo = det_fill(O.build_oracle_model("resnet18", "train_sbert", 1, dropout=0.0)).train()
    ids = torch.randint(1000, 30522, (B, L), generator=torch.Generator().manual_seed(2))
    batch = {"image": det_tensor("nimg", (B, 3, 64, 64), "normal"), "input_ids": ids, "attention_mask": torch.ones(B, L, dtype=torch.long)}
    u1, u2 = det_tensor("u1", (B, 512), "uniform"), det_tensor("u2", (B, 768), "uniform")
    M.loss.set_prior_noise(u1.cuda(), u2.cuda())
    out = M({k: v.cuda() for k, v in batch.items()})
    out["loss"].backward()
    # reference evaluation: the oracle's encoders / heads / priors, with the JSD term replaced by the InfoNCE definition
    imf = Mo.image_encoder(batch["image"])
    txf = Mo.text_encoder({"input_ids": ids, "attention_mask": batch["attention_mask"]})
    Lo = Mo.loss
    Lo.noise = (u1, u2)
    jsd = Lo(image_features=imf, text_features=txf)
    prior = (jsd["total_loss"] - (1 - Lo.prior_weight) * jsd["cross_modal_loss"]) / Lo.prior_weight
    cross = _infonce_torch(Lo.global_d.img_block(imf), Lo.global_d.text_block(txf), Lo.global_d.temperature)
    total = (1 - Lo.prior_weight) * cross + Lo.prior_weight * prior
    Mo.zero_grad()
    total.backward()
    assert abs(out["loss"].item() - total.item()) < 1e-4, (out["loss"].item(), total.item())
    assert abs(out["loss_components"]["cross_modal_loss"].item() - cross.item()) < 1e-4
    ref = dict(Mo.named_parameters())
    worst = 0.0
    for k, p in M.named_parameters():
        g, r = p.grad.float().cpu(), ref[k].grad
        err = (g - r).abs().max().item() / max(r.abs().max().item(), 1e-3)
        worst = max(worst, err)
        if k.startswith("loss.") or "layer4.1" in k or "pooler" in k:
            assert err < 2e-3, (k, err)
    print("worst relative gradient error over all parameters", worst)
