"""Pins the oracle (oracle/ref_model.py) to the REFERENCE: every fixture in tests/golden/ was produced by importing
4m4n5/CLIP-Lite itself (tests/golden/make_golden.py); here the oracle is run on the same seeded inputs with the same
deterministic weights and must reproduce the reference's outputs and gradients. Tolerance: 1e-5 relative (fp32 CPU both
sides; differences are op-ordering only). Runs without a GPU."""
import os

import numpy as np
import pytest
import torch

from detfill import det_fill
from oracle import ref_model as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return {k: v for k, v in np.load(os.path.join(G, name + ".npz"), allow_pickle=False).items()}


def close(got, ref, rel=1e-5, what=""):
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
    err = np.abs(got - ref).max()
    assert err <= rel * max(np.abs(ref).max(), 1e-3), (what, err, np.abs(ref).max())


def check_grads(module, fx, rel=2e-5, floor=2e-2):
    g = {k: p.grad for k, p in module.named_parameters() if p.grad is not None}
    names = [str(n) for n in fx["gnames"]]
    assert sorted(g) == names
    for n, ref in zip(names, fx["gnorms"]):
        assert abs(g[n].norm().item() - ref) <= rel * max(ref, floor), (n, g[n].norm().item(), ref)   # floor: gradients that are sums of cancelling terms
        # (temperature; BERT key biases, whose true gradient is exactly zero) carry absolute, not relative, rounding noise


@pytest.mark.parametrize("name,idim", [("loss_b8_rn18", 512), ("loss_b6_rn50", 2048)])
def test_loss_heads_match_reference(name, idim):
    fx = load(name)
    L = det_fill(O.OracleJSDInfoMaxLoss(idim, 768, "dot", 0.1, True, True)).train()
    img = torch.tensor(fx["img"], requires_grad=True)
    txt = torch.tensor(fx["txt"], requires_grad=True)
    L.noise = (torch.tensor(fx["u_img"]), torch.tensor(fx["u_txt"]))
    d = L(img, txt)
    d["total_loss"].backward()
    close(d["total_loss"], fx["total"], what="total")
    close(d["cross_modal_loss"], fx["cross"], what="cross")
    close(img.grad, fx["d_img"], 2e-5, "d_img")
    close(txt.grad, fx["d_txt"], 2e-5, "d_txt")
    check_grads(L, fx)
    sd = L.state_dict()
    close(sd["global_d.img_block.feature_nonlinear.1.running_mean"], fx["bn_img_rm"], what="running_mean (updated twice)")
    close(sd["global_d.img_block.feature_nonlinear.1.running_var"], fx["bn_img_rv"], what="running_var (updated twice)")
    assert int(sd["global_d.img_block.feature_nonlinear.1.num_batches_tracked"]) == int(fx["bn_img_nbt"]) == 2
    L.eval()
    with torch.no_grad():
        close(L.global_d.img_block(img), fx["eval_proj_img"], what="eval projection")


@pytest.mark.parametrize("name,layers", [("text_l2_b4_len7_ragged", 2), ("text_l1_b3_len30", 1)])
def test_text_encoder_matches_reference(name, layers):
    fx = load(name)
    te = O.OracleTextEncoder(mode="train_sbert", num_hidden_layers=layers)
    for m in te.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    det_fill(te).train()
    out = te({"input_ids": torch.tensor(fx["ids"]), "attention_mask": torch.tensor(fx["mask"])})
    close(out, fx["out"], what="pooler_output")
    (out * torch.tensor(fx["w"])).sum().backward()
    check_grads(te, fx, 5e-5, 0.2)
    close(te.strans.pooler.dense.bias.grad, fx["g_pooler_b"], 2e-5)
    close(te.strans.embeddings.position_embeddings.weight.grad[:fx["g_pos"].shape[0]], fx["g_pos"], 5e-5)


@pytest.mark.parametrize("name,mode,layers", [("model_rn18_sbert_b4", "sbert", 0), ("model_rn18_bert1_b4", "train_sbert", 1)])
def test_whole_model_matches_reference_wrapper(name, mode, layers):
    fx = load(name)
    M = O.build_oracle_model("resnet18", mode, max(layers, 1), dropout=0.0)
    det_fill(M).train()
    batch = {"image": torch.tensor(fx["image"])}
    if mode == "sbert":
        batch["caption_encodings"] = torch.tensor(fx["caption_encodings"])
    else:
        batch["input_ids"] = torch.tensor(fx["input_ids"])
        batch["attention_mask"] = torch.tensor(fx["attention_mask"])
    M.loss.noise = (torch.tensor(fx["u_img"]), torch.tensor(fx["u_txt"]))
    out = M(batch)
    out["loss"].backward()
    close(out["loss"], fx["total"], what="total")
    close(out["loss_components"]["cross_modal_loss"], fx["cross"], what="cross")
    check_grads(M, fx, 1e-4, 0.2)
    close(M.image_encoder.img_encoder.conv1.weight.grad, fx["g_conv1"], 1e-4)


def test_update_path_matches_reference():
    from detfill import det_tensor
    fx = load("optim")
    ps = [torch.nn.Parameter(det_tensor(f"p{i}", s, "normal")) for i, s in enumerate([(7, 5), (12,), (3, 4, 2)])]
    names = ["image_encoder.a", "text_encoder.b", "loss.c"]
    opt = O.build_optimizer(zip(names, ps), cnn_lr=0.2, trans_lr=1e-3, lr=1e-3, k=5, alpha=0.5, no_decay="loss.*")
    for step in range(1, 8):
        mult = O.lr_multiplier("cosine", step - 1, 20, 4, 0.0)
        for g in opt.param_groups:
            g["lr"] = g["initial_lr"] * mult
        assert np.allclose([g["lr"] for g in opt.param_groups], fx["lrs"][step - 1], rtol=1e-6, atol=0)
        opt.zero_grad()
        for i, p in enumerate(ps):
            p.grad = det_tensor(f"g{i}_{step}", tuple(p.shape), "normal") * (3.0 if step == 3 else 1.0)
        torch.nn.utils.clip_grad_norm_(ps, 10.0)
        opt.step()
        if step in (1, 5, 6, 7):
            for i, p in enumerate(ps):
                close(p, fx[f"p{i}_step{step}"], 1e-6, f"p{i} step {step}")
    ts = (0, 1, 9999, 10000, 255000, 499999, 500000)
    for cls, nm in (("LinearWarmupCosineAnnealingLR", "cosine"), ("LinearWarmupLinearDecayLR", "linear"), ("LinearWarmupNoDecayLR", "none")):
        assert np.allclose([O.lr_multiplier(nm, t, 500000, 10000, 0.0) for t in ts], fx["mult_" + cls], rtol=1e-12, atol=1e-15)
    assert np.allclose([O.lr_multiplier("multistep", t, 100, 10, 0.0, (30, 60), 0.1) for t in (0, 5, 10, 29, 30, 59, 60, 99)],
                       fx["mult_LinearWarmupMultiStepLR"], rtol=1e-12)


def test_no_decay_regex_matches_nothing_and_param_counts():
    """KATs from SURVEY.md: the NO_DECAY regex (config.py:172) matches none of this model's parameter names, so every
    tensor decays; parameter counts of the restated architectures equal the torchvision / HF / reference numbers."""
    import re
    M = O.build_oracle_model("resnet50", "train_sbert", 12)
    names = [n for n, _ in M.named_parameters()]
    assert not any(re.match(O.NO_DECAY_DEFAULT, n) for n in names)
    cnt = lambda m: sum(p.numel() for p in m.parameters())
    assert cnt(M.image_encoder) == 23508032
    assert cnt(M.text_encoder) == 109482240
    assert cnt(M.loss) == 23166323
    assert cnt(M) == 156156595
    assert cnt(O.OracleResNet("resnet18")) == 11176512 and cnt(O.OracleResNet("resnet101")) == 42500160
    opt = O.build_optimizer(M.named_parameters(), lookahead=False)
    lrs = {n: g["lr"] for (n, _), g in zip(M.named_parameters(), opt.param_groups)}
    assert lrs["image_encoder.img_encoder.conv1.weight"] == 0.2 and lrs["text_encoder.strans.pooler.dense.bias"] == 1e-3
    assert lrs["loss.global_d.temperature"] == 1e-3 and all(g["weight_decay"] == 1e-4 for g in opt.param_groups)
