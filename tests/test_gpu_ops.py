"""Kernel-level GPU parity (through the C ABI) of the bf16-only / bf16-specialised kernels against plain torch fp32 references evaluated
on the SAME bf16-rounded inputs: the MFMA attention forward / backward (reference: transformers BertSelfAttention behind encoder.py:165-196),
the BatchNorm-backward dgrad epilogue and the stride-2 parity-class dgrad in bf16 (torchvision Bottleneck backward behind encoder.py:36-65),
LayerNorm backward with its fused bias gradient and regenerated dropout masks, the embedding backward's run merging, and a whole-model
bf16 BACKWARD check of ResNet-50 + BERT at a conditioned size against the fp32 oracle.

Tolerances (stated per test): fp32 accumulation everywhere, so against an fp32 reference on the same inputs the error is one bf16 rounding of
the output (2^-9 relative per element) plus the bf16 rounding of intermediate operands the kernel keeps in bf16 (attention probabilities,
dS): a few 1e-3 of the tensor's max."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
BF16, F32 = 0, 1
FMIN = torch.finfo(torch.float32).min


def _hip():
    from clip_lite_amd import hip
    return hip


def _rel(got, ref):
    return ((got.float() - ref.float()).abs().max() / ref.float().abs().max().clamp_min(1e-6)).item()


def _attn_ref(qkv, mask, B, L, H, drop_mult=None):
    """softmax(q k^T / 8 + (1 - mask) * finfo.min) [* dropout multipliers] v per (batch, head); returns ctx [B*L][H*64] (fp32, with grad)."""
    x = qkv.float().view(B, L, 3, H, 64)
    q, k, v = (x[:, :, i].permute(0, 2, 1, 3) for i in range(3))                       # [B][H][L][64]
    s = q @ k.transpose(-1, -2) * 0.125 + ((1 - mask.float()) * FMIN)[:, None, None, :]
    p = torch.softmax(s, dim=-1)
    if drop_mult is not None:
        p = p * drop_mult
    return (p @ v).permute(0, 2, 1, 3).reshape(B * L, H * 64)


@pytest.mark.parametrize("B,L,H", [(16, 30, 12), (5, 32, 12), (3, 7, 4)])
def test_attention_mfma_forward_backward_match_torch(B, L, H):
    """bf16 MFMA attention (attention_mfma_{fwd,bwd}_kernel), ragged attention mask, dropout off. Bound: 1e-2 of max|ref| forward (P is fed to
    the second MFMA in bf16), 2e-2 backward (dS and P both pass through bf16)."""
    hip = _hip()
    g = torch.Generator(device="cuda").manual_seed(B * 100 + L)
    qkv = (torch.randn(B * L, 3 * H * 64, device="cuda", generator=g) * 0.7).bfloat16()
    lens = torch.randint(1, L + 1, (B,), generator=torch.Generator().manual_seed(L))
    lens[0] = L
    mask = (torch.arange(L)[None, :] < lens[:, None]).long().cuda()
    dctx = torch.randn(B * L, H * 64, device="cuda", generator=g).bfloat16()
    ctx = torch.empty(B * L, H * 64, device="cuda", dtype=torch.bfloat16)
    hip.attention_fwd(BF16, qkv, mask, ctx, B, L, H)
    dqkv = torch.empty_like(qkv)
    hip.attention_bwd(BF16, qkv, mask, dctx, dqkv, B, L, H)
    q32 = qkv.float().requires_grad_(True)
    ref = _attn_ref(q32, mask, B, L, H)
    ref.backward(dctx.float())
    assert _rel(ctx, ref.detach()) < 1e-2
    assert _rel(dqkv, q32.grad) < 2e-2
    # against the exact-f32 VALU kernel too (same masks, same definition), which the f32 parity tests tie to the oracle
    ctx32 = torch.empty(B * L, H * 64, device="cuda")
    hip.attention_fwd(F32, qkv.float(), mask, ctx32, B, L, H)
    assert _rel(ctx32, ref.detach()) < 1e-5


def test_attention_mfma_dropout_masks_regenerate_in_backward():
    """Dropout ON (p = 0.1): the bf16 MFMA kernels and the exact-f32 kernels draw the same Philox mask from (seed, site, element index), in
    forward and again in backward. The mask itself is recovered from the f32 kernel (V = identity columns makes ctx = dropped P), then a torch
    reference with that mask must match both bf16 kernels."""
    hip = _hip()
    B, L, H = 6, 30, 4
    g = torch.Generator(device="cuda").manual_seed(77)
    qkv = (torch.randn(B * L, 3 * H * 64, device="cuda", generator=g) * 0.7).bfloat16()
    mask = torch.ones(B, L, dtype=torch.long, device="cuda")
    mask[2, 20:] = 0
    drop = (0.1, 1234567, 9)
    # recover the multipliers: with V[j] = e_j (one-hot over the first L of the 64 head columns) ctx[i][j] = P_dropped[i][j]
    probe = qkv.float().view(B, L, 3, H, 64).clone()
    probe[:, :, 2] = 0
    for j in range(L):
        probe[:, j, 2, :, j] = 1.0
    probe = probe.view(B * L, 3 * H * 64).contiguous()
    pd = torch.empty(B * L, H * 64, device="cuda")
    hip.attention_fwd(F32, probe, mask, pd, B, L, H, drop)
    p0 = torch.empty(B * L, H * 64, device="cuda")
    hip.attention_fwd(F32, probe, mask, p0, B, L, H)
    pd, p0 = pd.view(B, L, H, 64)[..., :L].permute(0, 2, 1, 3), p0.view(B, L, H, 64)[..., :L].permute(0, 2, 1, 3)      # [B][H][i][j]
    mult = torch.where(p0 > 1e-20, pd / p0.clamp_min(1e-30), torch.full_like(p0, 1.0 / 0.9))      # 0 or 1/(1-p)
    keep = (mult > 0.5).float().mean().item()
    assert abs(keep - 0.9) < 0.02 and ((mult - 1 / 0.9).abs() < 1e-3).logical_or(mult.abs() < 1e-6).all()
    dctx = torch.randn(B * L, H * 64, device="cuda", generator=g).bfloat16()
    ctx = torch.empty(B * L, H * 64, device="cuda", dtype=torch.bfloat16)
    hip.attention_fwd(BF16, qkv, mask, ctx, B, L, H, drop)
    dqkv = torch.empty_like(qkv)
    hip.attention_bwd(BF16, qkv, mask, dctx, dqkv, B, L, H, drop)
    q32 = qkv.float().requires_grad_(True)
    ref = _attn_ref(q32, mask, B, L, H, drop_mult=(mult > 0.5).float() / 0.9)
    ref.backward(dctx.float())
    assert _rel(ctx, ref.detach()) < 1e-2
    assert _rel(dqkv, q32.grad) < 2e-2


# the last three shapes have more output tiles than the 512 resident slots of igemm_dma_bn_kernel: the row-range persistent form (several
# tiles per workgroup, the last one partial), at the bench's own sizes — 1024 <- 256 at 14 x 14, 64 <- 256 at 56 x 56 (256-row tiles),
# 512 <- 128 at 28 x 28 — under the automatic policy
@pytest.mark.parametrize("policy,N,H,W,Cc,K,R,st,pad,after",
                         [(p, *shp) for p in (0, 1, 2, 3, 4) for shp in [(4, 14, 14, 256, 256, 3, 1, 1, 0), (2, 28, 28, 128, 512, 1, 1, 0, 1), (8, 7, 7, 512, 2048, 1, 1, 0, 1),
                                                                          (2, 14, 14, 64, 64, 3, 1, 1, 0)]] +
                         [(0, 128, 14, 14, 1024, 256, 1, 1, 0, 1), (0, 127, 56, 56, 64, 256, 1, 1, 0, 0), (0, 125, 28, 28, 512, 128, 1, 1, 0, 1)])
def test_bn_backward_dgrad_epilogue_bf16(policy, N, H, W, Cc, K, R, st, pad, after):
    """conv dgrad with the BatchNorm-backward epilogue in bf16 (every tile family): dz = (dgrad [* relu'(aux)] + residual) [* relu'(aux)] stored
    in bf16, and the two reductions (sum dz, sum dz*(y - mean)) of the stored values. Bound: 4e-3 of max on dz (one bf16 rounding), 5e-3 on the
    reductions."""
    hip = _hip()
    hip.set_tile_policy(policy)
    try:
        cv = hip.conv_desc(BF16, N, H, W, Cc, K, R, R, st, pad)
        g = torch.Generator(device="cuda").manual_seed(H + K + after)
        M = N * H * W
        w = (torch.randn(K, R, R, Cc, device="cuda", generator=g) * 0.05).bfloat16()
        dy = torch.randn(N, cv.Ho, cv.Wo, K, device="cuda", generator=g).bfloat16()
        aux = torch.randn(M, Cc, device="cuda", generator=g).bfloat16()
        res = torch.randn(M, Cc, device="cuda", generator=g).bfloat16()
        y = (torch.randn(M, Cc, device="cuda", generator=g) + 2.0).bfloat16()
        fst = hip.Stats(torch.zeros(8, 3, Cc, device="cuda"), 8, Cc)
        fst.t[:, 0] = y.float().sum(0) / 8
        mean = fst.t[:, 0].sum(0) / M
        dz = torch.empty(M, Cc, device="cuda", dtype=torch.bfloat16)
        dst = hip.Stats(torch.zeros(8, 3, Cc, device="cuda"), 8, Cc)
        hip.conv_dgrad(dy, w, cv, hip.epilogue(dz, Cc, residual=res, dact_aux=aux, dact=hip.DACT_RELU, mask_after_residual=bool(after), colsum=dst,
                                               bn=(y, fst, M)))
        x32 = torch.zeros(N, Cc, H, W, device="cuda", requires_grad=True)
        with torch.backends.cudnn.flags(enabled=False):
            F.conv2d(x32, w.float().permute(0, 3, 1, 2), stride=st, padding=pad).backward(dy.float().permute(0, 3, 1, 2))
        gref = x32.grad.permute(0, 2, 3, 1).reshape(M, Cc)
        msk = (aux.float() > 0).float()
        v = (gref + res.float()) * msk if after else gref * msk + res.float()
        assert _rel(dz, v) < 4e-3
        got = dst.t.sum(0)
        stored = dz.float()
        assert _rel(got[0], stored.sum(0)) < 5e-3 and _rel(got[1], (stored * (y.float() - mean)).sum(0)) < 5e-3
        # the mask as packed bits (what the ResNet executor passes: clite_epilogue.relu_bits, written by bn_apply): bit-identical dz
        bits = torch.from_numpy(np.packbits((aux.float() > 0).cpu().numpy(), axis=-1, bitorder="little")).cuda()
        dz2 = torch.empty_like(dz)
        dst2 = hip.Stats(torch.zeros(8, 3, Cc, device="cuda"), 8, Cc)
        hip.conv_dgrad(dy, w, cv, hip.epilogue(dz2, Cc, residual=res, relu_bits=bits, mask_after_residual=bool(after), colsum=dst2, bn=(y, fst, M)))
        assert torch.equal(dz2, dz)
        assert _rel(dst2.t.sum(0)[:2], got[:2]) < 1e-4
    finally:
        hip.set_tile_policy(0)


@pytest.mark.parametrize("dt", [BF16, F32])
@pytest.mark.parametrize("M,Cc,res", [(6272, 2048, True), (25088, 256, False), (401408, 64, False)])
def test_bn_apply_writes_packed_relu_bits_and_backward_reads_them(dt, M, Cc, res):
    """bn_apply's relu_bits (one bit per element: out > 0) and the BatchNorm backward kernels on that mask against the tensor-mask form:
    the bits equal (out > 0) exactly, reductions agree to summation order, dy / dz are bit-identical."""
    hip = _hip()
    td = hip.TORCH_DTYPE[dt]
    g = torch.Generator(device="cuda").manual_seed(M + Cc)
    y = (torch.randn(M, Cc, device="cuda", generator=g) * 1.5 + 0.3).to(td)
    r = torch.randn(M, Cc, device="cuda", generator=g).to(td) if res else None
    st = hip.Stats(torch.zeros(8, 3, Cc, device="cuda"), 8, Cc)
    st.t[:, 0] = y.float().sum(0) / 8
    st.t[:, 1] = (y.float() ** 2).sum(0) / 8
    gamma, beta = torch.rand(Cc, device="cuda", generator=g) + 0.5, torch.randn(Cc, device="cuda", generator=g) * 0.2
    rm, rv = torch.zeros(Cc, device="cuda"), torch.ones(Cc, device="cuda")
    out = torch.empty_like(y)
    bits = torch.full((M, Cc // 8), 0x55, device="cuda", dtype=torch.uint8)
    hip.bn_apply(dt, hip.bn_desc(M, Cc, st, gamma, beta, rm, rv, True, False, 0.1, 1e-5, True, relu_bits=bits), y, r, out)
    want = torch.from_numpy(np.packbits((out.float() > 0).cpu().numpy(), axis=-1, bitorder="little")).cuda()
    assert torch.equal(bits, want)
    dout = torch.randn(M, Cc, device="cuda", generator=g).to(td)
    d1, d2 = hip.Stats(torch.zeros(8, 3, Cc, device="cuda"), 8, Cc), hip.Stats(torch.zeros(8, 3, Cc, device="cuda"), 8, Cc)
    hip.bn_bwd_reduce(dt, dout, out, y, st, d1, M, Cc)
    hip.bn_bwd_reduce(dt, dout, bits, y, st, d2, M, Cc)
    assert _rel(d2.t.sum(0)[:2], d1.t.sum(0)[:2]) < 1e-5
    desc = hip.bn_desc(M, Cc, st, gamma, beta, rm, rv, True, False, 0.1, 1e-5, False)
    dy1, dz1, dy2, dz2 = (torch.empty_like(y) for _ in range(4))
    hip.bn_bwd_apply(dt, desc, dout, out, y, d1, dy1, dz1, None, None)
    hip.bn_bwd_apply(dt, desc, dout, bits, y, d1, dy2, dz2, None, None)
    assert torch.equal(dy1, dy2) and torch.equal(dz1, dz2)


@pytest.mark.parametrize("N,H,W,Cc,K", [(4, 28, 28, 128, 128), (2, 14, 14, 256, 256), (3, 56, 56, 64, 64)])
def test_stride2_parity_class_dgrad_bf16(N, H, W, Cc, K):
    """clite_conv_dgrad_s2class in bf16: the four input-parity classes of a 3x3 / stride-2 / pad-1 dgrad together equal the full dgrad, each
    element written exactly once. Bound 4e-3 of max (bf16 output)."""
    hip = _hip()
    cv = hip.conv_desc(BF16, N, H, W, Cc, K, 3, 3, 2, 1)
    g = torch.Generator(device="cuda").manual_seed(H + K)
    w = (torch.randn(K, 3, 3, Cc, device="cuda", generator=g) * 0.05).bfloat16()
    dy = torch.randn(N, cv.Ho, cv.Wo, K, device="cuda", generator=g).bfloat16()
    dx = torch.full((N * H * W, Cc), 7.0, device="cuda", dtype=torch.bfloat16)
    assert hip.s2_classes_ok(cv)
    hip.conv_dgrad_s2(dy, w, cv, lambda: hip.epilogue(dx, Cc))
    x32 = torch.zeros(N, Cc, H, W, device="cuda", requires_grad=True)
    with torch.backends.cudnn.flags(enabled=False):
        F.conv2d(x32, w.float().permute(0, 3, 1, 2), stride=2, padding=1).backward(dy.float().permute(0, 3, 1, 2))
    assert _rel(dx, x32.grad.permute(0, 2, 3, 1).reshape(N * H * W, Cc)) < 4e-3


@pytest.mark.parametrize("dt", [BF16, F32])
@pytest.mark.parametrize("M,Cc", [(3840, 768), (128, 2048), (77, 512)])
def test_layernorm_backward_fused_bias_gradient_and_regenerated_masks(dt, M, Cc):
    """clite_layernorm_bwd: dx, dgamma, dbeta and the fused column sums (`dcolsum` = bias gradient of the nn.Linear in front of the LayerNorm)
    against torch, with BOTH dropout masks on: the input mask (dropout after this LayerNorm in forward) must equal the one clite_layernorm_fwd
    drew — it is recovered from the forward output — and the output mask (dropout in front of the producer's residual add) is recovered from
    dx_masked / dx. Bounds: f32 1e-4; bf16 1e-2 of max for tensors, 2e-2 for the column reductions (sums of bf16-rounded rows)."""
    hip = _hip()
    td = torch.bfloat16 if dt == BF16 else torch.float32
    g = torch.Generator(device="cuda").manual_seed(M + Cc)
    x = torch.randn(M, Cc, device="cuda", generator=g).to(td)
    gamma = torch.rand(Cc, device="cuda", generator=g) + 0.5
    beta = torch.randn(Cc, device="cuda", generator=g) * 0.1
    dy = torch.randn(M, Cc, device="cuda", generator=g).to(td)
    p, d_in, d_out = 0.1, (0.1, 4242, 3), (0.1, 4242, 5)
    out, st = torch.empty_like(x), torch.empty(M, 2, device="cuda")
    hip.layernorm_fwd(dt, x, gamma, beta, 1e-12, out, st, M, Cc, d_in)
    x32 = x.float().requires_grad_(True)
    g32, b32 = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ln = F.layer_norm(x32, (Cc,), g32, b32, 1e-12)
    m_in = (out.float() != 0).float() / (1 - p)                  # LN output is never exactly 0 on random data
    assert abs((m_in > 0).float().mean().item() - 0.9) < 0.01
    assert _rel(out, ln.detach() * m_in) < (1e-2 if dt == BF16 else 1e-5)
    dx, dxm = torch.empty_like(x), torch.empty_like(x)
    dgamma, dbeta, dcol = torch.zeros(Cc, device="cuda"), torch.zeros(Cc, device="cuda"), torch.zeros(Cc, device="cuda")
    hip.layernorm_bwd(dt, dy, x, st, gamma, dx, dxm, dgamma, dbeta, M, Cc, d_in, d_out, dcolsum=dcol)
    (ln * m_in).backward(dy.float())
    tol = 1e-2 if dt == BF16 else 1e-4
    assert _rel(dx, x32.grad) < tol
    assert _rel(dgamma, g32.grad) < 2 * tol and _rel(dbeta, b32.grad) < 2 * tol
    m_out = torch.where(dx.float().abs() > 1e-6, dxm.float() / dx.float(), torch.full_like(dx.float(), 1 / (1 - p)))
    frac = (m_out.abs() > 0.5).float().mean().item()
    assert abs(frac - 0.9) < 0.01 and ((m_out - 1 / (1 - p)).abs() < 2e-2).logical_or(m_out.abs() < 1e-6).float().mean().item() > 0.999
    assert not torch.equal(m_out > 0.5, m_in > 0)                # two sites, two masks
    assert _rel(dcol, dxm.float().sum(0)) < (2e-2 if dt == BF16 else 1e-4)


@pytest.mark.parametrize("dt", [BF16, F32])
def test_embedding_backward_merges_runs_and_skips_padding(dt):
    """clite_embed_bwd: word-embedding rows with many adders — every caption's [CLS] (position 0) and [SEP], repeated tokens inside a batch
    segment, the [PAD] row (padding_idx = 0: no gradient) — against torch index_add; position gradients are batch sums."""
    hip = _hip()
    B, L, Cc, V = 37, 30, 768, 5000
    td = torch.bfloat16 if dt == BF16 else torch.float32
    g = torch.Generator(device="cuda").manual_seed(11)
    ids = torch.randint(1, 60, (B, L), device="cuda", generator=g)                # small vocabulary range: many repeats
    ids[:, 0], ids[:, L - 1] = 101, 102
    ids[3, 20:], ids[9, 5:] = 0, 0                                                # right-padded captions
    d = torch.randn(B * L, Cc, device="cuda", generator=g).to(td)
    dword, dpos = torch.zeros(V, Cc, device="cuda"), torch.zeros(512, Cc, device="cuda")
    hip.embed_bwd(dt, ids.view(-1), d, dword, dpos, B * L, L, Cc, V, 0)
    ref_w = torch.zeros(V, Cc, device="cuda").index_add_(0, ids.view(-1), d.float())
    ref_w[0] = 0
    ref_p = d.float().view(B, L, Cc).sum(0)
    assert _rel(dword, ref_w) < 1e-5 and not dword[0].any()
    assert _rel(dpos[:L], ref_p) < 1e-5 and not dpos[L:].any()


def test_resnet50_bert_bf16_backward_against_fp32_oracle():
    """Whole-model bf16 BACKWARD at a conditioned size: ResNet-50 + 2-layer BERT + JSD heads, batch 64, 128 x 128 images, 30 tokens, the
    reference's default initialisation with the residual-branch BatchNorm gains (bn3.weight) scaled by 0.1, dropout off, prior noise pinned —
    gradients of the bf16 HIP path against the fp32 oracle on the CPU.
    Why the scaling: at the unscaled default init a 50-layer train-mode-BatchNorm network amplifies the 2^-9 storage roundings until bf16 and
    fp32 gradients decorrelate — tools/diag_bf16_cond.py on MI355X (profiles/r2_bf16_conditioning.txt): image-encoder cosine 0.13 at gain 1,
    0.52 at 0.5, 0.85 at 0.25, 0.95 at 0.1, 0.97 at 0 (PyTorch itself with emulated bf16 storage does the same: test_gpu_model.py). Small
    residual gains (the usual zero-init-residual recipe) are the conditioned version of the same problem.
    Bounds, asserted below (observed in that study: loss 2e-3, cosines 0.950 / 0.961 / 0.985, worst weight relative L2 0.34):
    loss within 1e-2; cosine between the bf16 and fp32 gradient vectors of each top-level module >= 0.93; relative L2 error <= 0.5 for every
    weight tensor with more than 4096 elements."""
    from detfill import det_tensor
    from oracle import ref_model as O
    from clip_lite_amd.encoder import ImageEncoder, TextEncoder
    from clip_lite_amd.loss import JSDInfoMaxLoss
    from clip_lite_amd.model import VLInfoModel
    torch.manual_seed(21)
    B, S, L = 64, 128, 30
    Mo = O.build_oracle_model("resnet50", "train_sbert", 2, dropout=0.0).train()
    with torch.no_grad():
        for n, p in Mo.named_parameters():
            if n.endswith("bn3.weight"):
                p.mul_(0.1)
    te = TextEncoder(mode="train_sbert", num_hidden_layers=2)
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
    torch.set_num_threads(16)
    ref = Mo(batch)
    ref["loss"].backward()
    assert abs(out["loss"].item() - ref["loss"].item()) < 1e-2, (out["loss"].item(), ref["loss"].item())
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
    cos = {k: v[0] / (v[1] * v[2]) ** 0.5 for k, v in per_top.items()}
    print("worst weight-gradient relative L2:", worst, cos)
    assert worst[1] <= 0.5, worst
    assert all(c >= 0.93 for c in cos.values()), cos


# the four conv3 -> bn3 shapes of ResNet-50 at batch 128: rows, K = 4 x width, Cin = width; + a ragged small one (partial tiles)
@pytest.mark.parametrize("policy,M,K,Cin", [(0, 401408, 256, 64), (0, 100352, 512, 128), (0, 25088, 1024, 256), (0, 6272, 2048, 512), (0, 3000, 128, 64),
                                            (4, 100352, 512, 128), (2, 25088, 1024, 256), (1, 3000, 128, 128)])
def test_folded_batchnorm_backward_matches_torch_bn_backward_then_conv_backward(policy, M, K, Cin):
    """The block-output BatchNorm backward folded into conv3's two gradients (include/clite.h ABI v12: clite_bn_fold_prepare, clite_conv_dgrad_bnfold,
    clite_wgrad_item.row_scale, clite_bn.out_sum, clite_bn_fold_wgrad_finish) against a plain torch fp32 evaluation of batch_norm backward followed by the 1 x 1 convolution's
    backward on the same bf16-rounded tensors: the masked input gradient dz2 and its two reductions (the next BatchNorm backward's), the weight gradient,
    dgamma / dbeta. Bars: 4e-3 of max on dz2 (one bf16 rounding on top of the existing 2e-3 bar), 5e-3 on its reductions, 2e-3 of max on the weight gradient
    and on dgamma / dbeta."""
    hip = _hip()
    hip.set_tile_policy(policy)
    try:
        g = torch.Generator(device="cuda").manual_seed(M + K)
        a = torch.relu(torch.randn(M, Cin, device="cuda", generator=g) + 0.3).bfloat16()          # conv3's input: post-ReLU activation
        W = (torch.randn(K, Cin, device="cuda", generator=g) * (2.0 / Cin) ** 0.5).bfloat16()
        y = (a.float() @ W.float().t()).bfloat16()                                                 # what conv3's forward stored
        dz = (torch.randn(M, K, device="cuda", generator=g) * 0.1 + 0.02 * torch.randn(K, device="cuda", generator=g)).bfloat16()
        dz = dz * (torch.rand(M, K, device="cuda", generator=g) > 0.4)                              # masked by the block output's relu'
        gamma = torch.rand(K, device="cuda", generator=g) + 0.5
        R = 2
        st = hip.Stats(torch.zeros(R, 3, K, device="cuda"), R, K)
        st.t[:, 0] = y.float().sum(0) / R
        st.t[:, 1] = (y.float() ** 2).sum(0) / R
        mean = y.float().mean(0)
        var = ((y.float() ** 2).mean(0) - mean * mean).clamp_min(0)
        rstd = torch.rsqrt(var + 1e-5)
        pre = hip.Stats(torch.zeros(R, 3, K, device="cuda"), R, K)
        S1, S2 = dz.float().sum(0), (dz.float() * (y.float() - mean)).sum(0)
        pre.t[:, 0], pre.t[:, 1] = S1 / R, S2 / R
        # reference (fp32): BatchNorm backward, then the convolution's two gradients
        xhat = (y.float() - mean) * rstd
        dy = gamma * rstd * (dz.float() - S1 / M - xhat * (dz.float() * xhat).sum(0) / M)
        da_ref = dy @ W.float()
        dW_ref = dy.t() @ a.float()
        dgamma_ref, dbeta_ref = (dz.float() * xhat).sum(0), S1
        # the unit in front (bn2): its input, forward sums and relu' bits
        y2 = (torch.randn(M, Cin, device="cuda", generator=g) + 1.0).bfloat16()
        st2 = hip.Stats(torch.zeros(R, 3, Cin, device="cuda"), R, Cin)
        st2.t[:, 0] = y2.float().sum(0) / R
        mean2 = y2.float().mean(0)
        mask = torch.rand(M, Cin, device="cuda", generator=g) > 0.3
        bits = torch.from_numpy(np.packbits(mask.cpu().numpy(), axis=-1, bitorder="little")).cuda()
        # HIP
        pair = torch.empty(2, M, K, device="cuda", dtype=torch.bfloat16)
        pair[0], pair[1] = dz, y
        rm, rv = torch.zeros(K, device="cuda"), torch.ones(K, device="cuda")
        dgamma, dbeta = torch.zeros(K, device="cuda"), torch.zeros(K, device="cuda")
        desc = hip.bn_desc(M, K, st, gamma, torch.zeros(K, device="cuda"), rm, rv, True, False, 0.1, 1e-5, False)
        f = hip.bn_fold_prepare(desc, pre, W.t().contiguous(), Cin, dgamma, dbeta)
        dz2 = torch.empty(M, Cin, device="cuda", dtype=torch.bfloat16)
        d2 = hip.Stats(torch.zeros(R, 3, Cin, device="cuda"), R, Cin)
        hip.conv_dgrad_bnfold(pair, f.w2, M, K, Cin, hip.epilogue(dz2, Cin, relu_bits=bits, colsum=d2, bn=(y2, st2, M), bias=f.bias))
        want = da_ref * mask
        assert _rel(dz2, want) < 4e-3
        stored = dz2.float()
        got = d2.t.sum(0)
        assert _rel(got[0], stored.sum(0)) < 5e-3 and _rel(got[1], (stored * (y2.float() - mean2)).sum(0)) < 5e-3
        assert _rel(dgamma, dgamma_ref) < 2e-3 and _rel(dbeta, dbeta_ref) < 2e-3
        # weight gradient: the grouped launch (ka . dz^T a, the Gram matrix) + the two correction terms
        dw = torch.zeros(K, Cin, device="cuda")
        G = torch.zeros(Cin, Cin, device="cuda")
        # colsum(a) as clite_bn_apply leaves it (clite_bn.out_sum): an identity BatchNorm + ReLU over a reproduces a and sums its columns
        asum = hip.Stats(torch.zeros(R, 3, Cin, device="cuda"), R, Cin)
        ist = hip.Stats(torch.zeros(R, 3, Cin, device="cuda"), R, Cin)
        ist.t[:, 1] = M * (1.0 - 1e-5) / R          # mean 0, var + eps = 1
        a_out = torch.empty_like(a)
        one, zero = torch.ones(Cin, device="cuda"), torch.zeros(Cin, device="cuda")
        hip.bn_apply(BF16, hip.bn_desc(M, Cin, ist, one, zero, zero.clone(), one.clone(), True, False, 0.1, 1e-5, True, out_sum=asum), a, None, a_out)
        assert torch.equal(a_out, a)
        assert _rel(asum.t[:, 0].sum(0), a.float().sum(0)) < 1e-4
        cv = hip.conv_desc(BF16, 1, 1, M, Cin, K, 1, 1, 1, 0)
        grp = hip.WgradGroup(BF16)
        grp.conv(dz, a, cv, dw, row_scale=f.coef[0])
        grp.conv(a, a, hip.conv_desc(BF16, 1, 1, M, Cin, Cin, 1, 1, 1, 0), G)
        Wt = W.t().contiguous()
        grp.after(lambda: hip.bn_fold_wgrad_finish(G, asum, f.coef, Wt, M, K, Cin, dw))
        grp.launch()
        torch.cuda.synchronize()
        assert _rel(G, a.float().t() @ a.float()) < 1e-3
        assert _rel(dw, dW_ref) < 2e-3, _rel(dw, dW_ref)
    finally:
        hip.set_tile_policy(0)


@pytest.mark.parametrize("B,Fin,U", [(128, 2048, 2048), (128, 768, 2048), (6, 512, 2048), (37, 768, 256)])
def test_fused_mi_block_matches_torch_fp32(B, Fin, U):
    """clite_mi_block_fwd1/2 + _bwd1/2 (ABI v12; csrc/heads_fused.hip) through loss.mi_block_forward / mi_block_backward: the MI projection block of reference
    loss.py:12-40 — LayerNorm(W2 relu(BatchNorm1d(W1 x)) + b2 + Ws x + bs) — in bf16 against a plain torch fp32 evaluation on the bf16-rounded weights and
    input: output within 2e-2 of max (the block is three bf16 roundings deep), input gradient within 3e-2, BatchNorm1d running statistics after the
    reference's TWO updates per step within 5e-3 / 2e-3, and every parameter gradient — weights through the grouped launch, BatchNorm1d / LayerNorm / both
    bias gradients from the fused kernels — cosine >= 0.999 against autograd. Also: the fused path really ran (no clite_bn_apply launch)."""
    from clip_lite_amd import hip as H
    from clip_lite_amd.loss import MILinearBlock, mi_block_backward, mi_block_forward
    from clip_lite_amd.runtime import DeviceRuntime
    torch.manual_seed(B + Fin)

    class Holder(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.loss = torch.nn.Module()
            self.loss.blk = MILinearBlock(Fin, U)
    Mh = Holder()
    with torch.no_grad():
        Mh.loss.blk.feature_shortcut.weight.copy_(torch.randn(U, Fin) * 0.03)
        Mh.loss.blk.feature_nonlinear[1].weight.copy_(torch.rand(U) + 0.5)
        Mh.loss.blk.feature_nonlinear[1].bias.copy_(torch.randn(U) * 0.1)
    Mh.cuda()
    rt = DeviceRuntime(Mh, torch.device("cuda", torch.cuda.current_device()), lowp=True)
    blk = Mh.loss.blk
    assert rt.fused_heads
    x = torch.randn(B, Fin, device="cuda").bfloat16()
    dout = (torch.randn(B, U, device="cuda") * 0.1).bfloat16()
    dres = (torch.randn(B, Fin, device="cuda") * 0.1).bfloat16()
    calls = []
    orig = H.bn_apply
    H.bn_apply = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        out, ctx = mi_block_forward(rt, blk, x, True, updates=2)
    finally:
        H.bn_apply = orig
    assert not calls
    grp = H.WgradGroup(H.BF16)
    dx = mi_block_backward(rt, blk, ctx, dout, dres, defer=grp)
    grp.launch()
    torch.cuda.synchronize()
    # reference: torch modules in fp32 on the bf16-rounded weights
    l1, bn, _, l2 = blk.feature_nonlinear
    r = lambda p_: p_.detach().float().bfloat16().float().requires_grad_(True)
    w1, w2, ws = r(l1.weight), r(l2.weight), r(blk.feature_shortcut.weight)
    g_, b_ = bn.weight.detach().clone().requires_grad_(True), bn.bias.detach().clone().requires_grad_(True)
    b2, bs = l2.bias.detach().clone().requires_grad_(True), blk.feature_shortcut.bias.detach().clone().requires_grad_(True)
    lw, lb = blk.feature_block_ln.weight.detach().clone().requires_grad_(True), blk.feature_block_ln.bias.detach().clone().requires_grad_(True)
    xf = x.float().requires_grad_(True)
    z = xf @ w1.t()
    mean, var = z.mean(0), z.var(0, unbiased=False)
    a = torch.relu((z - mean) * torch.rsqrt(var + bn.eps) * g_ + b_)
    t = a @ w2.t() + b2 + xf @ ws.t() + bs
    ref = F.layer_norm(t, (U,), lw, lb, 1e-5)
    ref.backward(dout.float())
    assert _rel(out, ref.detach()) < 2e-2
    assert _rel(dx, xf.grad + dres.float()) < 3e-2
    unb = var.detach() * B / max(B - 1, 1)
    rm, rv = torch.zeros(U, device="cuda"), torch.ones(U, device="cuda")
    for _ in range(2):
        rm, rv = 0.9 * rm + 0.1 * mean.detach(), 0.9 * rv + 0.1 * unb
    assert _rel(bn.running_mean, rm) < 5e-3 and _rel(bn.running_var, rv) < 2e-3          # (statistics of the bf16-STORED z, like the unfused path's)
    A = rt.arena
    for name, p_, want in (("w1", l1.weight, w1.grad), ("w2", l2.weight, w2.grad), ("ws", blk.feature_shortcut.weight, ws.grad), ("gamma", bn.weight, g_.grad),
                           ("beta", bn.bias, b_.grad), ("b2", l2.bias, b2.grad), ("bs", blk.feature_shortcut.bias, bs.grad),
                           ("ln.w", blk.feature_block_ln.weight, lw.grad), ("ln.b", blk.feature_block_ln.bias, lb.grad)):
        got = A.g(p_).flatten().float()
        cos = (got @ want.flatten() / (got.norm() * want.norm()).clamp_min(1e-20)).item()
        assert cos >= 0.999, (name, cos)
        assert abs(got.norm().item() / want.norm().item() - 1) < 2e-2, (name, got.norm().item(), want.norm().item())
