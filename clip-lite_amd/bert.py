"""BERT text encoder on the HIP kernels: module tree with transformers.BertModel's parameter names (what reference
encoder.py:165-170 builds with BertConfig(num_hidden_layers=n)) and the hand-scheduled forward/backward executor.

Per layer: one fused QKV GEMM ([3*768] outputs; q/k/v weights are adjacent in the parameter arena), per-head attention
kernel (L <= 32), output projection with bias + dropout + residual in the GEMM epilogue, LayerNorm, FFN with the GELU
(and its saved pre-activation) in the epilogue, second LayerNorm. Dropout masks are regenerated from (seed, site, index).
"""
import torch
import torch.nn as nn

from . import hip


class LinearParams(nn.Module):
    def __init__(self, cin, cout, bias=True, std=None):
        super().__init__()
        self.in_features, self.out_features = cin, cout
        self.weight = nn.Parameter(torch.empty(cout, cin))
        self.bias = nn.Parameter(torch.zeros(cout)) if bias else None
        if std is None:                      # nn.Linear default init (reference loss.py heads)
            nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)
            if bias:
                bound = 1 / cin ** 0.5
                nn.init.uniform_(self.bias, -bound, bound)
        else:                                # BertPreTrainedModel._init_weights
            self.weight.data.normal_(0.0, std)


class LayerNormParams(nn.Module):
    def __init__(self, c, eps):
        super().__init__()
        self.normalized_shape, self.eps = (c,), eps
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))


class EmbeddingParams(nn.Module):
    def __init__(self, n, c, std=0.02, padding_idx=None):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(n, c).normal_(0.0, std))
        if padding_idx is not None:
            self.weight.data[padding_idx].zero_()


class _SelfAttention(nn.Module):
    def __init__(self, h):
        super().__init__()
        self.query, self.key, self.value = LinearParams(h, h, std=0.02), LinearParams(h, h, std=0.02), LinearParams(h, h, std=0.02)


class _SelfOutput(nn.Module):
    def __init__(self, cin, h, eps):
        super().__init__()
        self.dense = LinearParams(cin, h, std=0.02)
        self.LayerNorm = LayerNormParams(h, eps)


class _Attention(nn.Module):
    def __init__(self, h, eps):
        super().__init__()
        self.self = _SelfAttention(h)
        self.output = _SelfOutput(h, h, eps)


class _Intermediate(nn.Module):
    def __init__(self, h, inner):
        super().__init__()
        self.dense = LinearParams(h, inner, std=0.02)


class BertLayer(nn.Module):
    def __init__(self, h, inner, eps):
        super().__init__()
        self.attention = _Attention(h, eps)
        self.intermediate = _Intermediate(h, inner)
        self.output = _SelfOutput(inner, h, eps)


class _Embeddings(nn.Module):
    def __init__(self, vocab, h, max_pos, eps):
        super().__init__()
        self.word_embeddings = EmbeddingParams(vocab, h, padding_idx=0)
        self.position_embeddings = EmbeddingParams(max_pos, h)
        self.token_type_embeddings = EmbeddingParams(2, h)
        self.LayerNorm = LayerNormParams(h, eps)


class _Encoder(nn.Module):
    def __init__(self, n, h, inner, eps):
        super().__init__()
        self.layer = nn.ModuleList([BertLayer(h, inner, eps) for _ in range(n)])


class _Pooler(nn.Module):
    def __init__(self, h):
        super().__init__()
        self.dense = LinearParams(h, h, std=0.02)


class BertModel(nn.Module):
    """BertConfig defaults (transformers): hidden 768, 12 heads, intermediate 3072, vocab 30522, 512 positions, GELU(erf),
    LayerNorm eps 1e-12, hidden/attention dropout 0.1."""

    def __init__(self, num_hidden_layers=12, hidden=768, heads=12, inner=3072, vocab=30522, max_pos=512, dropout=0.1, eps=1e-12):
        super().__init__()
        assert hidden == heads * 64, "the attention kernel is built for head size 64"
        self.hidden, self.heads, self.inner, self.vocab, self.max_pos = hidden, heads, inner, vocab, max_pos
        self.hidden_dropout_prob = self.attention_probs_dropout_prob = dropout
        self.embeddings = _Embeddings(vocab, hidden, max_pos, eps)
        self.encoder = _Encoder(num_hidden_layers, hidden, inner, eps)
        self.pooler = _Pooler(hidden)

    def contiguous_groups(self, prefix):
        """q/k/v weights (and biases) must be adjacent in the arena so one GEMM computes all three projections."""
        groups = []
        for i in range(len(self.encoder.layer)):
            base = f"{prefix}encoder.layer.{i}.attention.self."
            groups.append([base + "query.weight", base + "key.weight", base + "value.weight"])
            groups.append([base + "query.bias", base + "key.bias", base + "value.bias"])
        return groups


# ---------------------------------------------------------------------------------------------------- executor
def _alloc(rt, *shape):
    return torch.empty(shape, device=rt.device, dtype=rt.tdtype)


def _linear_grads(rt, lin, dy, x, M, dw=None, db=None, bias_done=False, defer=None):
    """dW[N][K] += dy^T x ; db += colsum(dy) for a LinearParams (or explicit arena views for fused q/k/v). bias_done: the kernel that
    produced dy already accumulated its column sums into the bias gradient (clite_layernorm_bwd's dcolsum). defer: a hip.WgradGroup that
    collects the launches instead (bert_backward)."""
    N, K = dy.shape[1], x.shape[1]
    if dw is None:
        dw = rt.arena.g(lin.weight) if lin.weight.requires_grad else None
        db = rt.arena.g(lin.bias) if (lin.bias is not None and lin.bias.requires_grad and not bias_done) else None
    if defer is not None:
        if dw is not None:
            defer.linear(dy, x, N, K, M, dw)
        if db is not None:
            defer.call(lambda: hip.colsum(rt.dt, dy, db, M, N))
        return

    def launch():
        if dw is not None:
            hip.gemm_tn(rt.dt, dy, x, N, K, M, hip.epilogue(dw, K, atomic=True, out_f32=True))
        if db is not None:
            hip.colsum(rt.dt, dy, db, M, N)
    rt.aux_launch(launch, dy, x)       # off the dgrad chain: runtime.DeviceRuntime.aux_launch


def bert_forward(rt, net, input_ids, attention_mask, step):
    """input_ids/attention_mask: int64 [B][L] on the device, L <= 32. Returns (pooler_output [B][H], ctx)."""
    B, L = input_ids.shape
    if L > 32:
        raise RuntimeError("clip_lite_amd: the attention kernel supports captions of at most 32 tokens (reference config.py:69: 30)")
    dt, A = rt.dt, rt.arena
    Hd, heads, inner = net.hidden, net.heads, net.inner
    M = B * L
    training = step.training
    p_h = net.hidden_dropout_prob if training else 0.0
    p_a = net.attention_probs_dropout_prob if training else 0.0
    drop = lambda p: (p, step.seed, step.site()) if p > 0 else hip.NO_DROP
    ids = input_ids.contiguous()
    mask = attention_mask.contiguous().to(torch.int64)
    emb = net.embeddings
    s0 = _alloc(rt, M, Hd)
    hip.embed_fwd(dt, ids, A.w(emb.word_embeddings.weight), A.w(emb.position_embeddings.weight), A.w(emb.token_type_embeddings.weight),
                  s0, M, L, Hd, net.vocab)
    # fp8 forward (BASELINE configs[4]; fp8.Fp8Text): QKV, FFN1 and FFN2 read e4m3 copies their producers wrote (LayerNorm forward, FFN1's epilogue)
    from .fp8 import text_state
    f8 = text_state(rt, net)
    if f8 is not None:
        f8.begin_step()
    nl = len(net.encoder.layer)
    prod = (lambda l, j, Cc: f8.producer(l, j, M, Cc, training)) if f8 is not None else (lambda l, j, Cc: (None, None))

    def linear(x, x8, w, w8, Mr, N, K, ep):
        """x [Mr][K] @ w[N][K]^T through the fused epilogue: bf16 MFMA, or OCP e4m3 operands when the producer of x left its e4m3 copy x8."""
        if x8 is not None:
            hip.gemm_nt_fp8(x8, w8(), Mr, N, K, ep)
        else:
            hip.gemm_nt(dt, x, w, Mr, N, K, ep)

    h = _alloc(rt, M, Hd)
    st0 = torch.empty(M, 2, device=rt.device, dtype=torch.float32)
    d0 = drop(p_h)
    q, h8 = prod(0, 0, Hd) if nl else (None, None)
    hip.layernorm_fwd(dt, s0, emb.LayerNorm.weight, emb.LayerNorm.bias, emb.LayerNorm.eps, h, st0, M, Hd, d0, fp8=q)
    ctx = {"B": B, "L": L, "ids": ids, "mask": mask, "s0": s0, "st0": st0, "d0": d0, "layers": []}

    for l, layer in enumerate(net.encoder.layer):
        sa, so = layer.attention.self, layer.attention.output
        wqkv = A.span([sa.query.weight, sa.key.weight, sa.value.weight])
        bqkv = A.span([sa.query.bias, sa.key.bias, sa.value.bias], lowp=False)
        qkv = _alloc(rt, M, 3 * Hd)
        linear(h, h8, wqkv, lambda: f8.weight(l, 0, (3 * Hd, Hd)), M, 3 * Hd, Hd, hip.epilogue(qkv, 3 * Hd, bias=bqkv))
        ctxt = _alloc(rt, M, Hd)
        da = drop(p_a)
        hip.attention_fwd(dt, qkv, mask, ctxt, B, L, heads, da)
        s1 = _alloc(rt, M, Hd)
        d1 = drop(p_h)
        hip.gemm_nt(dt, ctxt, A.w(so.dense.weight), M, Hd, Hd, hip.epilogue(s1, Hd, bias=so.dense.bias, drop=d1, residual=h))
        h1 = _alloc(rt, M, Hd)
        st1 = torch.empty(M, 2, device=rt.device, dtype=torch.float32)
        q, h18 = prod(l, 1, Hd)
        hip.layernorm_fwd(dt, s1, so.LayerNorm.weight, so.LayerNorm.bias, so.LayerNorm.eps, h1, st1, M, Hd, fp8=q)
        f = _alloc(rt, M, inner)      # FFN pre-activation (kept for GELU')
        g = _alloc(rt, M, inner)
        q, g8 = prod(l, 2, inner)
        if h18 is None and q is not None:
            # FFN1 itself still runs in bf16 (no scale for its input yet): its GELU output's amax / e4m3 copy come from the stand-alone quantiser's
            # path next step - record nothing here (the bf16 launch has no fused quantiser), so that slot becomes ready one step after h1's
            f8._seen.discard(3 * l + 2)
            q, g8 = None, None
        linear(h1, h18, A.w(layer.intermediate.dense.weight), lambda: f8.weight(l, 1, (inner, Hd)), M, inner, Hd,
               hip.epilogue(g, inner, bias=layer.intermediate.dense.bias, act=hip.ACT_GELU, preact=f, fp8=q))
        s2 = _alloc(rt, M, Hd)
        d2 = drop(p_h)
        linear(g, g8, A.w(layer.output.dense.weight), lambda: f8.weight(l, 2, (Hd, inner)), M, Hd, inner,
               hip.epilogue(s2, Hd, bias=layer.output.dense.bias, drop=d2, residual=h1))
        h2 = _alloc(rt, M, Hd)
        st2 = torch.empty(M, 2, device=rt.device, dtype=torch.float32)
        q, h28 = prod(l + 1, 0, Hd) if l + 1 < nl else (None, None)
        hip.layernorm_fwd(dt, s2, layer.output.LayerNorm.weight, layer.output.LayerNorm.bias, layer.output.LayerNorm.eps, h2, st2, M, Hd, fp8=q)
        ctx["layers"].append((layer, h, qkv, ctxt, da, s1, d1, st1, h1, f, g, s2, d2, st2))
        h, h8 = h2, h28
    if f8 is not None and training:          # (an eval forward leaves the delayed-scaling state alone)
        f8.end_step()
    pooled = _alloc(rt, B, Hd)
    hip.gemm_nt(dt, h, A.w(net.pooler.dense.weight), B, Hd, Hd, hip.epilogue(pooled, Hd, bias=net.pooler.dense.bias, act=hip.ACT_TANH, ws=rt.gemm_ws(B, Hd)), lda=L * Hd)
    ctx["h_last"], ctx["pooled"] = h, pooled
    return pooled, ctx


def _dgrad(rt, x, M, N, K, ep, *ws):
    """Input gradient of a Linear (or of BERT's fused q/k/v projection: `ws` = the adjacent weights): x[M][K] @ W[K][N]. bf16 mode reads the
    transposed copy W^T[N][K] (Arena.wt), so the GEMM runs in the forward form — both operands k-contiguous — instead of staging the weight as a
    k-strided image (3840 x 768 x 3072: 31 vs 41 us isolated); the exact-f32 mode keeps clite_gemm_nn."""
    A = rt.arena
    if rt.transposed_dgrad and A.has_wt(*ws):
        hip.gemm_nt(rt.dt, x, A.wt(*ws), M, N, K, ep)
    else:
        hip.gemm_nn(rt.dt, x, A.w(ws[0]) if len(ws) == 1 else A.span(list(ws)), M, N, K, ep)


def segment_layers(n_layers, seg):
    """(lo, hi) layer indices of backward segment `seg` = (i, n): the i-th of n runs of consecutive layers, walked from the top (i = 0 holds the last layers)."""
    i, n = seg
    bounds = [round(n_layers * j / n) for j in range(n + 1)]
    return bounds[n - i - 1], bounds[n - i]


def bert_backward(rt, net, ctx, dpooled, defer=None, seg=None):
    """defer: a hip.WgradGroup — the 4 x 12 + 1 linear weight gradients are collected and left to the caller to launch (as one grouped launch)
    instead of being enqueued one by one between the input-gradient GEMMs.
    seg = (i, n): only the i-th of n chain segments (segment_layers; 0 = pooler + the last layers, n - 1 = the first layers + the embeddings), the
    hidden-state gradient parked in ctx between calls — the captured data-parallel step records one graph and one weight-gradient group per segment, so
    that each segment's span of the gradient arena is handed to the exchange while the next one still runs (reference train.py:174-178: DDP's buckets
    fill in reverse order of the forward)."""
    dt, A = rt.dt, rt.arena
    B, L = ctx["B"], ctx["L"]
    Hd, heads, inner = net.hidden, net.heads, net.inner
    M = B * L
    nl = len(ctx["layers"])
    lo_layer, hi_layer = (0, nl) if seg is None else segment_layers(nl, seg)
    first, last = seg is None or seg[0] == 0, seg is None or seg[0] == seg[1] - 1
    own_group = None
    if seg is None and defer is None and rt.group_wgrad and not rt.overlap_wgrad and not rt._capturing:      # (a capture cannot allocate the pinned staging)
        defer = own_group = hip.WgradGroup(rt.dt)          # uncaptured backward: grouped launch at the end of this call
    staged, pooler_done = own_group is not None and getattr(rt, "exchange", None) is not None, False
    A.ensure_transposed(capturing=rt._capturing)
    if first:
        # pooler: dpre = dpooled * (1 - y^2); h[:, 0] rows only
        dpre = _alloc(rt, B, Hd)
        hip.tanh_bwd(dt, dpooled, ctx["pooled"], dpre, B * Hd)
        pw = net.pooler.dense
        if pw.weight.requires_grad:
            if defer is not None:
                defer.linear(dpre, ctx["h_last"], Hd, Hd, B, A.g(pw.weight), ldb=L * Hd)
            else:
                hip.gemm_tn(dt, dpre, ctx["h_last"], Hd, Hd, B, hip.epilogue(A.g(pw.weight), Hd, atomic=True, out_f32=True), ldb=L * Hd)
            hip.colsum(dt, dpre, A.g(pw.bias), B, Hd)
        if own_group is None:
            rt.grads_ready(net.pooler)
        dh = torch.zeros(M, Hd, device=rt.device, dtype=rt.tdtype)
        _dgrad(rt, dpre, B, Hd, Hd, hip.epilogue(dh, L * Hd, ws=rt.gemm_ws(B, Hd)), pw.weight)
    else:
        dh = ctx.pop("bwd_dh")
    for (layer, h, qkv, ctxt, da, s1, d1, st1, h1, f, g, s2, d2, st2) in reversed(ctx["layers"][lo_layer:hi_layer]):
        sa, so, out = layer.attention.self, layer.attention.output, layer.output
        # LayerNorm 2 -> (dropout) -> FFN
        ds2 = _alloc(rt, M, Hd)
        ds2m = _alloc(rt, M, Hd) if d2[0] > 0 else None
        hip.layernorm_bwd(dt, dh, s2, st2, out.LayerNorm.weight, ds2, ds2m, A.g(out.LayerNorm.weight), A.g(out.LayerNorm.bias), M, Hd, drop_out=d2,
                          dcolsum=A.g(out.dense.bias))
        dz2 = ds2m if ds2m is not None else ds2
        _linear_grads(rt, out.dense, dz2, g, M, bias_done=True, defer=defer)
        df = _alloc(rt, M, inner)
        # (the bias gradient of the FFN's first Linear = the column sums of df: accumulated by the epilogue that stores df, clite_epilogue.colsum
        # with colsum_rows = 1, instead of a pass over the 23 MB tensor per layer)
        ib = layer.intermediate.dense.bias
        fold = ib is not None and ib.requires_grad
        _dgrad(rt, dz2, M, inner, Hd, hip.epilogue(df, inner, dact_aux=f, dact=hip.DACT_GELU, colsum=A.g(ib) if fold else None, colsum_rows=1),
               out.dense.weight)
        _linear_grads(rt, layer.intermediate.dense, df, h1, M, bias_done=fold, defer=defer)
        dh1 = _alloc(rt, M, Hd)
        _dgrad(rt, df, M, Hd, inner, hip.epilogue(dh1, Hd, residual=ds2), layer.intermediate.dense.weight)
        # LayerNorm 1 -> (dropout) -> attention output projection
        ds1 = _alloc(rt, M, Hd)
        ds1m = _alloc(rt, M, Hd) if d1[0] > 0 else None
        hip.layernorm_bwd(dt, dh1, s1, st1, so.LayerNorm.weight, ds1, ds1m, A.g(so.LayerNorm.weight), A.g(so.LayerNorm.bias), M, Hd, drop_out=d1,
                          dcolsum=A.g(so.dense.bias))
        dz1 = ds1m if ds1m is not None else ds1
        _linear_grads(rt, so.dense, dz1, ctxt, M, bias_done=True, defer=defer)
        dctx = _alloc(rt, M, Hd)
        _dgrad(rt, dz1, M, Hd, Hd, hip.epilogue(dctx, Hd), so.dense.weight)
        dqkv = _alloc(rt, M, 3 * Hd)
        hip.attention_bwd(dt, qkv, ctx["mask"], dctx, dqkv, B, L, heads, da)
        _linear_grads(rt, None, dqkv, h, M, dw=A.span([sa.query.weight, sa.key.weight, sa.value.weight], grad=True).view(3 * Hd, Hd),
                      db=A.span([sa.query.bias, sa.key.bias, sa.value.bias], grad=True), defer=defer)
        dhp = _alloc(rt, M, Hd)
        _dgrad(rt, dqkv, M, Hd, 3 * Hd, hip.epilogue(dhp, Hd, residual=ds1), sa.query.weight, sa.key.weight, sa.value.weight)
        dh = dhp
        if own_group is None:
            rt.grads_ready(layer)
        elif staged:
            own_group.launch()          # eager data parallel: per-layer groups, so the layer's region can be exchanged now
            if not pooler_done:
                rt.grads_ready(net.pooler)          # its weight gradient rode in the first group
                pooler_done = True
            rt.grads_ready(layer)
            defer = own_group = hip.WgradGroup(rt.dt)
    if not last:
        ctx["bwd_dh"] = dh
        return
    emb = net.embeddings
    ds0 = _alloc(rt, M, Hd)
    hip.layernorm_bwd(dt, dh, ctx["s0"], ctx["st0"], emb.LayerNorm.weight, ds0, None, A.g(emb.LayerNorm.weight), A.g(emb.LayerNorm.bias), M, Hd,
                      drop_in=ctx["d0"], dcolsum=A.g(emb.token_type_embeddings.weight)[0])      # token_type_ids = 0: every row adds to row 0
    hip.embed_bwd(dt, ctx["ids"], ds0, A.g(emb.word_embeddings.weight), A.g(emb.position_embeddings.weight), M, L, Hd, net.vocab,
                  padding_idx=0)      # HF BertEmbeddings: nn.Embedding(vocab, hidden, padding_idx=pad_token_id = 0)
    if own_group is not None:
        own_group.launch()
        if staged:
            rt.grads_ready(emb)       # the layers and the pooler were handed over as their groups were launched
        else:
            rt.grads_ready(net)
    else:
        rt.join_aux()
        rt.grads_ready(emb)
