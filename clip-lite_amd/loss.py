"""JSD mutual-information loss on the HIP kernels — the call surface of reference loss.py (MILinearBlock :12-40,
PriorDiscriminator :43-53, GlobalDiscriminatorDot :76-107, JSDInfoMaxLoss :110-314) with identical attribute names and
state_dict keys (`global_d.{img_block,text_block,temperature}`, `prior_d`, `text_prior_d`; the eval CLIs reach into
`loss.global_d.img_block` / `text_block`, reference retrieval.py:70-74).

Scheduling notes: the reference runs both MI blocks twice per step (positive pairs, then negatives with
the text batch rolled by one). BatchNorm batch statistics are permutation-invariant and the image input is identical in both
calls, so the projections are computed ONCE and the critic kernel pairs sample n with n+1; to stay checkpoint-compatible the
BatchNorm1d running statistics are still updated twice and num_batches_tracked advances by 2.
"""
import numpy as np
import torch
import torch.nn as nn

from . import hip
from .bert import LayerNormParams, LinearParams
from .resnet import BatchNormParams


class MILinearBlock(nn.Module):
    def __init__(self, feature_sz, units=2048, bln=True):
        super().__init__()
        self.feature_nonlinear = nn.Sequential(LinearParams(feature_sz, units, bias=False), BatchNormParams(units), nn.Identity(),
                                               LinearParams(units, units))
        self.feature_shortcut = LinearParams(feature_sz, units)
        self.feature_block_ln = LayerNormParams(units, 1e-5)
        # "initialize the initial projection to a sort of noisy copy" (reference loss.py:25-32)
        eye = torch.zeros(units, feature_sz, dtype=torch.bool)
        i = torch.arange(min(units, feature_sz))
        eye[i, i] = True
        self.feature_shortcut.weight.data.uniform_(-0.01, 0.01)
        self.feature_shortcut.weight.data.masked_fill_(eye, 1.0)
        self.bln = bln
        self.feature_sz, self.units = feature_sz, units

    def forward(self, feat):
        """Stand-alone projection (used by the eval CLIs as `projector(encoder(x))`); no gradient."""
        rt = _runtime_of(self)
        x = feat.to(rt.tdtype).contiguous()
        out, _ = mi_block_forward(rt, self, x, self.training, updates=1)
        return out


class PriorDiscriminator(nn.Module):
    def __init__(self, sz):
        super().__init__()
        self.l0 = LinearParams(sz, 1000)
        self.l1 = LinearParams(1000, 200)
        self.l2 = LinearParams(200, 1)
        self.sz = sz


class GlobalDiscriminator(nn.Module):
    """The `concat` critic (reference loss.py:56-68): l2(relu(l1(relu(l0(cat(features1, features2)))))), sz -> 512 -> 512 -> 1."""

    def __init__(self, sz):
        super().__init__()
        self.l0 = LinearParams(sz, 512)
        self.l1 = LinearParams(512, 512)
        self.l2 = LinearParams(512, 1)
        self.sz = sz


class GlobalDiscriminatorDot(nn.Module):
    def __init__(self, image_sz, text_sz, units=2048, bln=True):
        super().__init__()
        self.img_block = MILinearBlock(image_sz, units=units, bln=bln)
        self.text_block = MILinearBlock(text_sz, units=units, bln=bln)
        self.temperature = nn.Parameter(torch.ones([]) * np.log(1 / 0.07))


class JSDInfoMaxLoss(nn.Module):
    def __init__(self, image_dim=2048, text_dim=768, type="dot", prior_weight=0.1, image_prior=True, text_prior=False,
                 visual_self_supervised=False, textual_self_supervised=False):
        super().__init__()
        if type not in ("dot", "concat", "condot", "dotcon"):
            raise ValueError(f"critic type {type!r} (reference loss.py:129-169 knows dot, concat, condot, dotcon)")
        self.prior_weight, self.image_prior, self.text_prior = prior_weight, image_prior, text_prior
        self.image_dim, self.text_dim, self.type = image_dim, text_dim, type
        # reference loss.py:129-169: `dot`/`dotcon` use the projection-head critic for the cross-modal term, `concat`/`condot` the MLP on the
        # concatenated features; the self-supervised critics are of the dot kind for `dot`/`condot`, of the concat kind otherwise
        cross_dot, ssl_dot = type in ("dot", "dotcon"), type in ("dot", "condot")
        self.global_d = GlobalDiscriminatorDot(image_sz=image_dim, text_sz=text_dim) if cross_dot else GlobalDiscriminator(image_dim + text_dim)
        if visual_self_supervised:
            self.visual_d = GlobalDiscriminatorDot(image_sz=image_dim, text_sz=image_dim) if ssl_dot else GlobalDiscriminator(2 * image_dim)
        if textual_self_supervised:
            self.textual_d = GlobalDiscriminatorDot(image_sz=text_dim, text_sz=text_dim) if ssl_dot else GlobalDiscriminator(2 * text_dim)
        if image_prior:
            self.prior_d = PriorDiscriminator(sz=image_dim)
        if text_prior:
            self.text_prior_d = PriorDiscriminator(sz=text_dim)
        self._noise = None      # test hook: pins the two rand_like draws (image, text)
        self.estimator = "jsd"  # cross-modal term: "jsd" (reference loss.py:206-222,254) or "infonce" (InfoNCELoss below)

    def set_prior_noise(self, image_noise, text_noise):
        """Pin the prior noise (reference loss.py:189,196 draws torch.rand_like(image) then (text)); parity tests only."""
        self._noise = (image_noise, text_noise)

    def forward(self, image_features, text_features, neg_image_features=None, neg_text_features=None,
                aug_image_features=None, aug_text_features=None):
        rt = _runtime_of(self)
        extras = (neg_image_features, neg_text_features, aug_image_features, aug_text_features)
        if all(t is None for t in extras) and isinstance(self.global_d, GlobalDiscriminatorDot):
            # every shipped YAML: dot critic, in-batch negatives, no augmented views — the path the captured train step replays
            total, comps = _JSDLossFn.apply(image_features, text_features, self, rt, rt.next_step(self.training))
        else:
            if getattr(self, "estimator", "jsd") != "jsd":
                raise NotImplementedError("InfoNCELoss is defined for the dot critic with in-batch negatives only")
            if (neg_image_features is None) != (neg_text_features is None):
                raise ValueError("cluster mode needs both neg_image_features and neg_text_features (reference loss.py:225-252)")
            present = [t is not None for t in extras]
            args = [t if t is not None else image_features.new_zeros(()) for t in extras]
            total, comps = _JSDGeneralFn.apply(image_features, text_features, *args, present, self, rt, rt.next_step(self.training))
        return {"total_loss": total, "cross_modal_loss": comps[1], "visual_loss": comps[3], "textual_loss": comps[4]}


class InfoNCELoss(JSDInfoMaxLoss):
    """All-pairs InfoNCE variant (BASELINE config 4; SURVEY §8f N4 — the reference has no such code). Same projection heads, learnable
    temperature and prior discriminators as JSDInfoMaxLoss (identical parameters / state_dict keys); the cross-modal term is the symmetric
    cross-entropy, with diagonal targets, over S = exp(tau) * normalize(img_block(I)) @ normalize(text_block(T)).T: the B x B similarity is
    one MFMA GEMM, its two gradients two more. Negatives are the other captions / images of the local batch."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.estimator = "infonce"


def _runtime_of(module):
    rt = getattr(module, "_clite_rt", None)
    if rt is None:
        raise RuntimeError("clip_lite_amd: module is not attached to a device runtime; build it through VLInfoModel and move it to "
                           "the GPU with .to(device) (there is no CPU path)")
    return rt


# ---------------------------------------------------------------------------------------------------- executors
def _alloc(rt, *shape):
    return torch.empty(shape, device=rt.device, dtype=rt.tdtype)


def _mi_fused_ok(rt, blk, B, training):
    """The four-kernel form of the block's non-GEMM work (hip.mi_block_call; csrc/heads_fused.hip): bf16 training steps of at most 128 rows."""
    return (rt.fused_heads and rt.lowp and training and blk.bln and B <= 128 and blk.feature_sz % 16 == 0 and blk.units % 16 == 0 and not rt.precise_bn
            and not hip.is_deterministic())


def mi_block_forward(rt, blk, x, training, updates=2):
    """x: [B][F] compute dtype. Returns (LN(W2 relu(bn(W1 x)) + b2 + Ws x + bs), ctx)."""
    dt, A = rt.dt, rt.arena
    B, Fin, U = x.shape[0], blk.feature_sz, blk.units
    l1, bn, _, l2 = blk.feature_nonlinear
    if _mi_fused_ok(rt, blk, B, training):
        # five launches instead of ten: the three products accumulate (split-K, float atomics) into zeroed f32 workspaces, and the column-wise work
        # (BatchNorm1d statistics, both running-statistics updates, relu) and the row-wise work (biases, shortcut, LayerNorm) are one kernel each
        sc, ln = blk.feature_shortcut, blk.feature_block_ln
        z, a, t, out = (_alloc(rt, B, U) for _ in range(4))
        stats = hip.Stats(torch.empty(3 * U, device=rt.device, dtype=torch.float32), 1, U)
        lst = torch.empty(B, 2, device=rt.device, dtype=torch.float32)
        ws1, ws2 = rt.zpool.take(B * 2 * U).view(B, 2 * U), rt.zpool.take(B * U).view(B, U)
        hip.gemm_nt(dt, x, A.w(l1.weight), B, U, Fin, hip.epilogue(ws1, 2 * U, atomic=True))
        hip.gemm_nt(dt, x, A.w(sc.weight), B, U, Fin, hip.epilogue(ws1[:, U:], 2 * U, atomic=True))
        desc = hip.mi_block(M=B, Fin=Fin, U=U, updates=updates, momentum=bn.momentum, eps=bn.eps, ln_eps=ln.eps, bs=sc.bias, b2=l2.bias, gamma=bn.weight,
                            beta=bn.bias, running_mean=bn.running_mean, running_var=bn.running_var, z=z, a=a, stats=stats.t, sc=ws1, t=t, out=out,
                            ln_gamma=ln.weight, ln_beta=ln.bias, ln_stats=lst, dxs=ws2)
        hip.mi_block_call("fwd1", desc, x)
        hip.gemm_nt(dt, a, A.w(l2.weight), B, U, U, hip.epilogue(ws2, U, atomic=True))
        hip.mi_block_call("fwd2", desc, x)
        return out, (x, z, stats, a, t, lst)
    z = _alloc(rt, B, U)
    stats = rt.new_stats(U, B) if training else None
    hip.gemm_nt(dt, x, A.w(l1.weight), B, U, Fin, hip.epilogue(z, U, colsum=stats, ws=rt.gemm_ws(B, U)))
    if training and rt.precise_bn:
        hip.bn_centered_var(dt, z, stats, B, U)
    a = _alloc(rt, B, U)
    desc = hip.bn_desc(B, U, stats, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, training, bn.momentum, bn.eps, True, centered=rt.precise_bn)
    hip.bn_apply(dt, desc, z, None, a)
    for _ in range(updates - 1 if training else 0):      # the reference's second pass updates the running stats again
        desc2 = hip.bn_desc(B, U, stats, bn.weight, bn.bias, bn.running_mean, bn.running_var, True, True, bn.momentum, bn.eps, True, centered=rt.precise_bn)
        hip.bn_apply(dt, desc2, z, None, a)
    f = _alloc(rt, B, U)
    hip.gemm_nt(dt, a, A.w(l2.weight), B, U, U, hip.epilogue(f, U, bias=l2.bias, ws=rt.gemm_ws(B, U)))
    t = _alloc(rt, B, U)
    hip.gemm_nt(dt, x, A.w(blk.feature_shortcut.weight), B, U, Fin, hip.epilogue(t, U, bias=blk.feature_shortcut.bias, residual=f, ws=rt.gemm_ws(B, U)))
    if not blk.bln:
        return t, (x, z, stats, a, t, None)
    out = _alloc(rt, B, U)
    lst = torch.empty(B, 2, device=rt.device, dtype=torch.float32)
    ln = blk.feature_block_ln
    hip.layernorm_fwd(dt, t, ln.weight, ln.bias, ln.eps, out, lst, B, U)
    return out, (x, z, stats, a, t, lst)


def mi_block_backward(rt, blk, ctx, dout, dx_residual=None, defer=None):
    """Returns dx [B][F] (+ dx_residual if given). Parameter gradients accumulate into the arena. defer: a hip.WgradGroup that collects the three
    Linear weight gradients (and their bias sums) instead of launching them between the input-gradient GEMMs — they feed nothing but the arena,
    and in the captured step this chain sits between the encoders' forward and backward with nothing else on the chip."""
    from .bert import _linear_grads
    dt, A = rt.dt, rt.arena
    x, z, stats, a, t, lst = ctx
    B, Fin, U = x.shape[0], blk.feature_sz, blk.units
    l1, bn, _, l2 = blk.feature_nonlinear
    ln = blk.feature_block_ln
    if blk.bln:
        dtt = _alloc(rt, B, U)
        hip.layernorm_bwd(dt, dout, t, lst, ln.weight, dtt, None, A.g(ln.weight), A.g(ln.bias), B, U)
    else:
        dtt = dout
    sc = blk.feature_shortcut
    if _mi_fused_ok(rt, blk, B, True) and stats.R == 1 and all(p_.requires_grad for p_ in (l1.weight, l2.weight, sc.weight, bn.weight, bn.bias, l2.bias, sc.bias)):
        # five launches instead of nine: dtt Ws and dtt W2 accumulate side by side in one f32 workspace [B][Fin + U], one kernel turns the second half into
        # dz (BatchNorm1d backward, its two parameter gradients, both bias gradients), dz W1 accumulates onto the first half, one kernel stores dx
        dz, dx = _alloc(rt, B, U), _alloc(rt, B, Fin)
        ws3 = rt.zpool.take(B * (Fin + U)).view(B, Fin + U)
        hip.gemm_nn(dt, dtt, A.w(sc.weight), B, Fin, U, hip.epilogue(ws3, Fin + U, atomic=True))
        hip.gemm_nn(dt, dtt, A.w(l2.weight), B, U, U, hip.epilogue(ws3[:, Fin:], Fin + U, atomic=True))
        desc = hip.mi_block(M=B, Fin=Fin, U=U, updates=0, momentum=bn.momentum, eps=bn.eps, gamma=bn.weight, z=z, a=a, stats=stats.t, dtt=dtt, dz=dz, dxs=ws3,
                            dres=dx_residual, dx=dx, dgamma=A.g(bn.weight), dbeta=A.g(bn.bias), db2=A.g(l2.bias), dbs=A.g(sc.bias))
        hip.mi_block_call("bwd1", desc, dtt)
        hip.gemm_nn(dt, dz, A.w(l1.weight), B, Fin, U, hip.epilogue(ws3, Fin + U, atomic=True))
        hip.mi_block_call("bwd2", desc, dtt)
        _linear_grads(rt, sc, dtt, x, B, bias_done=True, defer=defer)
        _linear_grads(rt, l2, dtt, a, B, bias_done=True, defer=defer)
        _linear_grads(rt, l1, dz, x, B, defer=defer)
        return dx
    _linear_grads(rt, sc, dtt, x, B, defer=defer)
    dx = _alloc(rt, B, Fin)
    hip.gemm_nn(dt, dtt, A.w(sc.weight), B, Fin, U, hip.epilogue(dx, Fin, residual=dx_residual, ws=rt.gemm_ws(B, Fin)))
    _linear_grads(rt, l2, dtt, a, B, defer=defer)
    da = _alloc(rt, B, U)
    hip.gemm_nn(dt, dtt, A.w(l2.weight), B, U, U, hip.epilogue(da, U, ws=rt.gemm_ws(B, U)))
    dstats = rt.new_stats(U, B)
    hip.bn_bwd_reduce(dt, da, a, z, stats, dstats, B, U)
    dz = _alloc(rt, B, U)
    desc = hip.bn_desc(B, U, stats, bn.weight, bn.bias, bn.running_mean, bn.running_var, True, False, bn.momentum, bn.eps, False, centered=rt.precise_bn)
    hip.bn_bwd_apply(dt, desc, da, a, z, dstats, dz, None, A.g(bn.weight), A.g(bn.bias))
    _linear_grads(rt, l1, dz, x, B, defer=defer)
    dx2 = _alloc(rt, B, Fin)
    hip.gemm_nn(dt, dz, A.w(l1.weight), B, Fin, U, hip.epilogue(dx2, Fin, residual=dx, ws=rt.gemm_ws(B, Fin)))
    return dx2


def prior_forward(rt, pd, feat, noise, acc_slot, step, site):
    """-(mean log D(u) + mean log(1 - D(feat))) accumulated into acc_slot (f32 scalar view)."""
    dt, A = rt.dt, rt.arena
    B, sz = feat.shape
    x2 = _alloc(rt, 2 * B, sz)
    if noise is not None:
        x2[:B].copy_(noise)
    else:
        hip.uniform_fill(dt, x2, B * sz, step.seed, site)
    x2[B:].copy_(feat)
    return _mlp_tail_forward(rt, pd, x2, B, acc_slot, softplus=False)


def _mlp_tail_forward(rt, net, x2, half, acc_slot, softplus):
    """x2 [2*half][sz] -> relu(l0) -> relu(l1) -> l2 logits; acc_slot += mean softplus(-logit[:half]) + mean softplus(logit[half:]).
    Shared by PriorDiscriminator (sz -> 1000 -> 200 -> 1; rows = [noise; features]) and the concat critic GlobalDiscriminator
    (sz -> 512 -> 512 -> 1; rows = [positive pairs; negative pairs])."""
    dt, A = rt.dt, rt.arena
    R, sz = x2.shape
    n0, n1 = net.l0.weight.shape[0], net.l1.weight.shape[0]
    h0 = _alloc(rt, R, n0)
    hip.gemm_nt(dt, x2, A.w(net.l0.weight), R, n0, sz, hip.epilogue(h0, n0, bias=net.l0.bias, act=hip.ACT_RELU, ws=rt.gemm_ws(R, n0)))
    h1 = _alloc(rt, R, n1)
    hip.gemm_nt(dt, h0, A.w(net.l1.weight), R, n1, n0, hip.epilogue(h1, n1, bias=net.l1.bias, act=hip.ACT_RELU, ws=rt.gemm_ws(R, n1)))
    logit = torch.empty(R, device=rt.device, dtype=torch.float32)
    hip.prior_tail_fwd(dt, h1, net.l2.weight, net.l2.bias, half, n1, logit, acc_slot, softplus=softplus)
    return (x2, h0, h1, logit)


def _mlp_tail_backward(rt, net, ctx, gout, scale, defer=None):
    """Parameter gradients of the three layers into the arena; returns dh0 [2*half][n0] (gradient at l0's pre-activation)."""
    from .bert import _linear_grads
    dt, A = rt.dt, rt.arena
    x2, h0, h1, logit = ctx
    R = x2.shape[0]
    n0, n1 = net.l0.weight.shape[0], net.l1.weight.shape[0]
    dh1 = _alloc(rt, R, n1)
    hip.prior_tail_bwd(dt, h1, net.l2.weight, logit, gout, scale, R // 2, n1, dh1, A.g(net.l2.weight), A.g(net.l2.bias))
    _linear_grads(rt, net.l1, dh1, h0, R, defer=defer)
    dh0 = _alloc(rt, R, n0)
    hip.gemm_nn(dt, dh1, A.w(net.l1.weight), R, n0, n1, hip.epilogue(dh0, n0, dact_aux=h0, dact=hip.DACT_RELU, ws=rt.gemm_ws(R, n0)))
    _linear_grads(rt, net.l0, dh0, x2, R, defer=defer)
    return dh0


def prior_backward(rt, pd, ctx, gout, scale, dfeat_residual, defer=None):
    """Returns d(feat) [B][sz] + dfeat_residual."""
    dt, A = rt.dt, rt.arena
    B2, sz = ctx[0].shape
    B = B2 // 2
    dh0 = _mlp_tail_backward(rt, pd, ctx, gout, scale, defer=defer)
    dfeat = _alloc(rt, B, sz)
    # only the feature rows (B..2B) need an input gradient; the noise rows have none
    hip.gemm_nn(dt, dh0[B:], A.w(pd.l0.weight), B, sz, dh0.shape[1], hip.epilogue(dfeat, sz, residual=dfeat_residual, ws=rt.gemm_ws(B, sz)))
    return dfeat


def jsd_forward(rt, mod, img, txt, step):
    """image/text features -> (out f32[4] = [total, cross, prior, 0], saved state for jsd_backward). No autograd involved: the
    autograd Function below and the captured multi-graph step (train_loop.TrainStep) both drive the heads through this pair."""
    dt = rt.dt
    training = step.training
    img = img.to(rt.tdtype).contiguous()
    txt = txt.to(rt.tdtype).contiguous()
    B = img.shape[0]
    acc = torch.zeros(8, device=rt.device, dtype=torch.float32)
    noise = mod._noise or (None, None)
    pctx_i = pctx_t = None
    if mod.image_prior:
        pctx_i = prior_forward(rt, mod.prior_d, img, noise[0], acc[2:3], step, step.site())
    if mod.text_prior:
        pctx_t = prior_forward(rt, mod.text_prior_d, txt, noise[1], acc[3:4], step, step.site())
    gd = mod.global_d
    f1, c1 = mi_block_forward(rt, gd.img_block, img, training)
    f2, c2 = mi_block_forward(rt, gd.text_block, txt, training)
    if training:
        rt.bump_counters("loss", 2)
    U = gd.img_block.units
    if getattr(mod, "estimator", "jsd") == "infonce":
        if B % 8:
            raise RuntimeError("InfoNCELoss needs a batch that is a multiple of 8 (GEMM column granularity)")
        a, b = _alloc(rt, B, U), _alloc(rt, B, U)
        hip.l2_normalize(dt, f1, a, B, U)
        hip.l2_normalize(dt, f2, b, B, U)
        Cm = torch.empty(B, B, device=rt.device, dtype=torch.float32)
        hip.gemm_nt(dt, a, b, B, B, U, hip.epilogue(Cm, B, out_f32=True))          # all-pairs cosines on the MFMA engine
        lse = torch.empty(2, B, device=rt.device, dtype=torch.float32)
        hip.infonce_fwd(Cm, B, B, gd.temperature, lse[0], lse[1], acc)
        work = (a, b, Cm, lse)
    else:
        work = torch.empty(B, 8, device=rt.device, dtype=torch.float32)
        hip.critic_jsd_fwd(dt, f1, f2, gd.temperature, B, U, work, acc)
    out = torch.empty(8, device=rt.device, dtype=torch.float32)
    hip.loss_finalize(acc, mod.prior_weight, out)
    return out, (f1, f2, c1, c2, work, pctx_i, pctx_t, B)


def jsd_backward(rt, mod, saved, gout):
    """gout: f32[1] device scalar dL/d(total). Returns (d image_features, d text_features) in the compute dtype; parameter
    gradients accumulate into the arena."""
    dt, A = rt.dt, rt.arena
    f1, f2, c1, c2, work, pctx_i, pctx_t, B = saved
    gd = mod.global_d
    U = gd.img_block.units
    df1, df2 = _alloc(rt, B, U), _alloc(rt, B, U)
    if getattr(mod, "estimator", "jsd") == "infonce":
        a, b, Cm, lse = work
        dC = _alloc(rt, B, B)
        hip.infonce_bwd(dt, Cm, B, B, gd.temperature, lse[0], lse[1], gout, 1.0 - mod.prior_weight, dC, B, A.g(gd.temperature).view(1))
        da, db = _alloc(rt, B, U), _alloc(rt, B, U)
        hip.gemm_nn(dt, dC, b, B, U, B, hip.epilogue(da, U))                        # da = dC  b
        hip.gemm_tn(dt, dC, a, B, U, B, hip.epilogue(db, U))                        # db = dC^T a
        hip.l2_normalize_bwd(dt, f1, a, da, df1, B, U)
        hip.l2_normalize_bwd(dt, f2, b, db, df2, B, U)
    else:
        hip.critic_jsd_bwd(dt, f1, f2, gd.temperature, work, gout, 1.0 - mod.prior_weight, B, U, df1, df2, A.g(gd.temperature).view(1))
    dimg = prior_backward(rt, mod.prior_d, pctx_i, gout, mod.prior_weight, None) if pctx_i is not None else None
    dtxt = prior_backward(rt, mod.text_prior_d, pctx_t, gout, mod.prior_weight, None) if pctx_t is not None else None
    dimg = mi_block_backward(rt, gd.img_block, c1, df1, dimg)
    dtxt = mi_block_backward(rt, gd.text_block, c2, df2, dtxt)
    rt.join_aux()
    rt.grads_ready(mod)
    return dimg, dtxt


# ---- the same heads cut along the modality boundary (captured step only: train_loop.TrainStep) --------------------------------------------
# Everything but the critic touches one modality only — the MI projection block and the prior discriminator of the image features, and
# those of the text features — and the prior terms' gradients need nothing but their own forward (dL/d total is the constant 1). So the
# text half runs on the text encoder's stream: its forward (and the prior's backward) right behind BERT's forward, hidden under the tail of
# the ResNet forward; its block backward right behind the critic, beside the image half. Only the critic sits on the join.
def jsd_half_forward(rt, mod, feat, which, step, site, acc, gout, defer=None):
    """which: "image" | "text". Prior discriminator forward + backward and MI-block forward of one modality. acc: the step's 8 zeroed
    accumulators (slot 2 / 3 = this prior's term). Returns the state jsd_join / jsd_half_backward need."""
    half = jsd_half_block(rt, mod, feat, which, step)
    half["dprior"] = jsd_half_prior(rt, mod, feat, which, step, site, acc, gout, defer=defer)
    return half


def jsd_half_prior(rt, mod, feat, which, step, site, acc, gout, defer=None):
    """The prior discriminator of one modality, forward and backward (its term goes to acc[2] / acc[3], its weight gradients to the arena);
    returns the gradient it sends to the features, or None when that prior is off. Independent of jsd_half_block: a captured step runs the
    two on different streams (train_loop.TrainStep)."""
    feat = feat.to(rt.tdtype).contiguous()
    img = which == "image"
    if not (mod.image_prior if img else mod.text_prior):
        return None
    noise = (mod._noise or (None, None))[0 if img else 1]
    pd = mod.prior_d if img else mod.text_prior_d
    pctx = prior_forward(rt, pd, feat, noise, acc[2:3] if img else acc[3:4], step, site)
    return prior_backward(rt, pd, pctx, gout, mod.prior_weight, None, defer=defer)


def jsd_half_block(rt, mod, feat, which, step):
    """MI-block forward of one modality (reference loss.py:21-45)."""
    feat = feat.to(rt.tdtype).contiguous()
    blk = mod.global_d.img_block if which == "image" else mod.global_d.text_block
    f, c = mi_block_forward(rt, blk, feat, step.training)
    return {"f": f, "c": c, "dprior": None, "blk": blk}


def jsd_join(rt, mod, hi, ht, acc, gout):
    """Critic forward, the total, critic backward (JSD estimator, or the InfoNCE all-pairs variant). Returns (out f32[8], df_image, df_text)."""
    dt, A = rt.dt, rt.arena
    gd = mod.global_d
    f1, f2 = hi["f"], ht["f"]
    B, U = f1.shape[0], gd.img_block.units
    out = torch.empty(8, device=rt.device, dtype=torch.float32)
    df1, df2 = _alloc(rt, B, U), _alloc(rt, B, U)
    if getattr(mod, "estimator", "jsd") == "infonce":
        if B % 8:
            raise RuntimeError("InfoNCELoss needs a batch that is a multiple of 8 (GEMM column granularity)")
        a, b = _alloc(rt, B, U), _alloc(rt, B, U)
        hip.l2_normalize(dt, f1, a, B, U)
        hip.l2_normalize(dt, f2, b, B, U)
        Cm = torch.empty(B, B, device=rt.device, dtype=torch.float32)
        hip.gemm_nt(dt, a, b, B, B, U, hip.epilogue(Cm, B, out_f32=True))
        lse = torch.empty(2, B, device=rt.device, dtype=torch.float32)
        hip.infonce_fwd(Cm, B, B, gd.temperature, lse[0], lse[1], acc)
        hip.loss_finalize(acc, mod.prior_weight, out)
        dC = _alloc(rt, B, B)
        hip.infonce_bwd(dt, Cm, B, B, gd.temperature, lse[0], lse[1], gout, 1.0 - mod.prior_weight, dC, B, A.g(gd.temperature).view(1))
        da, db = _alloc(rt, B, U), _alloc(rt, B, U)
        hip.gemm_nn(dt, dC, b, B, U, B, hip.epilogue(da, U))
        hip.gemm_tn(dt, dC, a, B, U, B, hip.epilogue(db, U))
        hip.l2_normalize_bwd(dt, f1, a, da, df1, B, U)
        hip.l2_normalize_bwd(dt, f2, b, db, df2, B, U)
    else:
        work = torch.empty(B, 8, device=rt.device, dtype=torch.float32)
        hip.critic_jsd_fwd(dt, f1, f2, gd.temperature, B, U, work, acc)
        hip.loss_finalize(acc, mod.prior_weight, out)
        hip.critic_jsd_bwd(dt, f1, f2, gd.temperature, work, gout, 1.0 - mod.prior_weight, B, U, df1, df2, A.g(gd.temperature).view(1))
    return out, df1, df2


def jsd_half_backward(rt, half, df, defer=None):
    """MI-block backward of one modality; returns the gradient of its features (prior part included). defer: see mi_block_backward."""
    return mi_block_backward(rt, half["blk"], half["c"], df, half["dprior"], defer=defer)


class _JSDLossFn(torch.autograd.Function):
    """image/text features -> (total, [total, cross, prior, 0]); backward drives the head kernels and returns feature grads."""

    @staticmethod
    def forward(ctx, img, txt, mod, rt, step):
        out, saved = jsd_forward(rt, mod, img, txt, step)
        ctx.mod, ctx.rt, ctx.saved = mod, rt, saved
        ctx.in_dtypes = (img.dtype, txt.dtype)
        ctx.mark_non_differentiable(out)
        return out[0].clone(), out

    @staticmethod
    def backward(ctx, gtotal, _gcomps):
        gout = gtotal.to(torch.float32).contiguous().view(1)
        dimg, dtxt = jsd_backward(ctx.rt, ctx.mod, ctx.saved, gout)
        return dimg.to(ctx.in_dtypes[0]), dtxt.to(ctx.in_dtypes[1]), None, None, None


# ---------------------------------------------------------------------------------------------------- general form
# Critic types concat / condot / dotcon (reference loss.py:129-169), the cluster hard-negative branch (:225-252) and the visual / textual
# self-supervised terms (:256-300). Eager launches only: these forms call the encoders several times per step, which the captured
# train step does not model.
def _neg_perm(rt, rows, cluster):
    """(neg, neg_inv) int32 device tensors for the critic kernels, or (None, None) for roll-by-one. Cluster mode stacks
    [batch (B rows); hard negatives (B rows)]: row i < B is paired with its hard negative B + i, row B + i with caption (i + 1) mod B
    (reference loss.py:236-246)."""
    if not cluster:
        return None, None
    cache = rt.__dict__.setdefault("_neg_perm_cache", {})
    if rows not in cache:
        half = rows // 2
        i = torch.arange(half)
        neg = torch.cat((i + half, (i + 1) % half)).to(torch.int32)
        inv = torch.empty(rows, dtype=torch.int32)
        inv[neg.long()] = torch.arange(rows, dtype=torch.int32)
        cache[rows] = (neg.to(rt.device), inv.to(rt.device))
    return cache[rows]


def _take_rows(t, index):
    """t[index] (a row gather = data movement only); index None = roll by one: out[i] = t[(i + 1) mod n]."""
    if index is None:
        return torch.cat((t[1:], t[:1]), dim=0)
    return t.index_select(0, index.long())


def _put_rows_back(t, index_inv):
    """Inverse of _take_rows: out[j] = t[i] where j = index[i]."""
    if index_inv is None:
        return torch.cat((t[-1:], t[:-1]), dim=0)
    return t.index_select(0, index_inv.long())


def _pair_forward(rt, critic, f1, f2, perm, acc2, training):
    """One (positive, negative) critic evaluation: acc2[0] += mean softplus(-critic(f1, f2)), acc2[1] += mean softplus(critic(f1, f2[neg]))."""
    dt = rt.dt
    neg, _ = perm
    R = f1.shape[0]
    if isinstance(critic, GlobalDiscriminatorDot):
        p1, c1 = mi_block_forward(rt, critic.img_block, f1, training)
        p2, c2 = mi_block_forward(rt, critic.text_block, f2, training)
        if training:      # the reference runs each block twice per step (positives, negatives): running stats and counters advance twice
            for blk in (critic.img_block, critic.text_block):
                blk.feature_nonlinear[1]._buffers["num_batches_tracked"] += 2
        work = torch.empty(R, 8, device=rt.device, dtype=torch.float32)
        hip.critic_jsd_fwd(dt, p1, p2, critic.temperature, R, critic.img_block.units, work, acc2, neg=neg)
        return ("dot", p1, p2, c1, c2, work)
    x2 = torch.cat((torch.cat((f1, f2), dim=1), torch.cat((f1, _take_rows(f2, neg)), dim=1)), dim=0).contiguous()
    return ("concat", _mlp_tail_forward(rt, critic, x2, R, acc2, softplus=True), f1.shape[1])


def _pair_backward(rt, critic, saved, perm, gout, scale):
    """-> (d f1, d f2); parameter gradients accumulate into the arena."""
    dt, A = rt.dt, rt.arena
    neg, neg_inv = perm
    if saved[0] == "dot":
        _, p1, p2, c1, c2, work = saved
        R, U = p1.shape
        dp1, dp2 = _alloc(rt, R, U), _alloc(rt, R, U)
        hip.critic_jsd_bwd(dt, p1, p2, critic.temperature, work, gout, scale, R, U, dp1, dp2, A.g(critic.temperature).view(1), neg=neg, neg_inv=neg_inv)
        return mi_block_backward(rt, critic.img_block, c1, dp1), mi_block_backward(rt, critic.text_block, c2, dp2)
    _, ctx, d1 = saved
    R2, sz = ctx[0].shape
    R, d2 = R2 // 2, sz - d1
    dh0 = _mlp_tail_backward(rt, critic, ctx, gout, scale)
    n0 = dh0.shape[1]
    w0 = A.w(critic.l0.weight)                       # [n0][d1 + d2]: column blocks address the two concatenated inputs
    df1a, df1 = _alloc(rt, R, d1), _alloc(rt, R, d1)
    hip.gemm_nn(dt, dh0[:R], w0, R, d1, n0, hip.epilogue(df1a, d1), ldb=sz)
    hip.gemm_nn(dt, dh0[R:], w0, R, d1, n0, hip.epilogue(df1, d1, residual=df1a), ldb=sz)
    w0b = w0[:, d1:]
    dneg = _alloc(rt, R, d2)
    hip.gemm_nn(dt, dh0[R:], w0b, R, d2, n0, hip.epilogue(dneg, d2), ldb=sz)
    df2 = _alloc(rt, R, d2)
    hip.gemm_nn(dt, dh0[:R], w0b, R, d2, n0, hip.epilogue(df2, d2, residual=_put_rows_back(dneg, neg_inv).contiguous()), ldb=sz)
    return df1, df2


def _sum(rt, a, b):
    if a is None:
        return b
    if b is None:
        return a
    out = torch.empty_like(a)
    hip.add(rt.dt, a.contiguous(), b.contiguous(), out)
    return out


def jsd_general_forward(rt, mod, img, txt, nimg, ntxt, aimg, atxt, step):
    training = step.training
    cast = lambda t: None if t is None else t.to(rt.tdtype).contiguous()
    img, txt, nimg, ntxt, aimg, atxt = (cast(t) for t in (img, txt, nimg, ntxt, aimg, atxt))
    B = img.shape[0]
    acc = torch.zeros(8, device=rt.device, dtype=torch.float32)
    noise = mod._noise or (None, None)
    pctx_i = prior_forward(rt, mod.prior_d, img, noise[0], acc[2:3], step, step.site()) if mod.image_prior else None
    pctx_t = prior_forward(rt, mod.text_prior_d, txt, noise[1], acc[3:4], step, step.site()) if mod.text_prior else None
    cluster = ntxt is not None
    if cluster:
        f1, f2 = torch.cat((img, nimg), dim=0).contiguous(), torch.cat((txt, ntxt), dim=0).contiguous()
    else:
        f1, f2 = img, txt
    perm = _neg_perm(rt, f1.shape[0], cluster)
    cross = _pair_forward(rt, mod.global_d, f1, f2, perm, acc[0:2], training)
    # reference loss.py:236-238 re-binds text_features to the rolled captions in cluster mode; the textual SSL term then pairs those
    txt_ssl = _take_rows(txt, None).contiguous() if cluster else txt
    vis = _pair_forward(rt, mod.visual_d, img, aimg, (None, None), acc[4:6], training) if aimg is not None else None
    tex = _pair_forward(rt, mod.textual_d, txt_ssl, atxt, (None, None), acc[6:8], training) if atxt is not None else None
    out = torch.empty(8, device=rt.device, dtype=torch.float32)
    hip.loss_finalize(acc, mod.prior_weight, out)
    return out, (cross, vis, tex, perm, pctx_i, pctx_t, B, cluster)


def jsd_general_backward(rt, mod, saved, gout):
    """-> gradients w.r.t. (img, txt, nimg, ntxt, aimg, atxt), None where the input was absent."""
    cross, vis, tex, perm, pctx_i, pctx_t, B, cluster = saved
    s = 1.0 - mod.prior_weight
    d1, d2 = _pair_backward(rt, mod.global_d, cross, perm, gout, s)
    dimg, dtxt, dnimg, dntxt = (d1[:B], d2[:B], d1[B:], d2[B:]) if cluster else (d1, d2, None, None)
    daimg = datxt = None
    if vis is not None:
        dv1, daimg = _pair_backward(rt, mod.visual_d, vis, (None, None), gout, s)
        dimg = _sum(rt, dimg, dv1)
    if tex is not None:
        dt1, datxt = _pair_backward(rt, mod.textual_d, tex, (None, None), gout, s)
        dtxt = _sum(rt, dtxt, _put_rows_back(dt1, None) if cluster else dt1)
    if pctx_i is not None:
        dimg = prior_backward(rt, mod.prior_d, pctx_i, gout, mod.prior_weight, dimg.contiguous())
    if pctx_t is not None:
        dtxt = prior_backward(rt, mod.text_prior_d, pctx_t, gout, mod.prior_weight, dtxt.contiguous())
    rt.join_aux()
    rt.grads_ready(mod)
    return dimg, dtxt, dnimg, dntxt, daimg, datxt


class _JSDGeneralFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, txt, nimg, ntxt, aimg, atxt, present, mod, rt, step):
        ins = [img, txt] + [t if ok else None for t, ok in zip((nimg, ntxt, aimg, atxt), present)]
        out, saved = jsd_general_forward(rt, mod, *ins, step)
        ctx.mod, ctx.rt, ctx.saved = mod, rt, saved
        ctx.in_dtypes = [None if t is None else t.dtype for t in ins]
        ctx.mark_non_differentiable(out)
        return out[0].clone(), out

    @staticmethod
    def backward(ctx, gtotal, _gcomps):
        gout = gtotal.to(torch.float32).contiguous().view(1)
        grads = jsd_general_backward(ctx.rt, ctx.mod, ctx.saved, gout)
        outs = tuple(None if (g is None or d is None) else g.to(d) for g, d in zip(grads, ctx.in_dtypes))
        return outs + (None, None, None, None)
