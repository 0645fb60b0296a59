"""ResNet image encoder on the HIP kernels: torchvision-topology module tree (parameter/buffer names identical to
torchvision.models.resnet*, which is what reference encoder.py:36-41 instantiates and encoder.py:84-94 renames) plus the
hand-scheduled forward/backward executor that drives the implicit-GEMM conv, BatchNorm and pooling kernels.

Data layout: activations NHWC, flattened to [N*H*W][C]; every conv output y (pre-BN) and every block output a (post
BN/ReLU) is kept for backward. Train-mode BN statistics come out of the conv epilogue (per-channel sum / sum of squares
through float atomics), so BN costs one apply pass forward and a reduce + apply pass backward.
"""
import math

import torch
import torch.nn as nn

from . import hip


class Conv2dParams(nn.Module):
    """Parameter holder with nn.Conv2d's state_dict surface (weight [K][C][R][S], bias=False)."""

    def __init__(self, cin, cout, k, stride, pad):
        super().__init__()
        self.in_channels, self.out_channels, self.k, self.stride, self.pad = cin, cout, k, stride, pad
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        nn.init.kaiming_normal_(self.weight, mode="fan_out", nonlinearity="relu")   # torchvision ResNet.__init__


class BatchNormParams(nn.Module):
    """Parameter/buffer holder with nn.BatchNorm{1,2}d's state_dict surface."""

    def __init__(self, c, eps=1e-5, momentum=0.1):
        super().__init__()
        self.num_features, self.eps, self.momentum = c, eps, momentum
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = Conv2dParams(inplanes, planes, 3, stride, 1)
        self.bn1 = BatchNormParams(planes)
        self.conv2 = Conv2dParams(planes, planes, 3, 1, 1)
        self.bn2 = BatchNormParams(planes)
        self.downsample = downsample

    def units(self):
        return [(self.conv1, self.bn1), (self.conv2, self.bn2)]


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = Conv2dParams(inplanes, planes, 1, 1, 0)
        self.bn1 = BatchNormParams(planes)
        self.conv2 = Conv2dParams(planes, planes, 3, stride, 1)     # stride on the 3x3 (torchvision >= 0.4 "v1.5")
        self.bn2 = BatchNormParams(planes)
        self.conv3 = Conv2dParams(planes, planes * 4, 1, 1, 0)
        self.bn3 = BatchNormParams(planes * 4)
        self.downsample = downsample

    def units(self):
        return [(self.conv1, self.bn1), (self.conv2, self.bn2), (self.conv3, self.bn3)]


SPECS = {"resnet18": (BasicBlock, (2, 2, 2, 2)), "resnet34": (BasicBlock, (3, 4, 6, 3)), "resnet50": (Bottleneck, (3, 4, 6, 3)),
         "resnet101": (Bottleneck, (3, 4, 23, 3)), "resnet152": (Bottleneck, (3, 8, 36, 3))}


class ResNet(nn.Module):
    def __init__(self, name="resnet50"):
        super().__init__()
        if name not in SPECS:
            raise ValueError(f"unsupported visual backbone {name!r}; supported: {sorted(SPECS)}")
        block, layers = SPECS[name]
        self.inplanes = 64
        self.conv1 = Conv2dParams(3, 64, 7, 2, 3)
        self.bn1 = BatchNormParams(64)
        self.layer1 = self._make_layer(block, 64, layers[0], 1)
        self.layer2 = self._make_layer(block, 128, layers[1], 2)
        self.layer3 = self._make_layer(block, 256, layers[2], 2)
        self.layer4 = self._make_layer(block, 512, layers[3], 2)
        self.fc = nn.Identity()                                        # reference encoder.py:41
        self.out_dim = 512 * block.expansion

    def _make_layer(self, block, planes, blocks, stride):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(Conv2dParams(self.inplanes, planes * block.expansion, 1, stride, 0),
                                       BatchNormParams(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return nn.Sequential(*layers)

    def blocks(self):
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            for b in layer:
                yield b


# ---------------------------------------------------------------------------------------------------- executor
class _Unit:
    """Saved tensors of one conv->BN unit."""
    __slots__ = ("conv", "bn", "cv", "x", "y", "stats", "out", "bits", "pair", "xsum", "x8")


def _alloc(rt, *shape):
    return torch.empty(shape, device=rt.device, dtype=rt.tdtype)


def _bn_desc(rt, bn, M, stats, relu, training, res=None, bits=None, fp8=None, out_sum=None):
    res_bn = None
    if res is not None:
        rbn, rstats = res
        res_bn = (rstats, rbn.weight, rbn.bias, rbn.running_mean, rbn.running_var)
    return hip.bn_desc(M, bn.num_features, stats, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                       training, training, bn.momentum, bn.eps, relu, res_bn, centered=rt.precise_bn, relu_bits=bits, fp8=fp8, out_sum=out_sum)


def _relu_bits(rt, M, Cc, training):
    """Packed ReLU mask of a [M][Cc] activation (one bit per element, written by bn_apply): what the backward pass reads instead of the
    activation itself wherever only its sign matters — the BatchNorm-backward dgrad epilogues and the BN backward kernels."""
    return torch.empty(M, Cc // 8, device=rt.device, dtype=torch.uint8) if training else None


def _conv(rt, x, N, H, W, conv, training, f8=None, x8=None, pair=False):
    """f8: the network's fp8.Fp8Forward (or None); x8: the e4m3 copy of x its producer wrote (hip.Fp8View), when it did.
    pair: y is slot 1 of a [2][M][K] buffer whose slot 0 will hold the masked gradient w.r.t. the BatchNorm output in backward (_fold_ok)."""
    cv = hip.conv_desc(rt.dt, N, H, W, conv.in_channels, conv.out_channels, conv.k, conv.k, conv.stride, conv.pad)
    M = N * cv.Ho * cv.Wo
    pr = _alloc(rt, 2, M, conv.out_channels) if pair else None
    y = pr[1] if pair else _alloc(rt, M, conv.out_channels)
    stats = rt.new_stats(conv.out_channels, M) if training else None
    xq = None
    if f8 is not None and f8.wants(conv, H):
        # OCP e4m3 operands (include/clite.h: clite_conv_fwd_fp8; policy and scaling: fp8.py); y, the statistics and everything backward stay as they are.
        # An input without a producer-written copy (no scale yet on the first step; the pooled stem output) is quantised here, current scaling
        xq = x8 if x8 is not None else hip.Fp8Tensor(x, rt.dt)
        hip.conv_fwd_fp8(xq, f8.weight(conv), cv, hip.epilogue(y, conv.out_channels, colsum=stats))
    else:
        hip.conv_fwd(x, rt.arena.w(conv.weight), cv, hip.epilogue(y, conv.out_channels, colsum=stats))
    if training and rt.precise_bn:
        hip.bn_centered_var(rt.dt, y, stats, M, conv.out_channels)
    u = _Unit()
    u.conv, u.cv, u.x, u.y, u.stats, u.pair, u.xsum = conv, cv, x, y, stats, pr, None
    u.x8 = xq if training else None          # the e4m3 copy of the input: the fp8 weight gradient's second operand (DeviceRuntime.fp8_wgrad)
    return u


def _fold_wanted(rt, conv, rows, training, last_block):
    """Forward-time half of the decision to fold a block-output BatchNorm's backward into conv3's gradients (DeviceRuntime.bn_fold; hip.bn_fold_prepare):
    a 1 x 1 / stride 1 convolution with enough rows that its apply pass is HBM-bound (the 56 x 56 and 28 x 28 stages at batch 128; at 14 x 14 the doubled
    K of the input gradient costs what the pass saves), bf16, not the network's last block (its gradient comes from the pooling, unmasked and without sums)."""
    return (training and rt.bn_fold and rt.lowp and rt.fuse_bn_backward and rt.transposed_dgrad and not last_block and conv.k == 1 and conv.stride == 1
            and conv.out_channels % 64 == 0 and rows >= rt.bn_fold_min_rows and 2 * rows * conv.out_channels < 2 ** 31)


def _wd(rt, conv):
    """(weight operand of the conv's dgrad, is it the transposed copy): bf16 mode reads [C][R][S][K] (Arena.wt) so the dgrad GEMM has both operands
    k-contiguous like the forward; the exact-f32 mode reads the weight as stored."""
    A = rt.arena
    if rt.transposed_dgrad and A.has_wt(conv.weight):
        return A.wt(conv.weight), True
    return A.w(conv.weight), False


def s2_class_weights(rt, net):
    """{id(conv): the four tap-subset weights} of every conv whose dgrad runs as parity classes (hip.conv_dgrad_s2); put into the forward
    context as ctx["s2w"] by a caller that wants them made ahead of backward (train_loop.TrainStep: on the side stream, at the start of the step)."""
    out = {}
    if not rt.s2_classes:
        return out
    for blk in net.blocks():
        for conv, _ in blk.units()[1:]:           # (a block's first conv takes the block-input path of resnet_backward, not the parity classes)
            if conv.k == 3 and conv.stride == 2 and conv.pad == 1 and conv.in_channels % 64 == 0 and conv.out_channels % 64 == 0:
                out[id(conv)] = hip.s2_class_weights(_wd(rt, conv)[0])
    return out


def stage_image(rt, image, out=None):
    """f32 NCHW [N][3][H][W] on the device -> the stem's input form: zero-padded NHWC4 in the compute dtype (pad 3, row pitch rounded up).
    `out` re-uses an earlier result's storage: the captured train step (train_loop.TrainStep) stages every batch into the one buffer its
    graphs read, straight from the caller's tensor — the only per-step copy of the image."""
    N, _, H, W = image.shape
    Hp, Wp = H + 6, W + 6 + 2
    Wp += Wp % 2
    xpad = _alloc(rt, N, Hp, Wp, 4) if out is None else out
    assert tuple(xpad.shape) == (N, Hp, Wp, 4) and xpad.dtype == rt.tdtype
    hip.image_to_nhwc4(rt.dt, image, xpad, N, H, W, 3, Hp, Wp)
    return xpad


def resnet_forward(rt, net, image, training, staged=None):
    """image: f32 NCHW [N][3][H][W] on the device, or None with `staged` = (stage_image(...) result, H, W).
    Returns (features [N][C] in compute dtype, ctx for backward)."""
    dt = rt.dt
    if staged is None:
        N, _, H, W = image.shape
        xpad = stage_image(rt, image)
    else:
        xpad, H, W = staged
        N = xpad.shape[0]
    ctx = {"N": N}
    # stem: 7x7/2 pad 3 on the pre-padded NHWC4 image
    Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    Hp, Wp = xpad.shape[1], xpad.shape[2]
    wv = _alloc(rt, 64, 7, 8, 4)
    hip.stem_pack(dt, rt.arena.w32(net.conv1.weight), wv)
    y0 = _alloc(rt, N * Ho * Wo, 64)
    st0 = rt.new_stats(64, N * Ho * Wo) if training else None
    hip.stem_fwd(dt, xpad, wv, N, Hp, Wp, Ho, Wo, hip.epilogue(y0, 64, colsum=st0))
    if training and rt.precise_bn:
        hip.bn_centered_var(dt, y0, st0, N * Ho * Wo, 64)
    # BatchNorm + ReLU + max-pool in one pass: the post-BN stem activation (the largest tensor of the step) is never stored
    Hq, Wq = (Ho + 2 - 3) // 2 + 1, (Wo + 2 - 3) // 2 + 1
    p0 = _alloc(rt, N * Hq * Wq, 64)
    idx = torch.empty(N * Hq * Wq, 64, device=rt.device, dtype=torch.uint8)
    # pooled-size operands of bn1's backward reductions (hip.stem_bn_pool_fwd / DeviceRuntime.stem_pooled_stats): y0 at each window's argmax and the
    # relu' bits of the pooled output, so that layer1's first block accumulates the two sums in the epilogue that writes the pooled gradient
    pooled_stats = training and rt.stem_pooled_stats and rt.lowp and rt.fuse_bn_backward and rt.transposed_dgrad
    ymax = _alloc(rt, N * Hq * Wq, 64) if pooled_stats else None
    sbits = _relu_bits(rt, N * Hq * Wq, 64, True) if pooled_stats else None
    hip.stem_bn_pool_fwd(dt, _bn_desc(rt, net.bn1, N * Ho * Wo, st0, True, training, bits=sbits), y0, p0, idx, N, Ho, Wo, ymax=ymax)
    ctx["stem"] = (xpad, Hp, Wp, Ho, Wo, y0, st0, idx, Hq, Wq, ymax, sbits)

    # fp8 forward (BASELINE configs[4]; fp8.py): the eligible convs read e4m3 copies that the producing bn_apply wrote beside its bf16 output
    from .fp8 import forward_state
    f8 = forward_state(rt, net)
    if f8 is not None:
        f8.begin_step()
    x, Hc, Wc = p0, Hq, Wq
    x8 = None                           # (the pooled stem output has no fused copy; 1x1 convs at 56 x 56 do not want one)
    recs = []
    blocks = list(net.blocks())
    for bi, blk in enumerate(blocks):
        units = []
        xin, xin8, Hin, Win = x, x8, Hc, Wc
        h, h8, Hh, Wh = xin, xin8, Hin, Win
        specs = blk.units()
        fold_blk = (len(specs) == 3 and _fold_wanted(rt, specs[-1][0], N * ((Hin + 2 - 3) // specs[1][0].stride + 1) * ((Win + 2 - 3) // specs[1][0].stride + 1), training,
                                                   bi == len(blocks) - 1) and not (f8 is not None and f8.wants(specs[-1][0], (Hin + 2 - 3) // specs[1][0].stride + 1)))
        for i, (conv, bn) in enumerate(specs):
            last = i == len(specs) - 1
            u = _conv(rt, h, N, Hh, Wh, conv, training, f8, h8, pair=last and fold_blk)
            u.bn = bn
            Hh, Wh = u.cv.Ho, u.cv.Wo
            M = N * Hh * Wh
            u.out = _alloc(rt, M, conv.out_channels)
            u.bits = _relu_bits(rt, M, conv.out_channels, training)
            if not last:
                q, h8 = f8.producer(bn, M, conv.out_channels, f8.wants(specs[i + 1][0], Hh), training) if f8 is not None else (None, None)
                # conv3's input in a folding block: its column sums come out of this pass (the folded weight gradient's colsum(a))
                # (32 replicas: the apply pass's ~1024 workgroups all reach their 8 atomics per thread at the end of the launch — with the 2 - 4 replicas
                # of the convolutions' statistics, whose tiles arrive spread over the launch, that burst cost 40 - 70 us per pass)
                osum = hip.Stats(rt.zpool.take(32 * 3 * conv.out_channels), 32, conv.out_channels) if (fold_blk and i == len(specs) - 2 and q is None) else None
                hip.bn_apply(dt, _bn_desc(rt, bn, M, u.stats, True, training, bits=u.bits, fp8=q, out_sum=osum), u.y, None, u.out)
                u.xsum = osum
            units.append(u)
            h = u.out
        ud = None
        last = units[-1]
        M = N * Hh * Wh
        q = x8 = None
        if f8 is not None and bi + 1 < len(blocks):          # the block output's readers: the next block's first conv and its downsample conv
            nb = blocks[bi + 1]
            readers = [nb.units()[0][0]] + ([nb.downsample[0]] if nb.downsample is not None else [])
            q, x8 = f8.producer(last.bn, M, last.conv.out_channels, any(f8.wants(c, Hh) for c in readers), training)
        if blk.downsample is not None:
            ud = _conv(rt, xin, N, Hin, Win, blk.downsample[0], training, f8, xin8)
            ud.bn = blk.downsample[1]
            hip.bn_apply(dt, _bn_desc(rt, last.bn, M, last.stats, True, training, res=(ud.bn, ud.stats), bits=last.bits, fp8=q), last.y, ud.y, last.out)
        else:
            hip.bn_apply(dt, _bn_desc(rt, last.bn, M, last.stats, True, training, bits=last.bits, fp8=q), last.y, xin, last.out)
        recs.append((units, ud, Hin, Win))
        x, Hc, Wc = last.out, Hh, Wh
    if f8 is not None and training:          # (an eval forward leaves the delayed-scaling state alone)
        f8.end_step()
    Cout = net.out_dim
    feat = _alloc(rt, N, Cout)
    hip.avgpool_fwd(dt, x, feat, N, Hc * Wc, Cout)
    ctx["recs"], ctx["final"] = recs, (Hc, Wc, Cout)
    return feat, ctx


def _bn_backward(rt, u, dout, mask, N, want_dz=False, train_params=True, fp8=None):
    """BN backward of unit u given dout (gradient w.r.t. the post-BN tensor) and the ReLU mask (packed bits, an activation tensor, or None).
    fp8: clite_bn.fp8_* triple - bn_bwd_apply also leaves the e5m2 copy of dy / records its amax (fp8.Fp8Forward.grad_producer)."""
    M, Cc = u.y.shape
    dstats = rt.new_stats(Cc, M)
    hip.bn_bwd_reduce(rt.dt, dout, mask, u.y, u.stats, dstats, M, Cc)
    dy = _alloc(rt, M, Cc)
    dz = _alloc(rt, M, Cc) if want_dz else None
    bn = u.bn
    desc = hip.bn_desc(M, Cc, u.stats, bn.weight, bn.bias, bn.running_mean, bn.running_var, True, False, bn.momentum, bn.eps, False, centered=rt.precise_bn, fp8=fp8)
    dg = rt.arena.g(bn.weight) if bn.weight.requires_grad else None
    db = rt.arena.g(bn.bias) if bn.bias.requires_grad else None
    hip.bn_bwd_apply(rt.dt, desc, dout, mask, u.y, dstats, dy, dz, dg, db)
    return dy, dz


def _bn_backward_apply(rt, u, dz, dstats, fp8=None):
    """BN backward of unit u given dz (gradient w.r.t. the BN output, ReLU mask already applied) and its two reductions `dstats`
    (sum dz, sum dz*(y - mean)), both produced by the epilogue of the GEMM that wrote dz."""
    M, Cc = u.y.shape
    bn = u.bn
    dy = _alloc(rt, M, Cc)
    desc = hip.bn_desc(M, Cc, u.stats, bn.weight, bn.bias, bn.running_mean, bn.running_var, True, False, bn.momentum, bn.eps, False, centered=rt.precise_bn, fp8=fp8)
    dg = rt.arena.g(bn.weight) if bn.weight.requires_grad else None
    db = rt.arena.g(bn.bias) if bn.bias.requires_grad else None
    hip.bn_bwd_apply(rt.dt, desc, dz, None, u.y, dstats, dy, None, dg, db)
    return dy


def _fold_wgrad(rt, group, u, asum, dz, f):
    """conv3's weight gradient with bn3's backward folded in: dW = diag(ka) (dz^T a) + kb (x) colsum(a) + diag(kc) W Cov(a) (include/clite.h, ABI v12). The
    first term and the Gram matrix a^T a are members of the grouped launch; the two correction terms follow it on its stream (WgradGroup.after): one small
    f32 kernel. asum: the column sums of a, left by the bn_apply that wrote it."""
    Mr, Kc, Cin = u.y.shape[0], u.y.shape[1], u.conv.in_channels
    dw = rt.arena.g(u.conv.weight)
    a = u.x
    group.conv(dz, a, u.cv, dw, row_scale=f.coef[0])
    G = rt.zpool.take(Cin * Cin)
    # the Gram matrix as a 1 x 1 "convolution" member (dy := a, x := a): the conv buckets have the 64-row tile that a 64 x 64 output wants
    group.conv(a, a, hip.conv_desc(rt.dt, 1, 1, Mr, Cin, Cin, 1, 1, 1, 0), G, short_k=True)
    wt = _wd(rt, u.conv)[0]
    group.keep += [f.coef, G, asum.t, wt]
    group.after(lambda: hip.bn_fold_wgrad_finish(G, asum, f.coef, wt, Mr, Kc, Cin, dw))


def resnet_backward(rt, net, ctx, dfeat, defer=None, stop_block=0, resume=False):
    """dfeat: [N][C] gradient of the pooled features (compute dtype). Accumulates parameter gradients into the arena.

    BatchNorm backward needs two per-channel reductions over the masked incoming gradient before it can produce its output. Where
    that gradient is written by a dgrad GEMM, the GEMM's epilogue applies the ReLU mask and accumulates both reductions while it
    stores (clite_epilogue.bn_y), so the separate reduction pass (3 tensor reads) disappears: inside a block for every unit but the
    last, and across blocks whenever the block-input gradient comes out of one kernel (identity shortcut).

    Only dgrad -> BN backward -> dgrad ... is a dependency chain; every weight-gradient GEMM needs just dy and the saved input and feeds
    nothing but the gradient arena. `defer` (a hip.WgradGroup) collects them instead of launching them, so the caller can run them together
    and elsewhere — the captured step (train_loop.TrainStep) launches the weight gradients of the late stages as ONE grouped launch on the
    text encoder's stream once BERT's backward has drained, beside the HBM-bound BatchNorm chain of the early stages. `stop_block` / `resume` cut the chain in two
    segments for that: the first call stops after block `stop_block` and parks (dout, pre) in ctx; `resume=True` continues from there."""
    N = ctx["N"]
    dt = rt.dt
    blocks = list(net.blocks())
    recs = ctx["recs"]
    own_group = None
    if defer is None and rt.group_wgrad and not rt.overlap_wgrad and not rt._capturing:      # (a capture cannot allocate the pinned staging)
        # uncaptured (eager / autograd) backward: the same grouped launch, issued at the end of this call on the same stream
        defer = own_group = hip.WgradGroup(rt.dt)
    staged = own_group is not None and getattr(rt, "exchange", None) is not None
    stage_first, pos = {}, 0         # index of the first block of each stage -> the stage (its gradients are final once that block is done)
    for layer in (net.layer1, net.layer2, net.layer3, net.layer4):
        stage_first[pos] = layer
        pos += len(layer)

    def wgrad(dy_, u_, dy8_=None):
        dw = rt.arena.g(u_.conv.weight)
        cv_ = u_.cv
        if defer is None:
            rt.aux_launch(lambda: hip.conv_wgrad(dy_, u_.x, cv_, dw), dy_)
        elif (f8 is not None and rt.fp8_wgrad and dy8_ is not None and u_.x8 is not None and cv_.K >= 256 and cv_.R * cv_.S * cv_.C >= 256
              and cv_.K % 16 == 0 and cv_.C % 16 == 0):
            # both operands exist in fp8 already (x: e4m3, written by the bn_apply in front for the fp8 forward; dy: e5m2, written by bn_bwd_apply for the
            # fp8 input gradient): the grouped 256 x 256 tile on the block-scaled MFMA (clite_wgrad_item kind 2) at half the operand bytes
            defer.conv_fp8(dy8_, u_.x8, cv_, dw)
        else:
            defer.conv(dy_, u_.x, cv_, dw)          # hip.WgradGroup: keeps dy / x referenced until it has been launched

    rt.arena.ensure_transposed(capturing=rt._capturing)
    # fp8 input gradients (DeviceRuntime.fp8_dgrad; fp8.Fp8Forward): the units' dgrads inside a block read the e5m2 copy of dy that the producing
    # bn_bwd_apply wrote and the e4m3 copy of the transposed weights (clite_conv_dgrad_fp8); weight gradients keep reading the bf16 dy
    from .fp8 import forward_state
    # (not in the deterministic-reduction mode: clite_conv_dgrad_fp8 accumulates its column sums with float atomics only)
    f8 = forward_state(rt, net) if (rt.fp8_dgrad and rt.fuse_bn_backward and rt.transposed_dgrad and not hip.is_deterministic()) else None
    if f8 is not None and f8.tgroup is None:
        f8 = None
    if f8 is not None and not resume:
        f8.begin_backward()

    def gq(u_, i_):
        """(fp8 triple for the bn_bwd_apply that writes unit u_'s dy, the e5m2 view its dgrad reads)"""
        if f8 is None or i_ == 0 or (rt.s2_classes and hip.s2_classes_ok(u_.cv)):
            return None, None
        return f8.grad_producer(u_.bn, u_.y.shape[0], u_.y.shape[1], f8.wants_dgrad(u_.conv, u_.cv.Ho))

    if resume:
        dout, pre, first = ctx.pop("bwd_state")
    else:
        Hc, Wc, Cout = ctx["final"]
        dout = _alloc(rt, N * Hc * Wc, Cout)
        hip.avgpool_bwd(dt, dfeat, dout, N, Hc * Wc, Cout)
        pre, first = None, len(recs) - 1          # pre: when set, `dout` is already masked by the block-output ReLU and `pre` holds the last unit's reductions
    for bi in range(first, stop_block - 1, -1):
        units, ud, Hin, Win = recs[bi]
        last = units[-1]
        identity = ud is None
        # block output = relu(bn_last(y_last) + shortcut): the mask is the block output itself
        dyd = None
        folded = None
        q8, dy8 = gq(last, len(units) - 1)
        if pre is None:
            dy, dz = _bn_backward(rt, last, dout, last.bits, N, want_dz=identity, fp8=q8)
            if ud is not None:
                dyd, _ = _bn_backward(rt, ud, dout, last.bits, N)
        else:
            dz = dout
            if (last.pair is not None and units[-2].xsum is not None and q8 is None and defer is not None and dz.data_ptr() == last.pair.data_ptr()
                    and not hip.is_deterministic() and _wd(rt, last.conv)[1]):
                # fold bn3's backward into conv3's two gradients: no apply pass, no dy (205 MB per layer1 block at batch 128)
                bn3 = last.bn
                desc = hip.bn_desc(last.y.shape[0], last.y.shape[1], last.stats, bn3.weight, bn3.bias, bn3.running_mean, bn3.running_var, True, False, bn3.momentum,
                                   bn3.eps, False, centered=rt.precise_bn)
                folded = hip.bn_fold_prepare(desc, pre, _wd(rt, last.conv)[0], last.conv.in_channels,
                                             rt.arena.g(bn3.weight) if bn3.weight.requires_grad else None, rt.arena.g(bn3.bias) if bn3.bias.requires_grad else None)
                dy = None
            else:
                dy = _bn_backward_apply(rt, last, dz, pre, fp8=q8)
            if ud is not None:
                dyd, _ = _bn_backward(rt, ud, dz, None, N)
        pre = None
        for i in range(len(units) - 1, -1, -1):
            u = units[i]
            if folded is not None and i == len(units) - 1:
                # conv3 with its BatchNorm's backward folded in (hip.bn_fold_prepare): dy is never formed. Weight gradient = ka . (dz^T a) as the group's
                # member (row_scale) + the two correction terms behind the group; input gradient = [dz | y] [ka W ; kc W] + a constant row through the
                # usual BatchNorm-backward epilogue (the mask and reductions of bn2)
                Mr, Kc, Cin = u.y.shape[0], u.y.shape[1], u.conv.in_channels
                if u.conv.weight.requires_grad:
                    _fold_wgrad(rt, defer, u, units[i - 1].xsum, dz, folded)
                prev = units[i - 1]
                dstats = rt.new_stats(Cin, prev.y.shape[0])
                dx = _alloc(rt, u.x.shape[0], Cin)
                hip.conv_dgrad_bnfold(u.pair, folded.w2, Mr, Kc, Cin,
                                      hip.epilogue(dx, Cin, relu_bits=prev.bits, colsum=dstats, bn=(prev.y, prev.stats, prev.y.shape[0]), bias=folded.bias))
                q8, dy8 = gq(prev, i - 1)
                dy = _bn_backward_apply(rt, prev, dx, dstats, fp8=q8)
                continue
            if u.conv.weight.requires_grad:
                wgrad(dy, u, dy8)
            Cin = u.conv.in_channels
            # the gradient this block hands to the previous one goes straight into slot 0 of that block's pair buffer (beside its y3) when it folds
            pl_pair = recs[bi - 1][0][-1].pair if (i == 0 and bi > 0) else None
            dx = pl_pair[0] if pl_pair is not None else _alloc(rt, u.x.shape[0], Cin)
            wd, wt = _wd(rt, u.conv)
            if i > 0 and not rt.fuse_bn_backward:
                hip.conv_dgrad(dy, wd, u.cv, hip.epilogue(dx, Cin), wt=wt)
                prev = units[i - 1]
                dy, _ = _bn_backward(rt, prev, dx, prev.bits, N)
            elif i > 0:
                prev = units[i - 1]
                dstats = rt.new_stats(Cin, prev.y.shape[0])
                mk = lambda: hip.epilogue(dx, Cin, relu_bits=prev.bits, colsum=dstats, bn=(prev.y, prev.stats, prev.y.shape[0]))
                if rt.s2_classes and hip.s2_classes_ok(u.cv):
                    hip.conv_dgrad_s2(dy, wd, u.cv, mk,       # 3x3 / stride 2: four parity classes, no zero taps
                                      wsubs=(ctx.get("s2w") or {}).get(id(u.conv)), wt=wt)
                elif dy8 is not None:
                    hip.conv_dgrad_fp8(dy8, f8.weight_t(u.conv), u.cv, mk())
                else:
                    hip.conv_dgrad(dy, wd, u.cv, mk(), wt=wt)
                q8, dy8 = gq(prev, i - 1)
                dy = _bn_backward_apply(rt, prev, dx, dstats, fp8=q8)
            else:
                # gradient w.r.t. the block input: main path + shortcut
                if identity and bi > 0 and rt.fuse_bn_backward:
                    pl = recs[bi - 1][0][-1]          # the previous block's last unit consumes this gradient
                    pre = rt.new_stats(Cin, pl.y.shape[0])
                    hip.conv_dgrad(dy, wd, u.cv,
                                   hip.epilogue(dx, Cin, residual=dz, relu_bits=pl.bits, mask_after_residual=True, colsum=pre,
                                                bn=(pl.y, pl.stats, pl.y.shape[0])), wt=wt)
                elif (ud is not None and bi > 0 and rt.fuse_bn_backward and rt.lowp and wt and ud.cv.stride == 2 and ud.cv.R == 1 and u.cv.R == 1
                      and u.cv.stride == 1 and u.cv.H % 2 == 0 and u.cv.W % 2 == 0 and not hip.is_deterministic() and rt.compact_shortcut
                      and _wd(rt, ud.conv)[1]):          # (the shortcut's GEMM below reads the TRANSPOSED copy as [Cin][K]: it must exist)
                    # First block of a stride-2 stage (Bottleneck): the block-input gradient is conv1's dgrad (dense, full resolution) plus the
                    # strided 1 x 1 shortcut's dgrad, which only reaches every second pixel. The shortcut's gradient stays COMPACT ([N][H/2][W/2][Cin],
                    # one forward-form GEMM) and enters conv1's dgrad as a row-mapped residual (clite_epilogue.residual_subsample) of the
                    # BatchNorm-backward epilogue — so the gradient reaching the previous block is already masked and comes with the two
                    # reductions of that block's last BatchNorm, like behind every identity block: no scatter-add read-modify-write over the
                    # full-size gradient and no separate reduction pass (3 of the step's 11 bn_bwd_reduce launches, the largest ones)
                    pl = recs[bi - 1][0][-1]
                    if ud.conv.weight.requires_grad:
                        wgrad(dyd, ud)
                    wdd, _ = _wd(rt, ud.conv)          # [Cin][1][1][K]
                    P, Kd = dyd.shape[0], ud.cv.K
                    dsc = _alloc(rt, P, Cin)
                    hip.gemm_nt(dt, dyd, wdd.view(Cin, Kd), P, Cin, Kd, hip.epilogue(dsc, Cin))
                    pre = rt.new_stats(Cin, pl.y.shape[0])
                    hip.conv_dgrad(dy, wd, u.cv,
                                   hip.epilogue(dx, Cin, residual=dsc, residual_subsample=2, relu_bits=pl.bits, mask_after_residual=True, colsum=pre,
                                                bn=(pl.y, pl.stats, pl.y.shape[0])), wt=wt)
                else:
                    # the network's first block: its input gradient is the pooled gradient of the stem. With the stem's pooled-size operands
                    # (resnet_forward: ymax, sbits) the LAST launch that writes it also masks it by relu'(pooled output) and accumulates bn1's two
                    # backward reductions (the BatchNorm-backward epilogue with bn_y := ymax): stem_pre replaces stem_bn_pool_bwd's reduction pass
                    # over the 4 x larger un-pooled tensors
                    ymax, sbits = ctx["stem"][10:12]
                    stem_ep = None
                    if bi == 0 and ymax is not None and wt and (ud is None or (ud.cv.stride == 1 and _wd(rt, ud.conv)[1])):
                        ctx["stem_pre"] = rt.new_stats(Cin, ctx["stem"][5].shape[0])
                        rows0 = ctx["stem"][5].shape[0]          # N * Ho * Wo of the un-pooled tensor: the count behind bn1's mean
                        stem_ep = lambda res: hip.epilogue(dx, Cin, residual=res, relu_bits=sbits, mask_after_residual=True, colsum=ctx["stem_pre"],
                                                           bn=(ymax, ctx["stem"][6], rows0))
                    if stem_ep is not None and ud is None:
                        hip.conv_dgrad(dy, wd, u.cv, stem_ep(dz), wt=wt)
                    else:
                        hip.conv_dgrad(dy, wd, u.cv, hip.epilogue(dx, Cin, residual=dz if identity else None), wt=wt)
                    if ud is not None:
                        if ud.conv.weight.requires_grad:
                            wgrad(dyd, ud)
                        # shortcut branch accumulated in place (dx += dgrad); a strided 1x1 shortcut takes the scatter-add path of clite_conv_dgrad
                        wdd, wtd = _wd(rt, ud.conv)
                        hip.conv_dgrad(dyd, wdd, ud.cv, stem_ep(dx) if stem_ep is not None else hip.epilogue(dx, Cin, residual=dx), wt=wtd)
                dout = dx
        if defer is None:
            rt.grads_ready(blocks[bi])
        elif staged and bi in stage_first:
            # eager data parallel: the grouped weight gradients of a finished STAGE are launched now and its region handed to the
            # gradient exchange, so that the all-reduce overlaps the remaining backward instead of starting after the whole encoder
            own_group.launch()
            rt.grads_ready(stage_first[bi])
            defer = own_group = hip.WgradGroup(rt.dt)
    if stop_block > 0:
        assert own_group is None
        ctx["bwd_state"] = (dout, pre, stop_block - 1)
        return
    if f8 is not None:
        f8.end_backward()
    xpad, Hp, Wp, Ho, Wo, y0, st0, idx, Hq, Wq = ctx["stem"][:10]
    # max-pool backward + ReLU mask + BatchNorm backward straight from (dpool, idx, y0): neither the un-pooled gradient nor the mask is stored
    bn1 = net.bn1
    stem_pre = ctx.pop("stem_pre", None)

    def stem_tail(dout=dout, stem_pre=stem_pre):
        """bn1's backward (the un-pooled gradient dy0) and conv1's weight gradient, its only reader. Nothing on the dependent chain needs either: with
        a deferring caller (the captured step) the whole tail joins the collected stand-alone launches, which replay on the side stream beside the
        last weight-gradient group (DeviceRuntime.stem_tail_deferred; 165 us of bn_bwd_apply off the chain)."""
        desc0 = hip.bn_desc(N * Ho * Wo, 64, st0, bn1.weight, bn1.bias, bn1.running_mean, bn1.running_var, True, False, bn1.momentum, bn1.eps, False,
                            centered=rt.precise_bn)
        dg = rt.arena.g(bn1.weight) if bn1.weight.requires_grad else None
        db = rt.arena.g(bn1.bias) if bn1.bias.requires_grad else None
        dst0 = rt.new_stats(64, N * Ho * Wo)
        dy0 = _alloc(rt, N * Ho * Wo, 64)
        if stem_pre is not None:          # the reductions came out of the epilogue that wrote `dout` (already masked by relu'(pooled output))
            hip.stem_bn_pool_bwd_apply(dt, desc0, dout, idx, y0, stem_pre, dy0, dg, db, N, Ho, Wo)
        else:
            hip.stem_bn_pool_bwd(dt, desc0, dout, idx, y0, dst0, dy0, dg, db, N, Ho, Wo)
        if not net.conv1.weight.requires_grad:
            return
        if rt.stem_wgrad_patch and rt.lowp and hip.stem_wgrad_patch(dt, dy0, xpad, N, Hp, Wp, Ho, Wo, rt.arena.g(net.conv1.weight)):
            return
        dwv = torch.zeros(64, 7, 8, 4, device=rt.device, dtype=torch.float32)
        hip.stem_wgrad(dt, dy0, xpad, N, Hp, Wp, Ho, Wo, dwv)
        hip.stem_unpack_grad(dwv, rt.arena.g(net.conv1.weight))

    if defer is None or not rt.stem_tail_deferred:
        stem_tail()
    else:
        defer.call(stem_tail)
    if own_group is not None:
        own_group.launch()
        if staged:                    # the four stages were handed over as their groups were launched: only the stem is left
            rt.grads_ready(net.conv1)
            rt.grads_ready(net.bn1)
        else:
            rt.grads_ready(net)       # every gradient of the encoder became final with that launch
    elif defer is None:
        rt.join_aux()
        rt.grads_ready(net.conv1)
        rt.grads_ready(net.bn1)
