"""The fp8 (OCP e4m3) forward of the image encoder — BASELINE.json configs[4]; the reference has no fp8 code (encoder.py:36-41 is the call site
whose convolutions this replaces), so the policy is this build's:

* which convolutions: the windowed (3x3) ones and the 1x1 ones at <= 14 x 14 — the launches bound by L2 -> LDS operand traffic, which e4m3 halves.
  The 1x1 convolutions of the 56 x 56 / 28 x 28 stages are HBM-bound on tensors that are written once and read once: an e4m3 copy costs the
  half-write it saves as a half-read, so they stay bf16. The stem, the heads and every backward GEMM stay bf16.
* weights: per-tensor CURRENT scaling, all eligible conv weights in two launches per step (hip.Fp8WeightGroup over the bf16 arena).
* activations: per-tensor DELAYED scaling, quantised by their producer — `bn_apply` writes the e4m3 copy beside the bf16 tensor with the scale
  made from the previous step's amax and records this step's amax (clite_bn.fp8_*); one `clite_fp8_scale_update` per step turns the amaxes
  into next step's scales. A tensor without a scale yet (the first step) is quantised by the stand-alone `clite_fp8_quantize` (current scaling)."""
import torch

from . import hip


def conv_eligible(conv):
    """Channel granularity of the e4m3 operand loaders (include/clite.h: clite_conv_fwd_fp8)."""
    return conv.in_channels % (64 if conv.k > 1 else 16) == 0


def dgrad_eligible(conv):
    """clite_conv_dgrad_fp8: stride 1, output channels (the contraction index) a multiple of 64."""
    return conv.stride == 1 and conv.out_channels % 64 == 0 and conv.in_channels % 8 == 0


class Fp8Forward:
    """Per-network state of the fp8 forward: the weight group, one (amax, scales) slot per BatchNorm whose output an fp8 convolution reads."""

    def __init__(self, rt, net):
        self.rt = rt
        convs, bns = [], []
        for blk in net.blocks():
            for conv, bn in blk.units():
                convs.append(conv)
                bns.append(bn)
            if blk.downsample is not None:
                convs.append(blk.downsample[0])
        self.windex = {}
        spans = []
        for conv in convs:
            if conv_eligible(conv):
                self.windex[id(conv)] = (len(spans), conv)
                spans.append(rt.arena.index[conv.weight._clite[1]])
        self.wgroup = hip.Fp8WeightGroup(rt.arena.flat_lp, spans) if spans else None
        self.slot = {id(bn): i for i, bn in enumerate(bns)}
        n = max(len(bns), 1)
        self.amax = torch.zeros(n, hip.FP8_AMAX_WORDS, dtype=torch.float32, device=rt.device)          # one slot of replicated words per tensor
        self.scales = torch.ones(n, 2, dtype=torch.float32, device=rt.device)
        self.ready, self._seen = set(), set()
        # fp8 input gradients (DeviceRuntime.fp8_dgrad): e4m3 copies of the TRANSPOSED weights (same offsets in Arena.flat_lpT) and one e5m2 slot per
        # BatchNorm for the gradient its bn_bwd_apply writes (the dy of the unit's conv)
        self.tindex = {}
        tspans = []
        for conv in convs:
            if conv_eligible(conv) and dgrad_eligible(conv) and rt.arena.has_wt(conv.weight):
                self.tindex[id(conv)] = (len(tspans), conv)
                tspans.append(rt.arena.index[conv.weight._clite[1]])
        self.tgroup = hip.Fp8WeightGroup(rt.arena.flat_lpT, tspans) if tspans else None
        self.gamax = torch.zeros(n, hip.FP8_AMAX_WORDS, dtype=torch.float32, device=rt.device)
        self.gscales = torch.ones(n, 2, dtype=torch.float32, device=rt.device)
        self.gready, self._gseen = set(), set()

    # ---- policy
    def wants(self, conv, H):
        """Does `conv`, reading an H x H (or smaller) input, run on e4m3 operands?"""
        return id(conv) in self.windex and (conv.k > 1 or H <= 14)

    # ---- per step
    def begin_step(self):
        if self.wgroup is not None:
            self.wgroup.quantize()

    def weight(self, conv):
        i, _ = self.windex[id(conv)]
        return self.wgroup.view(i, self.rt.arena.w(conv.weight).shape)

    def producer(self, bn, M, Cc, want, training=True):
        """(clite_bn.fp8_* triple for bn_apply, the Fp8View its consumers read) for the output of `bn`; (None, None) when no fp8 conv reads it.
        Eval-mode forwards (training=False) use the scales READ-ONLY: they record no amax and never make a slot ready, so a validation batch
        neither changes the scales the next training step quantises with nor depends on more than the training state it finds."""
        if not want:
            return None, None
        s = self.slot[id(bn)]
        if not training:
            if s not in self.ready:      # no training step has made a scale yet: the consumer quantises with current scaling
                return None, None
            q = torch.empty(M, Cc, dtype=torch.uint8, device=self.rt.device)
            return (q, self.scales[s], None), hip.Fp8View(q, self.scales[s])
        self._seen.add(s)
        amax = self.amax[s]
        if s not in self.ready:          # no scale yet: record the amax only; the consumer quantises the bf16 tensor itself this once
            return (None, None, amax), None
        q = torch.empty(M, Cc, dtype=torch.uint8, device=self.rt.device)
        return (q, self.scales[s], amax), hip.Fp8View(q, self.scales[s])

    def end_step(self):
        if self._seen:
            hip.fp8_scale_update(self.amax, self.scales)
            self.ready |= self._seen
            self._seen = set()

    # ---- backward (fp8 input gradients)
    def wants_dgrad(self, conv, H):
        """Does the input gradient of `conv` (output H x H) run on fp8 operands? The windowed stride-1 convs of >= 128 channels and the 1 x 1 ones at
        <= 14 x 14 - the forward's policy minus the 64-channel 3 x 3 at 56 x 56, whose input gradient is the HBM-bound patch-resident kernel."""
        return id(conv) in self.tindex and ((conv.k > 1 and conv.out_channels >= 128) or (conv.k == 1 and H <= 14))

    def begin_backward(self):
        """After Arena.ensure_transposed, before the first fp8 input gradient: this step's e4m3 copies of the transposed weights."""
        if self.tgroup is not None:
            self.tgroup.quantize()

    def weight_t(self, conv):
        i, _ = self.tindex[id(conv)]
        o, n = self.tgroup.spans[i]
        return hip.Fp8View(self.tgroup.q[o:o + n], self.tgroup.scales[i])

    def grad_producer(self, bn, M, Cc, want):
        """(clite_bn.fp8_* triple for bn_bwd_apply, the Fp8View clite_conv_dgrad_fp8 reads) for the gradient bn's backward writes; (None, None) when
        its consumer stays bf16. Delayed scaling as for the activations: the first step records the amax only."""
        if not want:
            return None, None
        s = self.slot[id(bn)]
        self._gseen.add(s)
        if s not in self.gready:
            return (None, None, self.gamax[s]), None
        q = torch.empty(M, Cc, dtype=torch.uint8, device=self.rt.device)
        return (q, self.gscales[s], self.gamax[s]), hip.Fp8View(q, self.gscales[s])

    def end_backward(self):
        if self._gseen:
            hip.fp8_scale_update(self.gamax, self.gscales)
            self.gready |= self._gseen
            self._gseen = set()


class Fp8Text:
    """The fp8 forward of the text encoder (DeviceRuntime.fp8_text, bf16 mode): BERT's QKV projection, FFN1 and FFN2 of every layer on e4m3 operands
    (clite_gemm_nt_fp8; the 768 x 768 attention output projection stays bf16: its A operand comes out of the attention kernel, and the launch is
    the smallest of the four). Weights: per-tensor current scaling, all 36 matrices in the two launches of one hip.Fp8WeightGroup per step (q / k / v
    as ONE tensor: the fused projection has one B scale). Activations: delayed scaling, quantised by their producers - the LayerNorm forward
    (clite_layernorm_fwd_q8: the layer input for QKV, the attention block's output for FFN1) and FFN1's own epilogue (clite_epilogue.fp8_*: the
    GELU output for FFN2); slot 3 l + {0, 1, 2} of layer l. A tensor without a scale yet (the first step) is recorded only and its consumer runs
    in bf16 that once."""

    def __init__(self, rt, net):
        self.rt = rt
        A = rt.arena
        spans = []
        self.windex = {}
        for l, layer in enumerate(net.encoder.layer):
            sa = layer.attention.self
            qkv = [A.index[p._clite[1]] for p in (sa.query.weight, sa.key.weight, sa.value.weight)]
            for (o, n), (o2, _) in zip(qkv[:-1], qkv[1:]):
                assert o + n == o2, "q / k / v weights are not contiguous in the arena"
            self.windex[(l, 0)] = len(spans)
            spans.append((qkv[0][0], sum(n for _, n in qkv)))
            for j, lin in ((1, layer.intermediate.dense), (2, layer.output.dense)):
                self.windex[(l, j)] = len(spans)
                spans.append(A.index[lin.weight._clite[1]])
        self.wgroup = hip.Fp8WeightGroup(A.flat_lp, spans)
        n = 3 * len(net.encoder.layer)
        self.amax = torch.zeros(n, hip.FP8_AMAX_WORDS, dtype=torch.float32, device=rt.device)
        self.scales = torch.ones(n, 2, dtype=torch.float32, device=rt.device)
        self.ready, self._seen = set(), set()

    def begin_step(self):
        self.wgroup.quantize()

    def weight(self, l, j, shape):
        return self.wgroup.view(self.windex[(l, j)], shape)

    def producer(self, l, j, M, Cc, training=True):
        """((q, scales, amax) for the producer's fp8 arguments, the Fp8View the consumer reads - None: run it in bf16) for activation j of layer l.
        Eval-mode forwards use the scales read-only, as Fp8Forward.producer."""
        s = 3 * l + j
        if not training:
            if s not in self.ready:
                return None, None
            q = torch.empty(M, Cc, dtype=torch.uint8, device=self.rt.device)
            return (q, self.scales[s], None), hip.Fp8View(q, self.scales[s])
        self._seen.add(s)
        if s not in self.ready:
            return (None, None, self.amax[s]), None
        q = torch.empty(M, Cc, dtype=torch.uint8, device=self.rt.device)
        return (q, self.scales[s], self.amax[s]), hip.Fp8View(q, self.scales[s])

    def end_step(self):
        if self._seen:
            hip.fp8_scale_update(self.amax, self.scales)
            self.ready |= self._seen
            self._seen = set()


def text_state(rt, net):
    """The text encoder's Fp8Text when DeviceRuntime.fp8_text is on (bf16 mode only), else None."""
    if not (rt.fp8_text and rt.lowp):
        return None
    st = rt.fp8_nets.get(id(net))
    if st is None:
        st = rt.fp8_nets[id(net)] = Fp8Text(rt, net)
    return st


def forward_state(rt, net):
    """The network's Fp8Forward when the runtime's fp8 forward is on (bf16 mode only), else None."""
    if not (rt.fp8 and rt.lowp):
        return None
    st = rt.fp8_nets.get(id(net))
    if st is None:
        st = rt.fp8_nets[id(net)] = Fp8Forward(rt, net)
    return st
