"""ORACLE — TEST INFRASTRUCTURE ONLY. CPU (plain PyTorch, fp32) restatement of the CLIP-Lite pretraining step.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product package
(clip-lite_amd/) never does. Each class cites the reference code it restates (paths relative to 4m4n5/CLIP-Lite).

Parity status: PINNED for loss.py / optim/* / model.py / encoder.TextEncoder by golden vectors generated in the build
container by importing the reference itself (tests/golden/make_golden.py -> tests/golden/*.npz; checked by
tests/test_oracle_golden.py). The image encoder is torchvision==0.8.0's ResNet (requirements.txt:97), which is NOT
vendored in the reference and NOT installed in the build image: `OracleResNet` restates its published topology
(conv7x7/2-BN-ReLU-maxpool3x3/2, BasicBlock/Bottleneck with the stride on the 3x3, 1x1-stride downsample + BN,
adaptive avg-pool, fc = Identity per encoder.py:41) directly on torch.nn.Conv2d/BatchNorm2d — the same ATen ops
torchvision calls — and is cross-checked structurally against the reference's own key map (encoder.py:84-94) and
parameter counts; for that component parity is "unpinned by the reference".
"""
import math
import re
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------------ image encoder
class _BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + identity)


class _Bottleneck(nn.Module):
    """1x1 -> 3x3(stride) -> 1x1(x4), post-add ReLU; same arithmetic as reference model_zoo/resnet.py:60-100."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return self.relu(out + identity)


RESNET_SPECS = {
    "resnet18": (_BasicBlock, (2, 2, 2, 2)), "resnet34": (_BasicBlock, (3, 4, 6, 3)),
    "resnet50": (_Bottleneck, (3, 4, 6, 3)), "resnet101": (_Bottleneck, (3, 4, 23, 3)),
    "resnet152": (_Bottleneck, (3, 8, 36, 3)),
}


class OracleResNet(nn.Module):
    """torchvision.models.resnet*(pretrained=False, zero_init_residual=False) with fc = nn.Identity()
    (reference encoder.py:36-41). Init: conv kaiming_normal_(fan_out, relu); BN weight 1, bias 0."""

    def __init__(self, name="resnet50"):
        super().__init__()
        block, layers = RESNET_SPECS[name]
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make_layer(block, 64, layers[0], 1)
        self.layer2 = self._make_layer(block, 128, layers[1], 2)
        self.layer3 = self._make_layer(block, 256, layers[2], 2)
        self.layer4 = self._make_layer(block, 512, layers[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Identity()
        self.out_dim = 512 * block.expansion
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, block, planes, blocks, stride):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False),
                                       nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return nn.Sequential(*layers)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))


class OracleImageEncoder(nn.Module):
    """reference encoder.py:13-65"""

    def __init__(self, img_enc_net="resnet50", pretrained=False, frozen=False):
        super().__init__()
        self.img_encoder = OracleResNet(img_enc_net)
        if frozen:
            for p in self.img_encoder.parameters():
                p.requires_grad = False
            self.img_encoder.eval()

    def forward(self, image):
        x = self.img_encoder(image)
        return x.view(x.size(0), x.size(1))


# ------------------------------------------------------------------------------------------------ text encoder
class _BertSelfAttention(nn.Module):
    def __init__(self, hidden, heads, p):
        super().__init__()
        self.heads = heads
        self.query = nn.Linear(hidden, hidden)
        self.key = nn.Linear(hidden, hidden)
        self.value = nn.Linear(hidden, hidden)
        self.dropout = nn.Dropout(p)

    def forward(self, h, ext_mask):
        B, L, Hd = h.shape
        split = lambda t: t.view(B, L, self.heads, Hd // self.heads).permute(0, 2, 1, 3)
        q, k, v = split(self.query(h)), split(self.key(h)), split(self.value(h))
        s = q @ k.transpose(-1, -2) / math.sqrt(Hd // self.heads) + ext_mask
        p = self.dropout(F.softmax(s, dim=-1))
        return (p @ v).permute(0, 2, 1, 3).reshape(B, L, Hd)


class _BertSelfOutput(nn.Module):
    def __init__(self, hidden, inner, p, eps):
        super().__init__()
        self.dense = nn.Linear(inner, hidden)
        self.LayerNorm = nn.LayerNorm(hidden, eps=eps)
        self.dropout = nn.Dropout(p)

    def forward(self, x, residual):
        return self.LayerNorm(self.dropout(self.dense(x)) + residual)


class _BertAttention(nn.Module):
    def __init__(self, hidden, heads, p, eps):
        super().__init__()
        self.self = _BertSelfAttention(hidden, heads, p)
        self.output = _BertSelfOutput(hidden, hidden, p, eps)


class _BertIntermediate(nn.Module):
    def __init__(self, hidden, inner):
        super().__init__()
        self.dense = nn.Linear(hidden, inner)


class _BertLayer(nn.Module):
    def __init__(self, hidden, heads, inner, p, eps):
        super().__init__()
        self.attention = _BertAttention(hidden, heads, p, eps)
        self.intermediate = _BertIntermediate(hidden, inner)
        self.output = _BertSelfOutput(hidden, inner, p, eps)

    def forward(self, h, ext_mask):
        a = self.attention.output(self.attention.self(h, ext_mask), h)
        return self.output(F.gelu(self.intermediate.dense(a)), a)


class _BertEmbeddings(nn.Module):
    def __init__(self, vocab, hidden, max_pos, p, eps):
        super().__init__()
        self.word_embeddings = nn.Embedding(vocab, hidden, padding_idx=0)
        self.position_embeddings = nn.Embedding(max_pos, hidden)
        self.token_type_embeddings = nn.Embedding(2, hidden)
        self.LayerNorm = nn.LayerNorm(hidden, eps=eps)
        self.dropout = nn.Dropout(p)

    def forward(self, ids):
        L = ids.shape[1]
        e = self.word_embeddings(ids) + self.position_embeddings.weight[:L] + self.token_type_embeddings.weight[0]
        return self.dropout(self.LayerNorm(e))


class _BertEncoder(nn.Module):
    def __init__(self, n, *a):
        super().__init__()
        self.layer = nn.ModuleList([_BertLayer(*a) for _ in range(n)])


class _BertPooler(nn.Module):
    def __init__(self, hidden):
        super().__init__()
        self.dense = nn.Linear(hidden, hidden)


class OracleBert(nn.Module):
    """transformers.BertModel(BertConfig(num_hidden_layers=n)) as built at reference encoder.py:165-170: post-LN encoder,
    hidden 768, 12 heads, FFN 3072 GELU(erf), LayerNorm eps 1e-12, dropout 0.1 (embeddings, attention probs, both
    sub-layer outputs), additive mask (1-m)*finfo.min, pooler tanh(W h[:,0] + b). Init: N(0, 0.02) for Linear/Embedding
    weights (padding row 0 of word embeddings zeroed), zeros for biases, ones/zeros for LayerNorm."""

    def __init__(self, num_hidden_layers=12, hidden=768, heads=12, inner=3072, vocab=30522, max_pos=512, p=0.1, eps=1e-12):
        super().__init__()
        self.embeddings = _BertEmbeddings(vocab, hidden, max_pos, p, eps)
        self.encoder = _BertEncoder(num_hidden_layers, hidden, heads, inner, p, eps)
        self.pooler = _BertPooler(hidden)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                m.weight.data.normal_(0.0, 0.02)
                m.bias.data.zero_()
            elif isinstance(m, nn.Embedding):
                m.weight.data.normal_(0.0, 0.02)
                if m.padding_idx is not None:
                    m.weight.data[m.padding_idx].zero_()
            elif isinstance(m, nn.LayerNorm):
                m.weight.data.fill_(1.0)
                m.bias.data.zero_()

    def forward(self, input_ids, attention_mask):
        ext = (1.0 - attention_mask[:, None, None, :].to(torch.float32)) * torch.finfo(torch.float32).min
        h = self.embeddings(input_ids)
        for layer in self.encoder.layer:
            h = layer(h, ext)
        return torch.tanh(self.pooler.dense(h[:, 0]))


class OracleTextEncoder(nn.Module):
    """reference encoder.py:115-205 for mode in {"sbert", "train_sbert"} with a *bert* model_name (pooler_output path)."""

    def __init__(self, word_dict=None, mode="train_sbert", transform_embedding=False, txt_enc_dim=512, num_hidden_layers=12, **_):
        super().__init__()
        self.mode = mode
        self.transform_embedding = transform_embedding
        if mode == "train_sbert":
            self.strans = OracleBert(num_hidden_layers)
        elif mode != "sbert":
            raise NotImplementedError(mode)
        if transform_embedding:
            self.fc1 = nn.Linear(768, txt_enc_dim)
            self.fc2 = nn.Linear(txt_enc_dim, txt_enc_dim)
            self.relu = nn.ReLU()

    def forward(self, x):
        if self.mode == "train_sbert":
            x = self.strans(x["input_ids"], x["attention_mask"])
        if self.transform_embedding:
            x = self.fc2(self.relu(self.fc1(x)))
        return x


# ------------------------------------------------------------------------------------------------ loss (reference loss.py)
class MILinearBlock(nn.Module):
    """loss.py:12-40"""

    def __init__(self, feature_sz, units=2048, bln=True):
        super().__init__()
        self.feature_nonlinear = nn.Sequential(nn.Linear(feature_sz, units, bias=False), nn.BatchNorm1d(units), nn.ReLU(),
                                               nn.Linear(units, units))
        self.feature_shortcut = nn.Linear(feature_sz, units)
        self.feature_block_ln = nn.LayerNorm(units)
        eye = torch.zeros(units, feature_sz, dtype=torch.bool)
        idx = torch.arange(min(units, feature_sz))
        eye[idx, idx] = True
        self.feature_shortcut.weight.data.uniform_(-0.01, 0.01)
        self.feature_shortcut.weight.data.masked_fill_(eye, 1.0)
        self.bln = bln

    def forward(self, feat):
        f = self.feature_nonlinear(feat) + self.feature_shortcut(feat)
        return self.feature_block_ln(f) if self.bln else f


class PriorDiscriminator(nn.Module):
    """loss.py:43-53"""

    def __init__(self, sz):
        super().__init__()
        self.l0 = nn.Linear(sz, 1000)
        self.l1 = nn.Linear(1000, 200)
        self.l2 = nn.Linear(200, 1)

    def forward(self, x):
        return torch.sigmoid(self.l2(F.relu(self.l1(F.relu(self.l0(x))))))


class GlobalDiscriminatorDot(nn.Module):
    """loss.py:76-107"""

    def __init__(self, image_sz, text_sz, units=2048, bln=True):
        super().__init__()
        self.img_block = MILinearBlock(image_sz, units=units, bln=bln)
        self.text_block = MILinearBlock(text_sz, units=units, bln=bln)
        self.temperature = nn.Parameter(torch.ones([]) * np.log(1 / 0.07))

    def forward(self, features1=None, features2=None):
        f1 = F.normalize(self.img_block(features1), p=2, dim=-1)
        f2 = F.normalize(self.text_block(features2), p=2, dim=-1)
        return (f1 * f2).sum(-1) * self.temperature.exp()


class GlobalDiscriminator(nn.Module):
    """loss.py:56-68 — the `concat` critic: an MLP on cat(features1, features2)."""

    def __init__(self, sz):
        super().__init__()
        self.l0 = nn.Linear(sz, 512)
        self.l1 = nn.Linear(512, 512)
        self.l2 = nn.Linear(512, 1)

    def forward(self, features1=None, features2=None):
        x = torch.cat((features1, features2), dim=1)
        return self.l2(F.relu(self.l1(F.relu(self.l0(x)))))


def _roll1(t):
    return torch.cat((t[1:], t[0].unsqueeze(0)), dim=0)


class OracleJSDInfoMaxLoss(nn.Module):
    """loss.py:110-314: critic types dot / concat / condot / dotcon (:129-169), the cluster hard-negative branch (:225-252) and the
    visual / textual self-supervised terms (:256-300); device-agnostic (the reference's `.cuda()` literals at loss.py:186,257,280 are
    dropped). `noise` optionally pins the two torch.rand_like draws (loss.py:189,196) in their draw order (image, text)."""

    def __init__(self, image_dim=2048, text_dim=768, type="dot", prior_weight=0.1, image_prior=True, text_prior=False,
                 visual_self_supervised=False, textual_self_supervised=False, **_):
        super().__init__()
        assert type in ("dot", "concat", "condot", "dotcon")
        self.prior_weight, self.image_prior, self.text_prior = prior_weight, image_prior, text_prior
        cross_dot = type in ("dot", "dotcon")
        ssl_dot = type in ("dot", "condot")
        self.global_d = GlobalDiscriminatorDot(image_dim, text_dim) if cross_dot else GlobalDiscriminator(image_dim + text_dim)
        if visual_self_supervised:
            self.visual_d = GlobalDiscriminatorDot(image_dim, image_dim) if ssl_dot else GlobalDiscriminator(2 * image_dim)
        if textual_self_supervised:
            self.textual_d = GlobalDiscriminatorDot(text_dim, text_dim) if ssl_dot else GlobalDiscriminator(2 * text_dim)
        if image_prior:
            self.prior_d = PriorDiscriminator(image_dim)
        if text_prior:
            self.text_prior_d = PriorDiscriminator(text_dim)
        self.noise = None

    @staticmethod
    def _pair_terms(critic, f1, f2_pos, f2_neg):
        ej = -F.softplus(-critic(features1=f1, features2=f2_pos)).mean()
        em = F.softplus(critic(features1=f1, features2=f2_neg)).mean()
        return em - ej

    def forward(self, image_features, text_features, neg_image_features=None, neg_text_features=None,
                aug_image_features=None, aug_text_features=None):
        prior = image_features.new_zeros(())
        if self.image_prior:
            u = torch.rand_like(image_features) if self.noise is None else self.noise[0]
            prior = prior - (torch.log(self.prior_d(u)).mean() + torch.log(1.0 - self.prior_d(image_features)).mean())
        if self.text_prior:
            u = torch.rand_like(text_features) if self.noise is None else self.noise[1]
            prior = prior - (torch.log(self.text_prior_d(u)).mean() + torch.log(1.0 - self.text_prior_d(text_features)).mean())
        if neg_text_features is None:
            cross = self._pair_terms(self.global_d, image_features, text_features, _roll1(text_features))
        else:
            # cluster mode (loss.py:225-252): hard negatives for the first half, rolled in-batch negatives for the second; note that the
            # reference re-binds text_features to the rolled batch here, so the textual SSL term below sees the rolled captions
            img_all = torch.cat((image_features, neg_image_features), dim=0)
            txt_all = torch.cat((text_features, neg_text_features), dim=0)
            text_features = _roll1(text_features)
            cross = self._pair_terms(self.global_d, img_all, txt_all, torch.cat((neg_text_features, text_features), dim=0))
        zero = image_features.new_zeros(())
        visual = textual = zero
        if aug_image_features is not None:
            visual = self._pair_terms(self.visual_d, image_features, aug_image_features, _roll1(aug_image_features))
        if aug_text_features is not None:
            textual = self._pair_terms(self.textual_d, text_features, aug_text_features, _roll1(aug_text_features))
        total = (1.0 - self.prior_weight) * (cross + visual + textual) + self.prior_weight * prior
        return {"total_loss": total, "cross_modal_loss": cross, "visual_loss": visual, "textual_loss": textual}


class OracleVLInfoModel(nn.Module):
    """model.py:15-113 (modes "sbert" and "train_sbert"; hard-negative and augmented-view branches :61-92)"""

    def __init__(self, text_encoder, image_encoder, loss, mode="sbert", is_amp=False):
        super().__init__()
        self.text_encoder, self.image_encoder, self.loss, self.mode = text_encoder, image_encoder, loss, mode

    def forward(self, batch):
        image_features = self.image_encoder(batch["image"])
        if self.mode == "sbert":
            text_features = self.text_encoder(batch["caption_encodings"])
        else:
            text_features = self.text_encoder({"input_ids": batch["input_ids"], "attention_mask": batch["attention_mask"]})
        extra = {}
        if self.mode != "sbert":
            enc_t = lambda i, m: self.text_encoder({"input_ids": batch[i], "attention_mask": batch[m]})
            if "neg_input_ids" in batch:
                extra["neg_image_features"] = self.image_encoder(batch["neg_image"])
                extra["neg_text_features"] = enc_t("neg_input_ids", "neg_attention_mask")
            if "aug_image" in batch:
                extra["aug_image_features"] = self.image_encoder(batch["aug_image"])
            if "aug_input_ids" in batch:
                extra["aug_text_features"] = enc_t("aug_input_ids", "aug_attention_mask")
        d = self.loss(image_features, text_features, **extra)
        return {"loss": d["total_loss"], "loss_components": {k: v.clone().detach() for k, v in d.items()},
                "image_features": image_features, "text_features": text_features}


def build_oracle_model(visual="resnet50", textual="train_sbert", num_hidden_layers=12, image_dim=None, text_dim=768,
                       image_prior=True, text_prior=True, prior_weight=0.1, dropout=0.1, critic="dot", visual_ssl=False, textual_ssl=False):
    ie = OracleImageEncoder(visual)
    te = OracleTextEncoder(mode=textual, num_hidden_layers=num_hidden_layers)
    if textual == "train_sbert" and dropout != 0.1:
        for m in te.modules():
            if isinstance(m, nn.Dropout):
                m.p = dropout
    loss = OracleJSDInfoMaxLoss(image_dim or ie.img_encoder.out_dim, text_dim, critic, prior_weight, image_prior, text_prior,
                                visual_self_supervised=visual_ssl, textual_self_supervised=textual_ssl)
    return OracleVLInfoModel(te, ie, loss, textual)


# ------------------------------------------------------------------------------------------------ update path
class Lookahead:
    """optim/lookahead.py:35-101"""

    def __init__(self, optimizer, k=5, alpha=0.8):
        self.optimizer, self.k, self.alpha, self._k_counter = optimizer, k, alpha, 0
        self.slow = {p: p.data.clone() for g in optimizer.param_groups for p in g["params"]}

    @property
    def param_groups(self):
        return self.optimizer.param_groups

    def zero_grad(self):
        self.optimizer.zero_grad()

    def step(self):
        self.optimizer.step()
        self._k_counter += 1
        if self._k_counter >= self.k:
            self._k_counter = 0
            for g in self.optimizer.param_groups:
                for p in g["params"]:
                    p.data.mul_(self.alpha).add_(self.slow[p], alpha=1.0 - self.alpha)
                    self.slow[p].copy_(p.data)


def lr_multiplier(name, step, total_steps, warmup_steps, min_mult=0.0, milestones=(), gamma=0.1):
    """optim/lr_scheduler.py: the four LinearWarmup*LR lambdas (none :60-70, multistep :100-112, linear :140-152, cosine :193-202)"""
    if step < warmup_steps:
        m = step / float(max(1, warmup_steps))
        return max(0, min_mult + m) if name == "cosine" else max(0, m)
    if name == "none":
        return 1.0
    if name == "multistep":
        return gamma ** sum(1 for s in milestones if s <= step)
    if name == "linear":
        return max(0, (total_steps - step) / (total_steps - warmup_steps))
    if name == "cosine":
        cf = (step - warmup_steps) / (total_steps - warmup_steps)
        return max(0, min_mult + math.cos(cf * (math.pi / 2)) ** 2)
    raise ValueError(name)


NO_DECAY_DEFAULT = ".*textual.(embedding|transformer).*(norm.*|bias)"


def build_optimizer(named_parameters, cnn_lr=0.2, trans_lr=1e-3, lr=1e-3, momentum=0.9, weight_decay=1e-4,
                    no_decay=NO_DECAY_DEFAULT, lookahead=True, k=5, alpha=0.5):
    """factories.py:464-487: one SGD param group per parameter; lr by name substring; wd off only where the regex matches."""
    groups = []
    for name, p in named_parameters:
        wd = 0.0 if re.match(no_decay, name) else weight_decay
        g_lr = cnn_lr if "image_encoder" in name else trans_lr if "text_encoder" in name else lr
        groups.append({"params": [p], "lr": g_lr, "weight_decay": wd})
    opt = torch.optim.SGD(groups, momentum=momentum)
    for g in opt.param_groups:
        g["initial_lr"] = g["lr"]
    return Lookahead(opt, k, alpha) if lookahead else opt


def train_step(model, optimizer, batch, step, sched=("cosine", 500000, 10000, 0.0), clip=10.0):
    """train.py:211-226 in fp32 (AMP off: autocast is a no-op on CPU): zero_grad, forward, backward, clip_grad_norm_, step, LR step.
    `step` is the scheduler's last_epoch *before* this call (LambdaLR starts at 0 => first step runs with multiplier(0))."""
    name, total, warm, min_mult = sched
    mult = lr_multiplier(name, step, total, warm, min_mult)
    for g in optimizer.param_groups:
        g["lr"] = g["initial_lr"] * mult
    optimizer.zero_grad()
    out = model(batch)
    out["loss"].backward()
    gn = torch.nn.utils.clip_grad_norm_([p for g in optimizer.param_groups for p in g["params"]], clip)
    optimizer.step()
    return out, gn


def train_step_shardwise(model, optimizer, shards, step, sched=("cosine", 500000, 10000, 0.0), clip=10.0, noises=None):
    """One DATA-PARALLEL step of the reference (train.py:174-178: DistributedDataParallel) emulated in one process, SURVEY.md §8(e): every rank
    runs the model on ITS shard — its own BatchNorm batch statistics, its own roll-by-one negatives (loss.py:214-216), its own prior noise —
    DDP averages the per-rank gradients, and every rank then clips and steps on that mean. This is NOT a step on the concatenated batch.
    `shards[r]` is rank r's batch, `noises[r]` its (image, text) prior noise. BatchNorm running statistics are rank-local under DDP and rank 0's
    are the ones that get checkpointed (utils/checkpointing.py:137-143): the buffers this function leaves in `model` are rank 0's.
    Returns ([per-rank output dicts], gradient norm before clipping)."""
    name, total, warm, min_mult = sched
    mult = lr_multiplier(name, step, total, warm, min_mult)
    for g in optimizer.param_groups:
        g["lr"] = g["initial_lr"] * mult
    params = [p for g in optimizer.param_groups for p in g["params"]]
    mean_grad = [torch.zeros_like(p) for p in params]
    outs = [None] * len(shards)
    for r in reversed(range(len(shards))):            # rank 0 last: its buffer updates are the ones that stay
        saved = {k: v.detach().clone() for k, v in model.named_buffers()}
        if noises is not None:
            model.loss.noise = noises[r]
        optimizer.zero_grad()
        outs[r] = model(shards[r])
        outs[r]["loss"].backward()
        for a, p in zip(mean_grad, params):
            if p.grad is not None:
                a += p.grad / len(shards)
        if r != 0:
            with torch.no_grad():
                for k, v in model.named_buffers():
                    v.copy_(saved[k])
    for a, p in zip(mean_grad, params):
        p.grad = a
    gn = torch.nn.utils.clip_grad_norm_(params, clip)
    optimizer.step()
    return outs, gn

