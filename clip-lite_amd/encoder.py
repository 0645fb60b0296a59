"""Image and text encoders with the reference's call surface (reference encoder.py:13-65 ImageEncoder, :115-205 TextEncoder),
running on the HIP executors in resnet.py / bert.py. Attribute names (`img_encoder`, `strans`, `fc1`/`fc2`) and state_dict keys
match the reference so checkpoints interchange."""
from typing import Any, Dict

import torch
import torch.nn as nn

from . import hip
from .bert import BertModel, LinearParams, bert_backward, bert_forward
from .resnet import ResNet, resnet_backward, resnet_forward


def _runtime_of(module):
    rt = getattr(module, "_clite_rt", None)
    if rt is None:
        raise RuntimeError("clip_lite_amd: module is not attached to a device runtime; build it through VLInfoModel (or "
                           "clip_lite_amd.model.attach_runtime) and move it to the GPU — there is no CPU path")
    return rt


class _ResNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, anchor, enc, rt, training):
        feat, saved = resnet_forward(rt, enc.img_encoder, image.to(torch.float32).contiguous(), training)
        if training:
            rt.bump_counters("image_encoder", 1)
        ctx.enc, ctx.rt, ctx.saved = enc, rt, saved
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        resnet_backward(ctx.rt, ctx.enc.img_encoder, ctx.saved, dfeat.contiguous())
        ctx.rt.arena.note_stream_work()
        return None, None, None, None, None


class ImageEncoder(nn.Module):
    r"""torchvision-topology ResNet with ``fc = Identity`` (reference encoder.py:28-65). ``pretrained`` weights cannot be
    downloaded here (no network) and raise; ``frozen`` keeps all weights fixed and the encoder in eval mode."""

    def __init__(self, img_enc_net: str = "resnet50", pretrained: bool = False, frozen: bool = False):
        super().__init__()
        if pretrained:
            raise RuntimeError("pretrained torchvision weights are not available offline; load a state_dict instead")
        self.img_encoder = ResNet(img_enc_net)
        self.frozen = frozen
        if frozen:
            for param in self.img_encoder.parameters():
                param.requires_grad = False
            self.img_encoder.eval()

    def forward(self, image: torch.Tensor) -> torch.Tensor:
        rt = _runtime_of(self)
        training = self.img_encoder.training and not self.frozen
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.img_encoder.parameters()):
            x = _ResNetFn.apply(image, rt.anchor, self, rt, training)
        else:
            x, _ = resnet_forward(rt, self.img_encoder, image.to(torch.float32).contiguous(), training)
            if training:
                rt.bump_counters("image_encoder", 1)
        return x.view(x.size(0), x.size(1))

    def detectron2_backbone_state_dict(self) -> Dict[str, Any]:
        """Same renaming as reference encoder.py:67-112 (torchvision names -> Detectron2 names)."""
        mapping = {"layer1": "res2", "layer2": "res3", "layer3": "res4", "layer4": "res5", "bn1": "conv1.norm", "bn2": "conv2.norm",
                   "bn3": "conv3.norm", "downsample.0": "shortcut", "downsample.1": "shortcut.norm"}
        out = {}
        for name, param in self.img_encoder.state_dict().items():
            for old, new in mapping.items():
                name = name.replace(old, new)
            if not name.startswith("res"):
                name = f"stem.{name}"
            out[name] = param
        return {"model": out, "__author__": "VLInfo", "matching_heuristics": True}


class _BertFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, enc, rt, input_ids, attention_mask, step):
        out, saved = bert_forward(rt, enc.strans, input_ids, attention_mask, step)
        ctx.enc, ctx.rt, ctx.saved = enc, rt, saved
        return out

    @staticmethod
    def backward(ctx, dout):
        bert_backward(ctx.rt, ctx.enc.strans, ctx.saved, dout.contiguous())
        rt = ctx.rt
        ev = rt.arena.note_stream_work()     # gradients were written on this node's stream: consumers of the arena join it
        # ... and so does everything the caller enqueues after backward() on the stream forward was called on (reading .grad,
        # a hand-written update): this node runs after the image encoder's (it was created first), so the join does not
        # serialise the two backward passes.
        main = getattr(rt, "main_stream", None)
        if ev is not None and main is not None and main != torch.cuda.current_stream(rt.device):
            main.wait_event(ev)
        return None, None, None, None, None, None


class _TransformFn(torch.autograd.Function):
    """optional fc1 -> ReLU -> fc2 head (reference encoder.py:182-185,200-203; TRANSFORM is false in every shipped YAML)"""

    @staticmethod
    def forward(ctx, x, enc, rt):
        B, D = x.shape
        E = enc.txt_enc_dim
        x = x.to(rt.tdtype).contiguous()
        h = torch.empty(B, E, device=rt.device, dtype=rt.tdtype)
        hip.gemm_nt(rt.dt, x, rt.arena.w(enc.fc1.weight), B, E, D, hip.epilogue(h, E, bias=enc.fc1.bias, act=hip.ACT_RELU))
        y = torch.empty(B, E, device=rt.device, dtype=rt.tdtype)
        hip.gemm_nt(rt.dt, h, rt.arena.w(enc.fc2.weight), B, E, E, hip.epilogue(y, E, bias=enc.fc2.bias))
        ctx.enc, ctx.rt, ctx.saved = enc, rt, (x, h)
        return y

    @staticmethod
    def backward(ctx, dy):
        from .bert import _linear_grads
        enc, rt = ctx.enc, ctx.rt
        x, h = ctx.saved
        B, D = x.shape
        E = enc.txt_enc_dim
        dy = dy.contiguous()
        _linear_grads(rt, enc.fc2, dy, h, B)
        dh = torch.empty(B, E, device=rt.device, dtype=rt.tdtype)
        hip.gemm_nn(rt.dt, dy, rt.arena.w(enc.fc2.weight), B, E, E, hip.epilogue(dh, E, dact_aux=h, dact=hip.DACT_RELU))
        _linear_grads(rt, enc.fc1, dh, x, B)
        dx = torch.empty(B, D, device=rt.device, dtype=rt.tdtype)
        hip.gemm_nn(rt.dt, dh, rt.arena.w(enc.fc1.weight), B, D, E, hip.epilogue(dx, D))
        rt.join_aux()
        return dx, None, None


class TextEncoder(nn.Module):
    r"""Reference encoder.py:122-205. Modes on the HIP path: ``"sbert"`` (frozen sentence embeddings passed through, 0 parameters)
    and ``"train_sbert"`` with a *bert* ``model_name`` (random-init BertModel, ``pooler_output``). GloVe / MPNet / pretrained
    downloads need assets or network that are unavailable and raise."""

    def __init__(self, word_dict=None, mode="train_sbert", transform_embedding=False, txt_enc_dim=512, glove_path=None, train_enc=False,
                 load_glove=True, model_name="bert-base-uncased", pretrained=False, num_hidden_layers=12):
        super().__init__()
        self.transform_embedding = transform_embedding
        self.txt_enc_dim = txt_enc_dim
        self.mode = mode
        self.model_name = model_name
        self.num_hidden_layers = num_hidden_layers
        if mode == "sbert":
            in_dim = 768
        elif mode == "train_sbert":
            if pretrained or "bert" not in model_name:
                raise RuntimeError("only a randomly initialised BERT text encoder can be built offline (no pretrained/MPNet download)")
            print("Using bert model with layers: " + str(self.num_hidden_layers))
            self.strans = BertModel(num_hidden_layers=self.num_hidden_layers)
            in_dim = 768
        else:
            raise NotImplementedError(f"text encoder mode {mode!r} needs assets that are not available (GloVe vectors / HF hub)")
        if transform_embedding:
            self.fc1 = LinearParams(in_dim, self.txt_enc_dim)
            self.fc2 = LinearParams(self.txt_enc_dim, self.txt_enc_dim)

    def forward(self, x):
        rt = _runtime_of(self)
        if self.mode == "train_sbert":
            step = rt.next_step(self.training)
            grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.strans.parameters())
            if grad:
                x = _BertFn.apply(rt.anchor, self, rt, x["input_ids"], x["attention_mask"], step)
            else:
                x, _ = bert_forward(rt, self.strans, x["input_ids"], x["attention_mask"], step)
        if self.transform_embedding:
            x = _TransformFn.apply(x, self, rt)
        return x

    def train_enc(self):
        for param in self.strans.parameters():
            param.requires_grad = True

    def dont_train_enc(self):
        for param in self.strans.parameters():
            param.requires_grad = False
