"""bf16 storage emulation for the oracle (test infrastructure): round to bf16 every tensor the HIP production path stores in bf16 —
conv / linear outputs, ReLU outputs, LayerNorm inputs and outputs, pooled features, the bf16 weight copies — and round the
gradients flowing through the same points. Used to tell bf16 rounding noise (inherent: a randomly initialised ResNet with
train-mode BatchNorm amplifies 2^-9 relative perturbations enormously) from implementation error."""
import torch
import torch.nn as nn


class Round(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float()


def emulate_bf16_storage(model):
    for m in model.modules():
        if isinstance(m, nn.ReLU):
            m.inplace = False
        if isinstance(m, (nn.Conv2d, nn.Linear)):
            m.weight.data = m.weight.data.bfloat16().float()
            m.register_forward_hook(lambda mod, inp, out: Round.apply(out))
        elif isinstance(m, (nn.ReLU, nn.AdaptiveAvgPool2d)):
            m.register_forward_hook(lambda mod, inp, out: Round.apply(out))
        elif isinstance(m, nn.LayerNorm):
            m.register_forward_pre_hook(lambda mod, inp: (Round.apply(inp[0]),))
            m.register_forward_hook(lambda mod, inp, out: Round.apply(out))
        elif isinstance(m, nn.Embedding):
            m.weight.data = m.weight.data.bfloat16().float()
    return model


def round_batch(batch):
    return {k: (v.bfloat16().float() if v.dtype.is_floating_point else v) for k, v in batch.items()}
