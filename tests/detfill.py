"""Deterministic weight fill shared by the golden generator, the oracle tests and the GPU parity tests.

Weights are never stored in fixtures (156 M parameters); instead every tensor of a state_dict is filled from a CPU
generator seeded by a hash of its key, so the reference modules (via load_state_dict), the oracle and the product get
bit-identical parameters from names alone."""
import hashlib
import math

import numpy as np
import torch


def _seed(key, salt):
    return int.from_bytes(hashlib.sha256(f"{salt}:{key}".encode()).digest()[:7], "little")


def det_tensor(key, shape, kind, salt="clite"):
    g = torch.Generator().manual_seed(_seed(key, salt))
    if kind == "normal":
        return torch.randn(shape, generator=g)
    return torch.rand(shape, generator=g)


def det_fill(module, salt="clite"):
    """Fill every float tensor of module.state_dict() deterministically (in place). Returns the module."""
    sd = module.state_dict()
    new = {}
    for k, v in sd.items():
        if not v.dtype.is_floating_point:
            new[k] = v.clone()
            continue
        shape = tuple(v.shape)
        leaf = k.split(".")[-1]
        is_norm = (v.dim() == 1 and leaf == "weight")           # BN / LayerNorm gain
        if k.endswith("temperature"):
            t = torch.ones(shape) * math.log(1 / 0.07)
        elif leaf == "running_var":
            t = 1.0 + 0.2 * det_tensor(k, shape, "uniform", salt)
        elif leaf == "running_mean":
            t = 0.1 * det_tensor(k, shape, "normal", salt)
        elif is_norm:
            t = 1.0 + 0.1 * det_tensor(k, shape, "normal", salt)
        elif leaf == "bias":
            t = 0.05 * det_tensor(k, shape, "normal", salt)
        elif k.endswith("feature_shortcut.weight"):
            t = (det_tensor(k, shape, "uniform", salt) * 0.02 - 0.01)
            n = min(shape)
            t[torch.arange(n), torch.arange(n)] = 1.0
        elif "embeddings" in k and v.dim() == 2:
            t = 0.05 * det_tensor(k, shape, "normal", salt)
        else:
            fan_in = int(np.prod(shape[1:])) if v.dim() > 1 else shape[0]
            t = det_tensor(k, shape, "normal", salt) * math.sqrt(2.0 / max(fan_in, 1))
        new[k] = t.to(v.dtype)
    module.load_state_dict(new)
    return module
