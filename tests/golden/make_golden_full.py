"""Full-size oracle sample of BASELINE.json configs[1] (the benchmarked workload): ResNet-50 + BERT-base (12 layers) + JSD-MI heads and both
priors, batch 128, 224 x 224 images, 30-token captions — one fp32 forward + backward of oracle/ref_model.py on the CPU, run ONCE in the build
container (`python tests/golden/make_golden_full.py`, ~2 minutes on 8 cores, ~25 GB) because it is too slow for the GPU box's test budget.
Stores only scalars and per-module gradient norms (tests/golden/full_c2_b128.npz): weights come from tests/detfill.py, inputs from
det_tensor / a seeded generator, so the GPU test regenerates both. Dropout off, prior noise pinned.

The oracle is pinned to the reference by tests/golden/make_golden.py + tests/test_oracle_golden.py (loss.py, encoder.TextEncoder over HF BERT,
model.VLInfoModel); its ResNet restates torchvision 0.8.0's topology, which the reference does not vendor."""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)

from detfill import det_fill, det_tensor      # noqa: E402
from oracle import ref_model as O             # noqa: E402

B, S, L = 128, 224, 30


def inputs():
    ids = torch.randint(1000, 30522, (B, L), generator=torch.Generator().manual_seed(1234))
    ids[:, 0], ids[:, -1] = 101, 102
    batch = {"image": det_tensor("full_image", (B, 3, S, S), "normal"), "input_ids": ids, "attention_mask": torch.ones(B, L, dtype=torch.long)}
    noise = (det_tensor("full_u1", (B, 2048), "uniform"), det_tensor("full_u2", (B, 768), "uniform"))
    return batch, noise


def main():
    torch.set_num_threads(os.cpu_count() or 8)
    M = det_fill(O.build_oracle_model("resnet50", "train_sbert", 12, dropout=0.0)).train()
    batch, noise = inputs()
    M.loss.noise = noise
    t0 = time.time()
    out = M(batch)
    out["loss"].backward()
    print(f"oracle forward + backward: {time.time() - t0:.1f} s, loss {out['loss'].item():.6f}")
    comps = {k: float(v) for k, v in out["loss_components"].items()}
    norms = {}
    for n, p in M.named_parameters():
        top = n.split(".")[0]
        norms[top] = norms.get(top, 0.0) + float((p.grad.double() ** 2).sum())
    small = {n: p.grad.detach().numpy().copy() for n, p in M.named_parameters()
             if n in ("loss.global_d.temperature", "loss.prior_d.l2.weight", "loss.text_prior_d.l2.bias", "image_encoder.img_encoder.bn1.weight")}
    np.savez(os.path.join(HERE, "full_c2_b128.npz"), loss=np.float64(out["loss"].item()),
             **{"comp_" + k: np.float64(v) for k, v in comps.items()},
             **{"gradnorm_" + k: np.float64(v ** 0.5) for k, v in norms.items()},
             **{"grad_" + k: v for k, v in small.items()})
    print(comps, {k: v ** 0.5 for k, v in norms.items()})


if __name__ == "__main__":
    main()
