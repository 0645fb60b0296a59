"""Diagnostic: is the bf16 HIP path's deviation from the fp32 oracle rounding noise or a bug?  The oracle is re-run with bf16
rounding emulated at the tensors the HIP path stores in bf16 (conv/linear outputs, ReLU outputs, LayerNorm inputs/outputs,
pooled features, bf16 weight copies; gradients rounded at the same points) and the HIP bf16 gradients are compared with it."""
import sys, os, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.nn as nn
from test_gpu_model import run_case
from detfill import det_fill, det_tensor
from oracle import ref_model as O


class Round(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float()


def emulate(model):
    for m in model.modules():
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


visual, mode, layers, B, S, Ls, idim = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7])
init = sys.argv[8] if len(sys.argv) > 8 else "det"
M, Mo, Md, out, ref = run_case(visual, mode, layers, True, B, S, Ls, idim, init)
Me = O.build_oracle_model(visual, mode, max(layers, 1), dropout=0.0).train()
Me.load_state_dict(Mo.state_dict())
for m in Me.modules():
    if isinstance(m, nn.ReLU):
        m.inplace = False
emulate(Me)
batch = {"image": det_tensor("image", (B, 3, S, S), "normal").bfloat16().float()}
if mode == "sbert":
    batch["caption_encodings"] = det_tensor("cap", (B, 768), "normal").bfloat16().float()
else:
    ids = torch.randint(1000, 30522, (B, Ls), generator=torch.Generator().manual_seed(1)); ids[:, 0] = 101; ids[:, -1] = 102
    mask = torch.ones(B, Ls, dtype=torch.long); mask[B - 1, Ls - 2:] = 0
    batch["input_ids"], batch["attention_mask"] = ids, mask
Me.loss.noise = (det_tensor("u1", (B, idim), "uniform").bfloat16().float(), det_tensor("u2", (B, 768), "uniform").bfloat16().float())
oe = Me(batch); oe["loss"].backward()
print(f"loss hip-bf16 {out['loss'].item():.6f}  emulated-bf16 oracle {oe['loss'].item():.6f}  fp32 oracle {ref['loss'].item():.6f}")
ge = {k: p.grad for k, p in Me.named_parameters()}
g32 = {k: p.grad for k, p in Mo.named_parameters()}
def cos(a, b): return float(a.flatten() @ b.flatten()) / (float(a.norm()) * float(b.norm()) + 1e-30)
rows = []
n1 = d1 = d2 = n2 = e1 = e2 = 0.0
for k, p in M.named_parameters():
    a = p.grad.detach().float().cpu(); b = ge[k]; c = g32[k]
    rows.append((cos(a, b), cos(b, c), float(c.norm()), k))
    n1 += float(a.flatten() @ b.flatten()); d1 += float(a.norm() ** 2); d2 += float(b.norm() ** 2)
    n2 += float(b.flatten() @ c.flatten()); e2 += float(c.norm() ** 2)
print("global cosine  hip vs emulated:", n1 / math.sqrt(d1 * d2), "  emulated vs fp32:", n2 / math.sqrt(d2 * e2))
for r in sorted(rows)[:25]:
    print(f"  hip~emul {r[0]:.4f}   emul~fp32 {r[1]:.4f}   |g| {r[2]:.3e}  {r[3]}")
