"""Diagnostic: per-parameter gradient error of the HIP f32 path and of the fp32 oracle, both against an fp64 oracle."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from test_gpu_model import run_case
from detfill import det_fill, det_tensor
from oracle import ref_model as O

visual, mode, layers, B, S, Ls, idim = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7])
lowp = len(sys.argv) > 8 and sys.argv[8] == "bf16"
M, Mo, _Md, out, ref = run_case(visual, mode, layers, lowp, B, S, Ls, idim)
# fp64 oracle
Md = det_fill(O.build_oracle_model(visual, mode, max(layers, 1), dropout=0.0)).double().train()
batch = {"image": det_tensor("image", (B, 3, S, S), "normal").double()}
if mode == "sbert":
    batch["caption_encodings"] = det_tensor("cap", (B, 768), "normal").double()
else:
    ids = torch.randint(1000, 30522, (B, Ls), generator=torch.Generator().manual_seed(1)); ids[:, 0] = 101; ids[:, -1] = 102
    mask = torch.ones(B, Ls, dtype=torch.long); mask[B - 1, Ls - 2:] = 0
    batch["input_ids"], batch["attention_mask"] = ids, mask
Md.loss.noise = (det_tensor("u1", (B, idim), "uniform").double(), det_tensor("u2", (B, 768), "uniform").double())
od = Md(batch); od["loss"].backward()
print(f"loss hip {out['loss'].item():.7f} oracle32 {ref['loss'].item():.7f} oracle64 {od['loss'].item():.7f}")
gd = {k: p.grad for k, p in Md.named_parameters()}
g32 = {k: p.grad for k, p in Mo.named_parameters()}
for k, p in M.named_parameters():
    r = gd[k]; sc = r.abs().max().item()
    e_hip = (p.grad.detach().double().cpu() - r).abs().max().item() / max(sc, 1e-12)
    e_o32 = (g32[k].double() - r).abs().max().item() / max(sc, 1e-12)
    flag = " <<<" if e_hip > 10 * max(e_o32, 1e-5) else ""
    print(f"{k:75s} scale {sc:9.3e}  hip {e_hip:9.2e}  oracle32 {e_o32:9.2e}{flag}")

# summary: cosine similarity of gradients against fp64 truth
import math
num = den1 = den2 = 0.0
worst = []
for k, p in M.named_parameters():
    a = p.grad.detach().double().cpu().flatten(); b = gd[k].flatten()
    c = float(a @ b) / (float(a.norm()) * float(b.norm()) + 1e-30)
    worst.append((c, k, float(b.norm())))
    num += float(a @ b); den1 += float(a @ a); den2 += float(b @ b)
print("global grad cosine", num / math.sqrt(den1 * den2))
for c, k, n in sorted(worst)[:12]:
    print(f"  cos {c:.4f}  |g| {n:.3e}  {k}")
