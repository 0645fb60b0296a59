import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.probe_bert import gemm
tag = sys.argv[1] if len(sys.argv) > 1 else ""
for k, M, N, K in [("nt", 8192, 8192, 8192), ("nn", 8192, 8192, 8192), ("tn", 8192, 8192, 8192), ("nt", 4096, 4096, 4096)]:
    us, tf = gemm(k, M, N, K)
    print(f"{tag:8s} gemm_{k} {M}x{N}x{K}: {us:9.1f} us {tf:8.1f} TF/s")
