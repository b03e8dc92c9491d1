import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from cartnet_amd import ops
dev = torch.device("cuda:0")
M = 65536
for K in (256, 4096):
    A = torch.randn(M, K, device=dev); B = torch.randn(256, K, device=dev) * 0.05; Cc = torch.empty(M, 256, device=dev)
    for _ in range(3): ops.gemm(A, B, Cc)
    torch.cuda.synchronize()
