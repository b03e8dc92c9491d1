"""Precision-2 (plain bf16 operand) activation x weight products against fp64, every pipeline length: debugging aid for
the K-loop of cn_gemm_x3nn_kernel<*, true> / cn_gemm_hnn_kernel (CARTNET_LIB selects the build)."""
import sys, torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from cartnet_amd import ops

dev = "cuda"
def rnd(*s, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*s, generator=g) * scale).to(dev)

for M in (100, 300):
    for K in (16, 32, 48, 64, 80, 96, 112, 128, 144, 256, 512):
        for act in (False, True):
            for stride_pad in (0, 16):
                X = rnd(M, K + stride_pad, seed=K)
                Xs = [X[:, :K]]
                W = rnd(256, K, seed=1, scale=0.1)
                img = ops.split_b([W.t()])
                C = [torch.full((M, 256), float("nan"), device=dev)]
                ops.gemm(Xs, [W.t().contiguous()], C, b_kstrided=True, a_act=act, b_split=img, precision=2)
                x = Xs[0].double()
                if act:
                    x = x * torch.sigmoid(x)
                ref = x.bfloat16().double() @ W.bfloat16().double().t()
                err = ((C[0].double() - ref).norm() / ref.norm()).item()
                print(f"M {M} K {K} act {act} pad {stride_pad}: {err:.2e} {'BAD' if not err < 1e-3 else ''}", flush=True)
