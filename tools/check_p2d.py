"""Debug: shape of the wrong blocks of the plain precision-2 launch at M = 177140 (reference: the a_act_out kernel)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from cartnet_amd import ops
dev = "cuda"
def rnd(*s, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*s, generator=g) * scale).to(dev)
M, K, N, groups = 177140, 256, 256, 2
X = rnd(M, groups * K + 16, seed=11)
Xs = [X[:, g * K:(g + 1) * K] for g in range(groups)]
Ws = [rnd(N, K, seed=20 + g, scale=0.1) for g in range(groups)]
Bt = [w.t().contiguous() for w in Ws]
H = torch.empty_like(X); Hs = [H[:, g * K:(g + 1) * K] for g in range(groups)]
imgs = ops.split_b([w.t() for w in Ws])
R = [torch.empty(M, N, device=dev) for _ in range(groups)]
ops.gemm(Xs, Bt, R, b_kstrided=True, a_act=True, b_split=imgs, precision=2, a_act_out=Hs)
for rep in range(3):
    C = [torch.full((M, N), float("nan"), device=dev) for _ in range(groups)]
    ops.gemm(Xs, Bt, C, b_kstrided=True, a_act=True, b_split=imgs, precision=2)
    for g in range(groups):
        ne = C[g] != R[g]
        rows = ne.any(1).nonzero().flatten()
        print(f"rep {rep} g {g}: {int(ne.sum())} wrong elements in {rows.numel()} rows")
        tiles = sorted(set((rows // 128).tolist()))
        for t in tiles[:6]:
            blk = ne[t * 128:(t + 1) * 128]
            rr = blk.any(1).nonzero().flatten(); cc = blk.any(0).nonzero().flatten()
            d = (C[g][t * 128:(t + 1) * 128] - R[g][t * 128:(t + 1) * 128]).abs()
            print(f"   tile {t}: local rows {rr.min().item()}..{rr.max().item()} ({rr.numel()}), cols {cc.min().item()}..{cc.max().item()} ({cc.numel()}), "
                  f"max |diff| {d.max().item():.3e} (|value| ~ {R[g][t*128:(t+1)*128].abs().mean().item():.2f})")
