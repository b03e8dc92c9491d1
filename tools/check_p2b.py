"""Debug: plain precision-2 launch vs the a_act_out launch vs fp64, big M, two groups (where do they differ?)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from cartnet_amd import ops
dev = "cuda"
def rnd(*s, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*s, generator=g) * scale).to(dev)
M, K, N, groups = 33000, 256, 256, 2
X = rnd(M, groups * K + 16, seed=11)
Xs = [X[:, g * K:(g + 1) * K] for g in range(groups)]
Ws = [rnd(N, K, seed=20 + g, scale=0.1) for g in range(groups)]
bs = [rnd(N, seed=30 + g) for g in range(groups)]
Bt = [w.t().contiguous() for w in Ws]
imgs = ops.split_b([w.t() for w in Ws])
for rep in range(3):
    C0 = [torch.full((M, N), float("nan"), device=dev) for _ in range(groups)]
    Cs = [torch.full((M, N), float("nan"), device=dev) for _ in range(groups)]
    H = torch.empty_like(X); Hs = [H[:, g * K:(g + 1) * K] for g in range(groups)]
    ops.gemm(Xs, Bt, C0, b_kstrided=True, a_act=True, bias=bs, b_split=imgs, precision=2)
    ops.gemm(Xs, Bt, Cs, b_kstrided=True, a_act=True, bias=bs, b_split=imgs, a_act_out=Hs, precision=2)
    C1 = [torch.full((M, N), float("nan"), device=dev) for _ in range(groups)]
    ops.gemm(Xs, Bt, C1, b_kstrided=True, a_act=True, bias=bs, b_split=imgs, precision=2)
    for g in range(groups):
        x = Xs[g].double(); a = (x * torch.sigmoid(x)).float().bfloat16().double()
        ref = a @ Ws[g].bfloat16().double().t() + bs[g].double()
        e0 = (C0[g].double() - ref).abs().max().item(); es = (Cs[g].double() - ref).abs().max().item()
        ne = (C0[g] != Cs[g]); n01 = (C0[g] != C1[g]).sum().item()
        rows = ne.any(1).nonzero().flatten()
        print(f"rep {rep} g {g}: max err plain {e0:.3e} act_out {es:.3e}; plain != act_out in {ne.sum().item()} elements, {rows.numel()} rows"
              f" (first rows {rows[:8].tolist()}, rows%128 {sorted(set((rows % 128).tolist()))[:12]}); plain run-to-run differs in {n01}", flush=True)
        if ne.any():
            r = rows[0].item(); cols = ne[r].nonzero().flatten()
            print("   row", r, "cols", cols[:8].tolist(), "plain", C0[g][r, cols[:4]].tolist(), "act_out", Cs[g][r, cols[:4]].tolist(), "ref", ref[r, cols[:4]].tolist())
