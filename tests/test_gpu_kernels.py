"""Per-kernel parity on the GPU, through the C ABI (cartnet_amd.ops -> libcartnet_hip.so).

Each hand-written kernel is compared with an fp64 torch evaluation of the same arithmetic (the oracle functions in
oracle/cartnet_ref.py where one exists).  Tolerance: max|delta| <= 1e-5 * max|ref| (north_star's fp32 bar),
integers bit-exact.
"""
import math

import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-5


@pytest.fixture(scope="module")
def ops():
    from cartnet_amd import ops as _ops
    from cartnet_amd import lib
    lib.load()
    return _ops


def dev():
    return torch.device("cuda:0")


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dev())


def silu64(x):
    return x * torch.sigmoid(x)


def dsilu64(x):
    s = torch.sigmoid(x)
    return s * (1 + x * (1 - s))


# --------------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K", [(300, 256, 256), (129, 64, 67), (1000, 128, 512), (5, 200, 40), (128, 256, 16),
                                   (257, 512, 100)])
def test_gemm_nt_plain(ops, M, N, K):
    A, B = rnd(M, K, seed=1), rnd(N, K, seed=2)
    C = torch.empty(M, N, device=dev())
    ops.gemm(A, B, C)
    ref = A.double() @ B.double().t()
    assert rel_err(C, ref) < TOL


def test_gemm_layout_asymmetric(ops):
    # A = I (padded) against an asymmetric B catches swapped row/col maps in the MFMA epilogue.
    M = N = K = 256
    A = torch.eye(M, device=dev())
    B = (torch.arange(N * K, device=dev(), dtype=torch.float32).reshape(N, K) % 1021) / 7.0
    C = torch.empty(M, N, device=dev())
    ops.gemm(A, B, C)
    assert torch.equal(C, B.t().contiguous())


def test_gemm_nt_strided_views_and_epilogue(ops):
    M, D = 700, 64
    W = rnd(D, 3 * D, seed=3)              # reference-shaped [D, 3D] weight, use the last column block in place
    A = rnd(M, D, seed=4)
    bias = rnd(D, seed=5)
    Nn = 90
    Pi, Pj = rnd(Nn, 2 * D, seed=6), rnd(Nn, 2 * D, seed=7)
    g = torch.Generator().manual_seed(8)
    tgt = torch.sort(torch.randint(0, Nn, (M,), generator=g)).values.to(torch.int32).to(dev())
    src = torch.randint(0, Nn, (M,), generator=g).to(torch.int32).to(dev())
    out = torch.empty(M, 2 * D, device=dev())
    tiles = ops.gemm_tiles_m(M)
    cs = torch.zeros(tiles * D, device=dev(), dtype=torch.float64)
    cq = torch.zeros(tiles * D, device=dev(), dtype=torch.float64)
    ops.gemm(A, W[:, 2 * D:], out[:, D:], bias=bias, gather_i=Pi[:, D:], gather_j=Pj[:, D:], tgt=tgt, src=src,
             colsum=cs, colsq=cq)
    ref = A.double() @ W[:, 2 * D:].double().t() + bias.double() + Pi[:, D:].double()[tgt.long()] + \
        Pj[:, D:].double()[src.long()]
    assert rel_err(out[:, D:], ref) < TOL
    assert rel_err(cs.view(tiles, D).sum(0), ref.sum(0)) < 1e-5
    assert rel_err(cq.view(tiles, D).sum(0), (ref ** 2).sum(0)) < 1e-5


def test_gemm_act_prologue_epilogue(ops):
    M, N, K = 333, 128, 256
    A, B, bias = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.1), rnd(N, seed=3)
    C = torch.empty(M, N, device=dev())
    Cpre = torch.empty(M, N, device=dev())
    ops.gemm(A, B, C, a_act=True, out_act=True, bias=bias, cpre=Cpre)
    pre = silu64(A.double()) @ B.double().t() + bias.double()
    assert rel_err(Cpre, pre) < TOL
    assert rel_err(C, silu64(pre)) < TOL


def test_gemm_nn_segments_resid_dact(ops):
    # dX = resid + sum_s dY_s @ W_s, then * silu'(pre) -- the shape of the layer's backward data GEMMs
    M, D = 450, 64
    dY = rnd(M, 2 * D, seed=1)
    W1, W2 = rnd(D, 3 * D, seed=2, scale=0.2), rnd(D, 3 * D, seed=3, scale=0.2)
    resid, pre = rnd(M, D, seed=4), rnd(M, D, seed=5)
    out = torch.empty(M, D, device=dev())
    ops.gemm([dY[:, :D], dY[:, D:]], [W1[:, 2 * D:], W2[:, 2 * D:]], out, b_kstrided=True, segments=True,
             resid=resid, dact=pre)
    ref = (resid.double() + dY[:, :D].double() @ W1[:, 2 * D:].double() + dY[:, D:].double() @ W2[:, 2 * D:].double()) \
        * dsilu64(pre.double())
    assert rel_err(out, ref) < TOL
    # in place: output aliases dact
    pre2 = pre.clone()
    ops.gemm([dY[:, :D], dY[:, D:]], [W1[:, 2 * D:], W2[:, 2 * D:]], pre2, b_kstrided=True, segments=True,
             resid=resid, dact=pre2)
    assert rel_err(pre2, ref) < TOL


@pytest.mark.parametrize("E,Dout,Din,splitk", [(5000, 64, 64, 7), (3000, 256, 256, 4), (777, 128, 67, 3),
                                              (100, 64, 256, 1)])
def test_gemm_tn_splitk_groups(ops, E, Dout, Din, splitk):
    # dW[g] = dY[g]^T @ silu(X[g]) with the reduction over E rows split across workgroups
    dY = [rnd(E, Dout, seed=1), rnd(E, Dout, seed=2)]
    X = [rnd(E, Din, seed=3), rnd(E, Din, seed=4)]
    if splitk > 1:
        slabs = [torch.empty(splitk * Dout, Din, device=dev()) for _ in range(2)]
        ops.gemm(dY, X, slabs, a_kstrided=True, b_kstrided=True, b_act=True, splitk=splitk)
        outs = [torch.empty(Dout, 3 * Din, device=dev()) for _ in range(2)]
        ops.splitk_reduce(slabs, splitk, [o[:, Din:2 * Din] for o in outs])
        got = [o[:, Din:2 * Din] for o in outs]
    else:
        got = [torch.empty(Dout, Din, device=dev()) for _ in range(2)]
        ops.gemm(dY, X, got, a_kstrided=True, b_kstrided=True, b_act=True)
    for gI in range(2):
        ref = dY[gI].double().t() @ silu64(X[gI].double())
        assert rel_err(got[gI], ref) < TOL


def test_gemm_four_groups(ops):
    N, D = 200, 64
    x = rnd(N, D, seed=1)
    Wg, Wa = rnd(D, 3 * D, seed=2), rnd(D, 3 * D, seed=3)
    bg, ba = rnd(D, seed=4), rnd(D, seed=5)
    P = torch.empty(N, 4 * D, device=dev())
    ops.gemm([x, x, x, x], [Wg[:, :D], Wa[:, :D], Wg[:, D:2 * D], Wa[:, D:2 * D]],
             [P[:, 0:D], P[:, D:2 * D], P[:, 2 * D:3 * D], P[:, 3 * D:]], bias=[bg, ba, None, None])
    xd = x.double()
    ref = torch.cat([xd @ Wg[:, :D].double().t() + bg.double(), xd @ Wa[:, :D].double().t() + ba.double(),
                     xd @ Wg[:, D:2 * D].double().t(), xd @ Wa[:, D:2 * D].double().t()], dim=1)
    assert rel_err(P, ref) < TOL


def test_gemm_rejects_bad_shapes(ops):
    A, B = rnd(10, 16), rnd(12, 20)
    with pytest.raises(ValueError):
        ops.gemm(A, B, torch.empty(10, 12, device=dev()))
    with pytest.raises(ValueError):
        ops.gemm(A, rnd(12, 16), torch.empty(10, 13, device=dev()))
    with pytest.raises(ValueError):
        ops.gemm(A.cpu(), rnd(12, 16), torch.empty(10, 12, device=dev()))


# ---- second-generation kernels: pre-arranged weight images (gemm_f32.h / gemm_x3.h), every pipeline length
@pytest.mark.parametrize("precision", [0, 1])
@pytest.mark.parametrize("M,K,N,groups,act", [(1, 16, 256, 1, False), (127, 32, 256, 2, True), (128, 48, 512, 1, False),
                                              (300, 64, 256, 4, False), (1000, 80, 256, 1, True),
                                              (129, 256, 512, 2, False), (513, 512, 256, 1, True),
                                              (12416, 256, 256, 1, True), (9000, 64, 512, 1, False),
                                              (8200, 48, 256, 2, False)])
def test_gemm_with_weight_images(ops, precision, M, K, N, groups, act):
    """Y[g] = (silu?)(X[g]) W[g]^T + b[g] through the DMA-fed kernels: K from one K-step (pipeline head only) to 32
    (steady-state loop), ragged and single-row M, one and two column tiles, grouped launches."""
    Xs = [rnd(M, K, seed=10 + g) for g in range(groups)]
    Ws = [rnd(N, K, seed=20 + g, scale=0.1) for g in range(groups)]
    bs = [rnd(N, seed=30 + g) for g in range(groups)]
    Bt = [w.t().contiguous() for w in Ws]
    imgs = (ops.pack_b if precision == 0 else ops.split_b)([w.t() for w in Ws])
    Cs = [torch.full((M, N), float("nan"), device=dev()) for _ in range(groups)]
    ops.gemm(Xs, Bt, Cs, b_kstrided=True, a_act=act, bias=bs, precision=precision, b_split=imgs)
    for g in range(groups):
        x = Xs[g].double()
        ref = (silu64(x) if act else x) @ Ws[g].double().t() + bs[g].double()
        assert rel_err(Cs[g], ref) < TOL, g


@pytest.mark.parametrize("act", [False, True])
@pytest.mark.parametrize("M,pad", [(1, 0), (100, 16), (300, 0), (4000, 16)])
def test_gemm_plain_bf16_operands_every_pipeline_length(ops, M, pad, act):
    """Precision 2 through the DMA-fed kernel against fp64 on the bf16-rounded operands (what the MFMA multiplies), for
    every K-loop length from the pipeline head alone (1-3 K-steps: the head re-reads its last tile) over every phase
    of the four-slot ring to the steady state: 1e-6, i.e. only the fp32 accumulation differs."""
    for K in (16, 32, 48, 64, 80, 96, 112, 128, 144, 256, 528):
        X = rnd(M, K + pad, seed=K)
        x = X[:, :K]
        W = rnd(256, K, seed=1, scale=0.1)
        C = torch.full((M, 256), float("nan"), device=dev())
        ops.gemm([x], [W.t().contiguous()], [C], b_kstrided=True, a_act=act, b_split=ops.split_b([W.t()]), precision=2)
        a = silu64(x.double()).float() if act else x
        ref = a.bfloat16().double() @ W.bfloat16().double().t()
        assert rel_err(C, ref) < (1e-3 if act else 1e-6), K     # the kernel's own SiLU may round a few operands the other way


@pytest.mark.parametrize("a_dtype", [torch.float32, torch.bfloat16])
def test_plain_bf16_products_are_bitwise_repeatable_when_the_chip_is_oversubscribed(ops, a_dtype):
    """2,768 workgroups (two per CU, five rounds of them): waves of a workgroup drift apart by as much as the barriers
    allow, which is what exposes a missing one -- a first version of the precision-2 K-loop read its first fragments
    after the head's barrier and overwrote their buffer in step 0 without a second one, and was wrong in ~1 tile per
    launch at this size while every small case passed.  Fresh weight images per repetition (first touch from HBM)."""
    M, K, N, groups = 177140, 256, 256, 2
    X = rnd(M, groups * K + 16, seed=11).to(a_dtype)
    Xs = [X[:, g * K:(g + 1) * K] for g in range(groups)]
    Ws = [rnd(N, K, seed=20 + g, scale=0.1) for g in range(groups)]
    Bt = [w.t().contiguous() for w in Ws]
    first = None
    for rep in range(12):
        imgs = ops.split_b([w.t() for w in Ws])
        C = [torch.full((M, N), float("nan"), device=dev()) for _ in range(groups)]
        ops.gemm(Xs, Bt, C, b_kstrided=True, a_act=True, b_split=imgs, precision=2)
        if first is None:
            first = C
            x = Xs[0][:4096].double()
            ref = silu64(x).float().bfloat16().double() @ Ws[0].bfloat16().double().t()
            assert rel_err(C[0][:4096], ref) < 2e-3
        else:
            assert all(torch.equal(a, b) for a, b in zip(C, first)), rep


@pytest.mark.parametrize("precision", [0, 1, 2])
@pytest.mark.parametrize("M,K,N,groups", [(1, 16, 256, 1), (127, 32, 256, 2), (300, 48, 256, 2), (1000, 80, 512, 1),
                                          (33000, 256, 256, 2), (513, 512, 256, 1)])
def test_gemm_writes_the_activated_operand(ops, precision, M, K, N, groups):
    """a_act_out: Y = silu(X) W^T + b as before (bitwise the kernel without the by-product) and silu(X) written with X's
    row stride -- the groups are column blocks of one [M, groups*K] matrix as in the model (pre = [gate | aggr]); cells
    outside the written blocks stay untouched; every pipeline length; ragged / single-row M; two column tiles."""
    X = rnd(M, groups * K + 16, seed=11)
    Xs = [X[:, g * K:(g + 1) * K] for g in range(groups)]
    Ws = [rnd(N, K, seed=20 + g, scale=0.1) for g in range(groups)]
    bs = [rnd(N, seed=30 + g) for g in range(groups)]
    Bt = [w.t().contiguous() for w in Ws]
    imgs = (ops.pack_b if precision == 0 else ops.split_b)([w.t() for w in Ws])
    Cs = [torch.full((M, N), float("nan"), device=dev()) for _ in range(groups)]
    C0 = [torch.full((M, N), float("nan"), device=dev()) for _ in range(groups)]
    H = torch.full_like(X, -7.0)
    Hs = [H[:, g * K:(g + 1) * K] for g in range(groups)]
    tiles = ops.gemm_tiles_m(M)
    cs = [torch.zeros(tiles * N, dtype=torch.float64, device=dev()) for _ in range(groups)]
    ops.gemm(Xs, Bt, C0, b_kstrided=True, a_act=True, bias=bs, b_split=imgs, precision=precision)
    ops.gemm(Xs, Bt, Cs, b_kstrided=True, a_act=True, bias=bs, b_split=imgs, a_act_out=Hs, precision=precision,
             colsum=[cs[0]] + [None] * (groups - 1))
    for g in range(groups):
        if precision == 1:
            # bf16x3: the plain launch runs on the 16x16x32 MFMA shape (csrc/gemm_x3s.h), the one with the by-product on
            # the 32x32x16 kernel -- the same six products per element in another summation order
            assert rel_err(Cs[g], C0[g].double()) < 2e-6, g
        else:
            assert torch.equal(Cs[g], C0[g]), g
        ref = silu64(Xs[g].double())
        assert rel_err(Hs[g], ref) < 1e-6, g
    assert bool((H[:, groups * K:] == -7.0).all())
    assert rel_err(cs[0].view(tiles, N).sum(0), Cs[0].double().sum(0)) < 1e-6
    # the weight gradient from the kept operand == the one that recomputes the SiLU
    dY = [rnd(M, N, seed=90 + g) for g in range(groups)]
    W1 = [torch.empty(N, K, device=dev()) for _ in range(groups)]
    W2 = [torch.empty(N, K, device=dev()) for _ in range(groups)]
    ops.gemm(dY, Xs, W1, a_kstrided=True, b_kstrided=True, b_act=True, precision=precision)
    ops.gemm(dY, Hs, W2, a_kstrided=True, b_kstrided=True, precision=precision)
    tol = (TOL, 1e-5, 2e-2)[precision]
    for g in range(groups):
        ref = dY[g].double().t() @ silu64(Xs[g].double())
        assert rel_err(W1[g], ref) < tol and rel_err(W2[g], ref) < tol


@pytest.mark.parametrize("precision", [0, 1])
def test_gemm_activated_operand_without_the_fused_kernel(ops, precision):
    """No weight image / few row tiles / bf16x3: silu(X) comes from the elementwise pass; a_act=False is refused."""
    X, W = rnd(70, 40, seed=1)[:, :32], rnd(256, 32, seed=2)
    C_ = torch.empty(70, 256, device=dev())
    H = torch.full((70, 40), -7.0, device=dev())
    ops.gemm(X, W.t().contiguous(), C_, b_kstrided=True, a_act=True, a_act_out=H[:, :32], precision=precision)
    assert rel_err(H[:, :32], silu64(X.double())) < 1e-6 and bool((H[:, 32:] == -7.0).all())
    assert rel_err(C_, silu64(X.double()) @ W.double().t()) < (TOL if precision == 0 else 1e-5)
    with pytest.raises(RuntimeError, match="a_act_out"):
        ops.gemm(X, W.t().contiguous(), C_, b_kstrided=True, a_act=False, a_act_out=H[:, :32])


@pytest.mark.parametrize("precision", [0, 1])
def test_gemm_atom_sized_rows_folded_segments(ops, precision):
    """The dX product of the node terms at the benchmark shape: M = 12,416 atoms (97 row tiles: at precision 0 the
    128-wide DMA-fed kernel), four K-segments that are column blocks of one [M, 1024] matrix, residual, silu' and bias
    gradient sums (layer 0's form)."""
    M, K, N = 12416 + 5, 256, 256
    X = rnd(M, 4 * K, seed=3)
    Ws = [rnd(K, N, seed=20 + i, scale=0.1) for i in range(4)]
    resid, pre = rnd(M, N, seed=5), rnd(M, N, seed=6)
    tiles = ops.gemm_tiles_m(M)
    cs = torch.full((tiles * N,), float("nan"), dtype=torch.float64, device=dev())
    C_ = torch.full((M, N), float("nan"), device=dev())
    img = torch.cat((ops.pack_b if precision == 0 else ops.split_b)(Ws))
    ops.gemm([X[:, i * K:(i + 1) * K] for i in range(4)], Ws, C_, b_kstrided=True, segments=True, resid=resid, dact=pre,
             colsum=cs, b_split_folded=img, precision=precision)
    ref = (sum(X[:, i * K:(i + 1) * K].double() @ Ws[i].double() for i in range(4)) + resid.double()) * dsilu64(pre.double())
    assert rel_err(C_, ref) < TOL
    assert rel_err(cs.view(tiles, N).sum(0), C_.double().sum(0)) < 1e-6


@pytest.mark.parametrize("case", ["plain", "gather", "dact_colsum", "resid_folded"])
def test_gemm_edge_sized_rows_through_every_model_epilogue(ops, case):
    """M > 32768 rows at precision 0 with a weight image (hundreds of row tiles, ragged last tile), through every
    epilogue the model gives the DMA-fed kernel: bias, node-term gather, silu' with bias-gradient sums, residual with
    folded K-segments."""
    M = 32768 + (300 if case != "plain" else 77)
    K, N, G = 256, 256, 2
    Xs = [rnd(M, K, seed=40 + g) for g in range(G)]
    Ws = [rnd(N, K, seed=50 + g, scale=0.1) for g in range(G)]
    imgs = ops.pack_b([w.t() for w in Ws])
    Bt = [w.t().contiguous() for w in Ws]
    Cs = [torch.full((M, N), float("nan"), device=dev()) for _ in range(G)]
    ref = [x.double() @ w.double().t() for x, w in zip(Xs, Ws)]
    if case == "plain":
        bs = [rnd(N, seed=60 + g) for g in range(G)]
        ops.gemm(Xs, Bt, Cs, b_kstrided=True, bias=bs, b_split=imgs)
        ref = [r + b.double() for r, b in zip(ref, bs)]
    elif case == "gather":
        nn = 999
        P = rnd(nn, 4 * N, seed=7)
        g_ = torch.Generator().manual_seed(3)
        tgt = torch.sort(torch.randint(0, nn, (M,), generator=g_)).values.to(torch.int32).to(dev())
        src = torch.randint(0, nn, (M,), generator=g_).to(torch.int32).to(dev())
        ops.gemm(Xs, Bt, Cs, b_kstrided=True, gather_i=[P[:, :N], P[:, N:2 * N]], gather_j=[P[:, 2 * N:3 * N], P[:, 3 * N:]],
                 tgt=tgt, src=src, b_split=imgs)
        Pd = P.double()
        ref = [ref[0] + Pd[tgt.long(), :N] + Pd[src.long(), 2 * N:3 * N], ref[1] + Pd[tgt.long(), N:2 * N] + Pd[src.long(), 3 * N:]]
    elif case == "dact_colsum":
        pre = [rnd(M, N, seed=70 + g) for g in range(G)]
        tiles = ops.gemm_tiles_m(M)
        cs = [torch.full((tiles * N,), float("nan"), dtype=torch.float64, device=dev()) for _ in range(G)]
        ops.gemm(Xs, Bt, Cs, b_kstrided=True, dact=pre, colsum=cs, b_split=imgs)
        ref = [r * dsilu64(p_.double()) for r, p_ in zip(ref, pre)]
        for g in range(G):
            assert rel_err(cs[g].view(tiles, N).sum(0), ref[g].sum(0)) < 1e-6
    else:
        resid = rnd(M, N, seed=80)
        X2 = rnd(M, 2 * K, seed=81)
        W2 = [rnd(K, N, seed=82 + s, scale=0.1) for s in range(2)]
        C1 = torch.full((M, N), float("nan"), device=dev())
        ops.gemm([X2[:, :K], X2[:, K:]], W2, C1, b_kstrided=True, segments=True, resid=resid,
                 b_split_folded=torch.cat(ops.pack_b(W2)))
        r2 = X2[:, :K].double() @ W2[0].double() + X2[:, K:].double() @ W2[1].double() + resid.double()
        assert rel_err(C1, r2) < TOL
        return
    for g in range(G):
        assert rel_err(Cs[g], ref[g]) < TOL, g


@pytest.mark.parametrize("precision,M,with_resid,image", [(0, 40000, True, True), (0, 40000, False, True), (1, 40000, True, True),
                                                          (0, 333, True, False), (0, 12416, False, True)])
def test_gemm_softplus_backward_epilogue(ops, precision, M, with_resid, image):
    """CartnetGemmArgs.dact_kind = 1 (ABI 10; iComformer's RBF branches): v = (sum_s X_s W_s (+ resid)) * sigmoid(pre) with the
    bias-gradient column sums, two folded K-segments -- the form cartnet_icomformer_backward gives the 128-wide fp32 kernel
    (compiled kinds 268 / 270), the bf16x3 kernel and the general kernel (flag read at run time)."""
    K, N = 256, 256
    X = rnd(M, 2 * K, seed=3)
    Ws = [rnd(K, N, seed=20 + i, scale=0.1) for i in range(2)]
    resid = rnd(M, N, seed=5) if with_resid else None
    pre = rnd(M, N, seed=6, scale=6.0)
    tiles = ops.gemm_tiles_m(M)
    cs = torch.full((tiles * N,), float("nan"), dtype=torch.float64, device=dev())
    C_ = torch.full((M, N), float("nan"), device=dev())
    img = torch.cat((ops.pack_b if precision == 0 else ops.split_b)(Ws)) if image else None
    ops.gemm([X[:, :K], X[:, K:]], Ws, C_, b_kstrided=True, segments=True, resid=resid, dact=pre, colsum=cs,
             b_split_folded=img, precision=precision, dact_kind=1)
    ref = X[:, :K].double() @ Ws[0].double() + X[:, K:].double() @ Ws[1].double()
    if with_resid:
        ref = ref + resid.double()
    ref = ref * torch.sigmoid(pre.double())
    assert rel_err(C_, ref) < TOL
    assert rel_err(cs.view(tiles, N).sum(0), C_.double().sum(0)) < 1e-6
    with pytest.raises(RuntimeError, match="dact_kind"):
        ops.gemm([X[:, :K], X[:, K:]], Ws, C_, b_kstrided=True, segments=True, dact=pre, dact_kind=2)


@pytest.mark.parametrize("precision,M,image", [(0, 40000, True), (1, 40000, True), (0, 192, False), (0, 12416, True)])
def test_gemm_softplus_forward_epilogue(ops, precision, M, image):
    """dact_kind = 1 with cpre + out_act: pre = X W + b kept, out = softplus(pre) (threshold 20) -- an RBF branch of iComformer
    forward in one launch; the values are those of cartnet_eltwise op 0 on the kept pre-activation, bitwise."""
    K, N = 256, 256
    X, W, b = rnd(M, K, seed=3, scale=3.0), rnd(K, N, seed=4, scale=0.4), rnd(N, seed=5)
    pre = torch.full((M, N), float("nan"), device=dev())
    out = torch.full((M, N), float("nan"), device=dev())
    img = (ops.pack_b if precision == 0 else ops.split_b)([W])[0] if image else None
    ops.gemm(X, W, out, b_kstrided=True, bias=b, cpre=pre, out_act=True, b_split=img, precision=precision, dact_kind=1)
    ref = X.double() @ W.double() + b.double()
    assert rel_err(pre, ref) < TOL
    assert float(pre.max()) > 20.0 and float(pre.min()) < -10.0          # both tails of the softplus are exercised
    assert rel_err(out, torch.nn.functional.softplus(ref)) < TOL
    chk = torch.empty_like(out)
    ops.eltwise(0, pre, None, chk)
    assert torch.equal(out, chk)


@pytest.mark.parametrize("precision", [0, 1, 2])
def test_gemm_folded_segments_with_images(ops, precision):
    """sum_s X[:, sK:(s+1)K] W_s: K-segments that are adjacent column blocks run as one product (b_split_folded); the
    same call without the folded image must give the same numbers (segment form)."""
    M, K, N, S = 700, 256, 256, 4
    X = rnd(M, S * K, seed=1)
    Ws = [rnd(K, N, seed=2 + s, scale=0.1) for s in range(S)]         # operands as [K, N] (backward form)
    segs = [X[:, s * K:(s + 1) * K] for s in range(S)]
    resid = rnd(M, N, seed=9)
    ref = sum(a.double() @ w.double() for a, w in zip(segs, Ws)) + resid.double()
    make = ops.pack_b if precision == 0 else ops.split_b
    folded = torch.cat(make(Ws))
    C1 = torch.empty(M, N, device=dev())
    ops.gemm(segs, Ws, C1, b_kstrided=True, segments=True, resid=resid, precision=precision, b_split_folded=folded)
    tol = TOL if precision < 2 else 2e-2
    assert rel_err(C1, ref) < tol
    if precision == 2:
        assert rel_err(C1, ref) > 1e-6          # the bf16 kernel really ran
    C2 = torch.empty(M, N, device=dev())
    ops.gemm(segs, Ws, C2, b_kstrided=True, segments=True, resid=resid, precision=min(precision, 1))
    assert rel_err(C2, ref) < TOL


@pytest.mark.parametrize("precision", [0, 1])
@pytest.mark.parametrize("E,M,N,splitk,act", [(5000, 80, 512, 8, False), (3001, 128, 256, 5, True), (777, 300, 256, 3, False),
                                              (4096, 256, 512, 1, False)])
def test_gemm_weight_gradient_ragged_rows(ops, precision, E, M, N, splitk, act):
    """dW = dY^T (silu?)(X) with a ragged last row tile (M % 128 != 0), two column tiles, a K tail (E % 16 != 0) and
    the un-split form, at both precisions (the transposing-read kernel takes all of them at precision 1)."""
    dY, X = rnd(E, M, seed=3), rnd(E, N, seed=4)
    ref = dY.double().t() @ (silu64(X.double()) if act else X.double())
    out = torch.full((M, N), float("nan"), device=dev())
    if splitk > 1:
        slabs = torch.full((splitk * M, N), float("nan"), device=dev())
        ops.gemm(dY, X, slabs, a_kstrided=True, b_kstrided=True, b_act=act, splitk=splitk, precision=precision)
        ops.splitk_reduce(slabs, splitk, out)
    else:
        ops.gemm(dY, X, out, a_kstrided=True, b_kstrided=True, b_act=act, precision=precision)
    assert rel_err(out, ref) < TOL


def test_colsum_finalize(ops):
    parts = rnd(37, 100, seed=1).double()
    out = torch.empty(100, device=dev())
    ops.colsum_finalize(parts, 37, out)
    assert rel_err(out, parts.double().sum(0)) < 1e-6


# --------------------------------------------------------------------------------------------------- graph layout
def _random_graph_batch(n_graphs, n_lo, n_hi, deg, seed):
    g = torch.Generator().manual_seed(seed)
    srcs, tgts, ptr = [], [], [0]
    for _ in range(n_graphs):
        n = int(torch.randint(n_lo, n_hi + 1, (1,), generator=g))
        d = torch.randint(0, deg + 1, (n,), generator=g)
        t = torch.repeat_interleave(torch.arange(n), d)
        s = torch.randint(0, n, (int(d.sum()),), generator=g)
        srcs.append(s + ptr[-1])
        tgts.append(t + ptr[-1])
        ptr.append(ptr[-1] + n)
    ei = torch.stack([torch.cat(srcs), torch.cat(tgts)]).to(torch.int64)
    return ei, torch.tensor(ptr, dtype=torch.int64)


@pytest.mark.parametrize("n_graphs,n_lo,n_hi,deg", [(1, 5, 5, 3), (7, 1, 60, 20), (3, 300, 700, 30), (4, 1, 3, 0)])
def test_csr_csc_build_exact(ops, n_graphs, n_lo, n_hi, deg):
    ei, ptr = _random_graph_batch(n_graphs, n_lo, n_hi, deg, seed=n_graphs)
    N, E = int(ptr[-1]), ei.shape[1]
    lay = ops.GraphLayout(ei.to(dev()), N, ptr.to(dev()))
    lay.validate()
    assert torch.equal(lay.src[:E].cpu().long(), ei[0])
    assert torch.equal(lay.tgt[:E].cpu().long(), ei[1])
    rowptr = torch.zeros(N + 1, dtype=torch.int64)
    rowptr[1:] = torch.cumsum(torch.bincount(ei[1], minlength=N), 0)
    assert torch.equal(lay.rowptr.cpu().long(), rowptr)
    colptr = torch.zeros(N + 1, dtype=torch.int64)
    colptr[1:] = torch.cumsum(torch.bincount(ei[0], minlength=N), 0)
    assert torch.equal(lay.colptr.cpu().long(), colptr)
    perm = torch.argsort(ei[0], stable=True)
    assert torch.equal(lay.perm[:E].cpu().long(), perm)


def test_csr_build_flags_unsorted_and_out_of_range(ops):
    ei = torch.tensor([[0, 1, 2], [2, 1, 0]], dtype=torch.int64)
    lay = ops.GraphLayout(ei.to(dev()), 3, None)
    with pytest.raises(ValueError, match="sorted"):
        lay.validate()
    ei = torch.tensor([[0, 5], [0, 1]], dtype=torch.int64)
    lay = ops.GraphLayout(ei.to(dev()), 3, None)
    with pytest.raises(ValueError, match="outside"):
        lay.validate()


def test_csr_empty_graph(ops):
    ei = torch.zeros(2, 0, dtype=torch.int64)
    lay = ops.GraphLayout(ei.to(dev()), 4, torch.tensor([0, 4]).to(dev()))
    lay.validate()
    assert lay.rowptr.cpu().tolist() == [0, 0, 0, 0, 0]
    assert lay.colptr.cpu().tolist() == [0, 0, 0, 0, 0]


# --------------------------------------------------------------------------------------------------- edge ops
@pytest.mark.parametrize("invariant", [False, True])
def test_edge_features_vs_oracle(ops, invariant):
    from oracle import cartnet_ref as orc
    E, R, radius = 1000, 64, 5.0
    g = torch.Generator().manual_seed(0)
    dist = (0.01 + 5.2 * torch.rand(E, generator=g))
    dist[:3] = torch.tensor([5.0, 4.9999995, 0.0101])
    dirv = torch.nn.functional.normalize(torch.randn(E, 3, generator=g), dim=-1)
    means, betas = orc.rbf_constants(radius, R)
    ldf = 72
    feat = torch.full((E, ldf), 7.0, device=dev())
    env = torch.empty(E, device=dev())
    ops.edge_features(dist.to(dev()), dirv.to(dev()), means.to(dev()), betas.to(dev()), invariant, radius, 4.0, feat,
                      env)
    rbf = orc.exp_normal_smearing(dist.double(), means.double(), betas.double(), radius)
    assert rel_err(feat[:, :R], rbf) < TOL
    if invariant:
        assert torch.count_nonzero(feat[:, R:]) == 0
    else:
        assert torch.equal(feat[:, R:R + 3].cpu(), dirv)
        assert torch.count_nonzero(feat[:, R + 3:]) == 0
    assert rel_err(env, orc.cosine_cutoff(dist.double(), 4.0)) < TOL


def _gate_reference(gs, e_in, env, tgt, N, mean, rstd, gamma, beta):
    D = e_in.shape[1]
    g, s = gs[:, :D], gs[:, D:]
    sig = env[:, None] * torch.sigmoid((g - mean) * rstd * gamma + beta)
    aggr = torch.zeros(N, D, dtype=gs.dtype).index_add_(0, tgt, sig * s)
    return e_in + sig, aggr


@pytest.mark.parametrize("D", [16, 64, 256, 320])
def test_gate_scatter_fwd_bwd(ops, D):
    ei, ptr = _random_graph_batch(5, 10, 60, 25, seed=D)
    N, E = int(ptr[-1]), ei.shape[1]
    lay = ops.GraphLayout(ei.to(dev()), N, ptr.to(dev()))
    gs, e_in = rnd(E, 2 * D, seed=1), rnd(E, D, seed=2)
    env = torch.rand(E, generator=torch.Generator().manual_seed(3)).to(dev())
    mean, var = rnd(D, seed=4, scale=0.3), (0.5 + torch.rand(D, generator=torch.Generator().manual_seed(5))).to(dev())
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    mean_rstd = torch.cat([mean, rstd]).contiguous()
    gamma, beta = rnd(D, seed=6), rnd(D, seed=7)
    e_out, aggr = torch.empty(E, D, device=dev()), torch.empty(N, D, device=dev())
    nparts = ops.gate_nparts(N)
    ps, pq = (torch.zeros(nparts * D, device=dev(), dtype=torch.float64) for _ in range(2))
    ops.gate_scatter_fwd(gs, e_in, env, lay, mean_rstd, gamma, beta, e_out, aggr, ps, pq)

    c = lambda t: t.detach().double().cpu()
    gs64 = c(gs).requires_grad_(True)
    eo_ref, ag_ref = _gate_reference(gs64, c(e_in), c(env), ei[1], N, c(mean), c(rstd), c(gamma), c(beta))
    assert rel_err(e_out, eo_ref) < TOL
    assert rel_err(aggr, ag_ref) < TOL
    assert rel_err(ps.view(nparts, D).sum(0), ag_ref.sum(0)) < 1e-5
    assert rel_err(pq.view(nparts, D).sum(0), (ag_ref ** 2).sum(0)) < 1e-5

    # backward with "training" BatchNorm: mean/rstd are functions of g -> use batch statistics for the reference
    g64 = gs64[:, :D]
    bmean = g64.mean(0)
    bvar = g64.var(0, unbiased=False)
    brstd = torch.rsqrt(bvar + 1e-5)
    mean_rstd_b = torch.cat([bmean, brstd]).detach().float().to(dev()).contiguous()
    eo_ref, ag_ref = _gate_reference(gs64, c(e_in), c(env), ei[1], N, bmean, brstd, c(gamma), c(beta))
    de_out, daggr = rnd(E, D, seed=8), rnd(N, D, seed=9)
    (eo_ref * c(de_out)).sum().add((ag_ref * c(daggr)).sum()).backward()
    pa, pb = (torch.zeros(nparts * D, device=dev(), dtype=torch.float64) for _ in range(2))
    ops.gate_scatter_bwd_stats(gs, de_out, daggr, env, lay, mean_rstd_b, gamma, beta, pa, pb)
    sums = torch.empty(2 * D, device=dev())
    ops.colsum_finalize([pa, pb], nparts, [sums[:D], sums[D:]])
    gs_work = gs.clone()
    pdg, pds = (torch.zeros(nparts * D, device=dev(), dtype=torch.float64) for _ in range(2))
    ops.gate_scatter_bwd_apply(gs_work, de_out, daggr, env, lay, mean_rstd_b, gamma, beta, sums, True, pdg, pds)
    assert rel_err(gs_work, gs64.grad) < 2e-5
    assert rel_err(pds.view(nparts, D).sum(0), gs64.grad[:, D:].sum(0)) < 1e-4


def test_segment_sum(ops):
    ei, ptr = _random_graph_batch(6, 5, 80, 18, seed=11)
    N, E = int(ptr[-1]), ei.shape[1]
    lay = ops.GraphLayout(ei.to(dev()), N, ptr.to(dev()))
    W = 128
    rows = rnd(E, W, seed=1)
    out_t, out_s = torch.empty(N, W, device=dev()), torch.empty(N, 2 * W, device=dev())
    ops.segment_sum(rows, lay.rowptr, None, out_t)
    ops.segment_sum(rows, lay.colptr, lay.perm, out_s[:, W:])
    ref_t = torch.zeros(N, W, dtype=torch.float64).index_add_(0, ei[1], rows.double().cpu())
    ref_s = torch.zeros(N, W, dtype=torch.float64).index_add_(0, ei[0], rows.double().cpu())
    assert rel_err(out_t, ref_t) < TOL
    assert rel_err(out_s[:, W:], ref_s) < TOL


def _graph_with_degrees(degs, seed=0):
    """One crystal whose node t has exactly degs[t] incoming edges (sources drawn at random), target-sorted."""
    g = torch.Generator().manual_seed(seed)
    n = len(degs)
    tgt = torch.repeat_interleave(torch.arange(n), torch.tensor(degs))
    src = torch.randint(0, n, (int(tgt.numel()),), generator=g)
    return torch.stack([src, tgt]).to(torch.int64), torch.tensor([0, n], dtype=torch.int64)


@pytest.mark.parametrize("W", [256, 512, 320])
def test_segment_sum_every_batch_remainder_is_exact(ops, W):
    """Segment lengths 0..17, 24, 40 (the kernel loads 8 rows per round and clamps the remainder round) with
    integer-valued rows: sums are exact, so the comparison is bitwise, by target (ascending sweep) and by source
    (permuted rows, descending sweep)."""
    degs = list(range(18)) + [24, 40, 0, 8, 16, 1]
    ei, ptr = _graph_with_degrees(degs, seed=W)
    N, E = len(degs), ei.shape[1]
    lay = ops.GraphLayout(ei.to(dev()), N, ptr.to(dev()))
    g = torch.Generator().manual_seed(1)
    rows = torch.randint(-8, 9, (E, W), generator=g).float().to(dev())
    out_t, out_s = torch.full((N, W), 7.0, device=dev()), torch.full((N, W), 7.0, device=dev())
    ops.segment_sum(rows, lay.rowptr, None, out_t)
    ops.segment_sum(rows, lay.colptr, lay.perm, out_s)
    ref_t = torch.zeros(N, W).index_add_(0, ei[1], rows.cpu())
    ref_s = torch.zeros(N, W).index_add_(0, ei[0], rows.cpu())
    assert torch.equal(out_t.cpu(), ref_t) and torch.equal(out_s.cpu(), ref_s)
    # both sums in one launch (round 5: the model's backward), into the two halves of one matrix
    both = torch.full((N, 2 * W), 7.0, device=dev())
    ops.segment_sum_pair(rows, lay, both[:, :W], both[:, W:])
    assert torch.equal(both[:, :W].cpu(), ref_t) and torch.equal(both[:, W:].cpu(), ref_s)
    if W == 512:     # iComformer's layout: chunk j of 256 columns at column j * 512 -> [t0 | s0 | t1 | s1]
        inter = torch.full((N, 4 * 256), 7.0, device=dev())
        ops.segment_sum_pair(rows, lay, inter[:, :768], inter[:, 256:], ochunk=512)
        want = torch.cat([ref_t[:, :256], ref_s[:, :256], ref_t[:, 256:], ref_s[:, 256:]], 1)
        assert torch.equal(inter.cpu(), want)


def test_gate_scatter_fwd_every_batch_remainder(ops):
    """Forward gate with in-degrees 0..9 and 13 (rounds of 4 edges with a clamped remainder): against the fp64
    formula, and the e_out rows of a clamped round are each written exactly once (checked through e_out itself)."""
    D = 256
    degs = list(range(10)) + [13, 0, 4, 8]
    ei, ptr = _graph_with_degrees(degs, seed=3)
    N, E = len(degs), ei.shape[1]
    lay = ops.GraphLayout(ei.to(dev()), N, ptr.to(dev()))
    gs, e_in = rnd(E, 2 * D, seed=1), rnd(E, D, seed=2)
    env = torch.rand(E, generator=torch.Generator().manual_seed(3)).to(dev())
    gamma, beta = rnd(D, seed=4) * 0.2 + 1.0, rnd(D, seed=5) * 0.1
    mean, rstd = rnd(D, seed=6) * 0.1, torch.rand(D, generator=torch.Generator().manual_seed(7)).to(dev()) + 0.5
    mr = torch.cat([mean, rstd]).contiguous()
    nparts = ops.gate_nparts(N)
    e_out, aggr = torch.full((E, D), float("nan"), device=dev()), torch.full((N, D), float("nan"), device=dev())
    ps, pq = (torch.zeros(nparts * D, dtype=torch.float64, device=dev()) for _ in range(2))
    ops.gate_scatter_fwd(gs, e_in, env, lay, mr, gamma, beta, e_out, aggr, ps, pq)
    c = lambda t: t.detach().double().cpu()
    eo_ref, ag_ref = _gate_reference(c(gs), c(e_in), c(env), ei[1], N, c(mean), c(rstd), c(gamma), c(beta))
    assert torch.isfinite(e_out).all() and torch.isfinite(aggr).all()
    assert rel_err(e_out, eo_ref) < TOL and rel_err(aggr, ag_ref) < TOL
    assert rel_err(ps.view(nparts, D).sum(0), ag_ref.sum(0)) < 1e-6


# --------------------------------------------------------------------------------------------------- node ops
def test_bn_finalize_and_node_update(ops):
    N, D = 777, 64
    aggr, x_in = rnd(N, D, seed=1, scale=2.0) + 0.5, rnd(N, D, seed=2)
    gamma, beta = rnd(D, seed=3), rnd(D, seed=4)
    # statistics from 5 partial blocks
    chunks = torch.chunk(aggr, 5, dim=0)
    ps = torch.stack([ch.double().sum(0) for ch in chunks]).contiguous().view(-1)
    pq = torch.stack([(ch.double() ** 2).sum(0) for ch in chunks]).contiguous().view(-1)
    rm, rv = torch.zeros(D, device=dev()), torch.ones(D, device=dev())
    nbt = torch.zeros(1, dtype=torch.int64, device=dev())
    mean_rstd = torch.empty(2 * D, device=dev())
    ops.bn_finalize(ps, pq, 5, N, D, 1e-5, 0.1, True, rm, rv, nbt, mean_rstd)
    a64 = aggr.double().cpu().requires_grad_(True)
    bn = torch.nn.BatchNorm1d(D).double()
    with torch.no_grad():
        bn.weight.copy_(gamma.double().cpu())
        bn.bias.copy_(beta.double().cpu())
    xn = bn(a64)
    assert rel_err(mean_rstd[:D], a64.mean(0)) < 1e-5
    assert rel_err(mean_rstd[D:], torch.rsqrt(a64.var(0, unbiased=False) + 1e-5)) < 1e-5
    assert rel_err(rm, bn.running_mean) < 1e-5 and rel_err(rv, bn.running_var) < 1e-5
    assert int(nbt.item()) == 1
    x_out = torch.empty(N, D, device=dev())
    ops.node_update_fwd(aggr, x_in, mean_rstd, gamma, beta, x_out)
    ref = torch.nn.functional.silu(xn) + x_in.double().cpu()
    assert rel_err(x_out, ref) < TOL
    # backward
    dx = rnd(N, D, seed=5)
    (ref * dx.double().cpu()).sum().backward()
    nparts = ops.node_nparts(N)
    pa, pb = (torch.zeros(nparts * D, device=dev(), dtype=torch.float64) for _ in range(2))
    ops.node_update_bwd_stats(aggr, dx, mean_rstd, gamma, beta, pa, pb)
    sums = torch.empty(2 * D, device=dev())
    ops.colsum_finalize([pa, pb], nparts, [sums[:D], sums[D:]])
    daggr = torch.empty(N, D, device=dev())
    ops.node_update_bwd_apply(aggr, dx, mean_rstd, gamma, beta, sums, True, daggr)
    assert rel_err(daggr, a64.grad) < 2e-5
    assert rel_err(sums[:D], bn.bias.grad) < 2e-5
    assert rel_err(sums[D:], bn.weight.grad) < 2e-5
    # eval mode uses running statistics
    ops.bn_finalize(None, None, 0, 0, D, 1e-5, 0.1, False, rm, rv, None, mean_rstd)
    assert rel_err(mean_rstd[:D], rm) == 0
    assert rel_err(mean_rstd[D:], torch.rsqrt(rv.double() + 1e-5)) < 1e-6


@pytest.mark.parametrize("N", [500, 5, 3000])
def test_node_embed_fwd_bwd(ops, N):
    Cc, Bg = 128, 6
    g = torch.Generator().manual_seed(0)
    z = torch.where(torch.rand(N, generator=g) < 0.8, torch.randint(1, 4, (N,), generator=g),
                    torch.randint(1, 119, (N,), generator=g))      # a few very common elements, like H/C/N/O
    batch = torch.sort(torch.randint(0, Bg, (N,), generator=g)).values
    T = torch.randn(Bg, generator=g)
    emb, wt, bt = rnd(119, Cc, seed=1), rnd(Cc, 1, seed=2), rnd(Cc, seed=3)
    x0 = torch.empty(N, Cc, device=dev())
    ops.node_embed(z.to(dev()), batch.to(dev()), T.to(dev()), emb, wt, bt, None, x0)
    e64, w64, b64 = (t.double().cpu().requires_grad_(True) for t in (emb, wt, bt))
    ref = e64[z] + (T.double()[:, None] @ w64.t() + b64)[batch]
    assert rel_err(x0, ref) < 1e-6
    dx0 = rnd(N, Cc, seed=4)
    (ref * dx0.double().cpu()).sum().backward()
    demb = torch.empty(119, Cc, device=dev())
    nparts = ops.node_nparts(N)
    pw, pb = (torch.zeros(nparts * Cc, device=dev(), dtype=torch.float64) for _ in range(2))
    ops.node_embed_bwd(batch.to(dev()), T.to(dev()), dx0, pw, pb)
    perm, zptr, status = ops.sort_by_key(z.to(dev()), 119)
    assert int(status.item()) == 0
    assert torch.equal(perm[:N].cpu().long(), torch.argsort(z, stable=True))
    assert torch.equal(zptr.cpu().long()[1:], torch.cumsum(torch.bincount(z, minlength=119), 0))
    ops.segment_sum_long(dx0, zptr, perm, N, demb)
    dwt, dbt = torch.empty(Cc, device=dev()), torch.empty(Cc, device=dev())
    ops.colsum_finalize([pw, pb], nparts, [dwt, dbt])
    assert rel_err(demb, e64.grad) < TOL
    assert rel_err(dwt, w64.grad.view(-1)) < TOL
    assert rel_err(dbt, b64.grad) < TOL


@pytest.mark.parametrize("total,nseg,W,ld", [(1, 1, 4, 4), (31, 3, 256, 256), (32, 1, 512, 1536), (1000, 7, 516, 520),
                                              (50000, 64, 1536, 1536), (97, 40, 8, 8)])
def test_segment_sum_chunked_equals_the_long_form_and_torch(ops, total, nseg, W, ld):
    """cartnet_segment_sum_chunked (numbered partial rows, workspace of chunks + segments rows): bitwise the sums of
    cartnet_segment_sum_long on the same input -- same chunks, same order -- and torch's index_add_ within rounding; with
    empty segments, segments that start inside a chunk, a permutation, strided rows."""
    g = torch.Generator().manual_seed(total + nseg)
    rows = rnd(total, ld, seed=total)[:, :W]
    cuts = torch.sort(torch.randint(0, total + 1, (nseg - 1,), generator=g)).values if nseg > 1 else torch.zeros(0, dtype=torch.int64)
    ptr = torch.cat([torch.zeros(1, dtype=torch.int64), cuts, torch.tensor([total])]).int().to(dev())
    for use_perm in (False, True):
        perm = torch.randperm(total, generator=g).int().to(dev()) if use_perm else None
        a = torch.full((nseg, W), float("nan"), device=dev())
        b = torch.full((nseg, W), float("nan"), device=dev())
        ops.segment_sum_chunked(rows, ptr, perm, total, a)
        ops.segment_sum_long(rows.contiguous(), ptr, perm, total, b)
        assert torch.equal(a, b)
        seg = torch.repeat_interleave(torch.arange(nseg), (ptr[1:] - ptr[:-1]).cpu().long())
        src = rows.double().cpu() if perm is None else rows.double().cpu()[perm.cpu().long()]
        ref = torch.zeros(nseg, W, dtype=torch.float64).index_add_(0, seg, src)
        assert rel_err(a, ref) < 1e-5
    lib = __import__("cartnet_amd.lib", fromlist=["load"]).load()
    assert int(lib.cartnet_segment_chunked_rows(nseg, total)) == (total + 31) // 32 + nseg


@pytest.mark.parametrize("total,nseg,W", [(1, 1, 12), (31, 3, 768), (32, 1, 1536), (1000, 7, 516), (50000, 64, 1536), (97, 40, 24)])
def test_segment_sum_chunked_fold3_is_both_passes_in_one(ops, total, nseg, W):
    """Round 5 (iComformer's edge layer backward): the per-segment sums of cartnet_segment_sum_chunked AND every row's sum
    over its three pieces from ONE read -- bitwise the chunked sums, bitwise (a + b) + c for the fold."""
    g = torch.Generator().manual_seed(3 * total + nseg)
    rows = rnd(total, W, seed=total + 1)
    cuts = torch.sort(torch.randint(0, total + 1, (nseg - 1,), generator=g)).values if nseg > 1 else torch.zeros(0, dtype=torch.int64)
    ptr = torch.cat([torch.zeros(1, dtype=torch.int64), cuts, torch.tensor([total])]).int().to(dev())
    a = torch.full((nseg, W), float("nan"), device=dev())
    b = torch.full((nseg, W), float("nan"), device=dev())
    fold = torch.full((total, W // 3), float("nan"), device=dev())
    ops.segment_sum_chunked_fold3(rows, ptr, total, a, fold)
    ops.segment_sum_chunked(rows, ptr, None, total, b)
    assert torch.equal(a, b)
    q = W // 3
    assert torch.equal(fold, (rows[:, :q] + rows[:, q:2 * q]) + rows[:, 2 * q:])


@pytest.mark.parametrize("R,Cc", [(1, 4), (777, 256), (12416, 256), (300, 320)])
def test_coldot_bc(ops, R, Cc):
    """cartnet_coldot_bc_partial (iComformer's conv layers: sum_t daggr[t] B[t], sum_t daggr[t] C[t]) against fp64."""
    d, bc = rnd(R, Cc, seed=R), rnd(R, 2 * Cc, seed=R + 1)
    a, b = torch.full((Cc,), 9.0, device=dev()), torch.full((Cc,), 9.0, device=dev())
    ops.coldot_bc(d, bc, a, b)
    d64, bc64 = d.double().cpu(), bc.double().cpu()
    assert rel_err(a, (d64 * bc64[:, :Cc]).sum(0)) < 1e-6 and rel_err(b, (d64 * bc64[:, Cc:]).sum(0)) < 1e-6


@pytest.mark.parametrize("S,Cc,maxdeg", [(1, 4, 3), (500, 256, 30), (3000, 256, 1), (12416, 128, 40)])
def test_rowmul_bwd_sums_is_the_plain_pass_plus_both_column_sums(ops, S, Cc, maxdeg):
    """cartnet_rowmul_bwd_sums (iComformer: the bias gradients of key_update.2 and lin_query out of the pass that writes
    dkey and dq): dkey / dq bitwise those of cartnet_rowmul_bwd, the sums against fp64."""
    g = torch.Generator().manual_seed(S + Cc)
    deg = torch.randint(0, maxdeg + 1, (S,), generator=g)
    ptr = torch.cat([torch.zeros(1, dtype=torch.int64), deg.cumsum(0)]).int().to(dev())
    R = int(deg.sum())
    buf = rnd(max(R, 1), 2 * Cc, seed=1)[:R]
    key, q = rnd(max(R, 1), 2 * Cc, seed=2)[:R, :Cc], rnd(S, 3 * Cc, seed=3)[:, :Cc]
    a, b = buf.clone(), buf.clone()
    dqa, dqb = (torch.full((S, 3 * Cc), float("nan"), device=dev())[:, :Cc] for _ in range(2))
    sk, sq = torch.full((Cc,), 9.0, device=dev()), torch.full((Cc,), 9.0, device=dev())
    ops.rowmul_bwd(a[:, :Cc], key, q, ptr, 0.0625, dqa)
    ops.rowmul_bwd(b[:, :Cc], key, q, ptr, 0.0625, dqb, sum_dkey=sk, sum_dq=sq)
    assert torch.equal(a, b) and torch.equal(dqa, dqb)
    if R == 0:
        assert float(sk.abs().max()) == 0.0 and float(sq.abs().max()) == 0.0 and float(dqb.abs().max()) == 0.0
    else:
        assert rel_err(sk, a[:, :Cc].double().cpu().sum(0)) < 1e-6
        assert rel_err(sq, dqa.double().cpu().sum(0)) < 1e-6


@pytest.mark.parametrize("S,D,maxdeg,training", [(3, 8, 4, True), (700, 256, 30, True), (700, 256, 30, False), (5000, 320, 3, True)])
def test_att_gate_bwd_apply_is_gate_apply_then_rowmul(ops, S, D, maxdeg, training):
    """cartnet_att_gate_bwd_apply (iComformer, round 5) against the two passes it replaces -- cartnet_gate_scatter_bwd_apply
    (no edge residual, no envelope) then cartnet_rowmul_bwd -- and its three column sums against fp64."""
    from cartnet_amd.ops import GraphLayout
    g = torch.Generator().manual_seed(S + D)
    deg = torch.randint(0 if S > 3 else 1, maxdeg + 1, (S,), generator=g)
    ptr = torch.cat([torch.zeros(1, dtype=torch.int64), deg.cumsum(0)]).int().to(dev())
    R = int(deg.sum())
    gs0, key, q, daggr = rnd(R, 2 * D, seed=1), rnd(R, 2 * D, seed=2)[:, :D], rnd(S, 3 * D, seed=3)[:, :D], rnd(S, D, seed=4)
    mr = torch.cat([rnd(D, seed=5, scale=0.1), rnd(D, seed=6).abs() + 0.5])
    gam, bet, sums = rnd(D, seed=7), rnd(D, seed=8), rnd(2 * D, seed=9)
    scale = 1.0 / D ** 0.5
    # reference: the two passes
    lay = GraphLayout.__new__(GraphLayout)
    lay.E, lay.N, lay.rowptr = R, S, ptr
    a = gs0.clone()
    npart = ops.gate_nparts(S)
    pg, ps = (torch.empty(npart * D, dtype=torch.float64, device=dev()) for _ in range(2))
    ops.gate_scatter_bwd_apply(a, None, daggr, None, lay, mr, gam, bet, sums, training, pg, ps)
    dqa = torch.empty(S, D, device=dev())
    ops.rowmul_bwd(a[:, :D], key, q, ptr, scale, dqa)
    # fused
    b = gs0.clone()
    dqb = torch.full((S, 3 * D), float("nan"), device=dev())[:, :D]
    sk, sm, sq = (torch.full((D,), 9.0, device=dev()) for _ in range(3))
    ops.att_gate_bwd_apply(b, key, q, daggr, ptr, mr, gam, bet, sums, R, training, scale, dqb, sk, sm, sq)
    assert rel_err(b, a) < 2e-6 and rel_err(dqb, dqa) < 2e-6
    assert rel_err(sk, a[:, :D].double().cpu().sum(0)) < 1e-5
    assert rel_err(sm, a[:, D:].double().cpu().sum(0)) < 1e-5
    assert rel_err(sq, dqa.double().cpu().sum(0)) < 1e-5
    # key = None: gs carries [key | msg] and alpha = key q[s] scale is recomputed (the alpha-free forward)
    rep = torch.repeat_interleave(torch.arange(S, device=dev()), (ptr[1:] - ptr[:-1]).long())
    c = torch.cat([key, gs0[:, D:]], dim=1).contiguous()
    g2 = c.clone()
    g2[:, :D] = key * (q[rep] * scale)          # what the stored-alpha form would have been given
    dq2, dq3 = torch.empty(S, D, device=dev()), torch.empty(S, D, device=dev())
    ops.att_gate_bwd_apply(g2, key, q, daggr, ptr, mr, gam, bet, sums, R, training, scale, dq2, sk, sm, sq)
    ops.att_gate_bwd_apply(c, None, q, daggr, ptr, mr, gam, bet, sums, R, training, scale, dq3, sk, sm, sq)
    assert rel_err(c, g2) < 2e-6 and rel_err(dq3, dq2) < 2e-6


@pytest.mark.parametrize("S,D,maxdeg,with_bc", [(3, 8, 4, True), (700, 256, 30, True), (700, 256, 30, False), (5000, 320, 3, True)])
def test_att_gate_fwd_is_rowmul_then_gate(ops, S, D, maxdeg, with_bc):
    """cartnet_att_gate_fwd (alpha recomputed from the key rows) against cartnet_rowmul_fwd + cartnet_gate_scatter_fwd(_bc);
    the statistics of cartnet_rowmul_fwd with alpha = None are those of the storing pass, bitwise."""
    from cartnet_amd.ops import GraphLayout
    g = torch.Generator().manual_seed(S + D + 1)
    deg = torch.randint(0 if S > 3 else 1, maxdeg + 1, (S,), generator=g)
    ptr = torch.cat([torch.zeros(1, dtype=torch.int64), deg.cumsum(0)]).int().to(dev())
    R = int(deg.sum())
    gs = rnd(R, 2 * D, seed=1)                     # [key | msg]
    q = rnd(S, 3 * D, seed=3)[:, :D]
    gam, bet = rnd(D, seed=7), rnd(D, seed=8)
    scale = 1.0 / D ** 0.5
    npart = ops.segment_nparts(S)
    ps, pq, ps2, pq2 = (torch.empty(npart * D, dtype=torch.float64, device=dev()) for _ in range(4))
    ref = gs.clone()
    ops.rowmul_fwd(gs[:, :D], q, ptr, scale, ref[:, :D], ps, pq)
    ops.rowmul_fwd(gs[:, :D], q, ptr, scale, None, ps2, pq2)
    assert torch.equal(ps, ps2) and torch.equal(pq, pq2)
    cnt = max(R, 1)
    mean = ps.view(npart, D).sum(0) / cnt
    var = (pq.view(npart, D).sum(0) / cnt - mean * mean).clamp_min(0)
    mr = torch.cat([mean, 1.0 / torch.sqrt(var + 1e-5)]).float()
    lay = GraphLayout.__new__(GraphLayout)
    lay.E, lay.N, lay.rowptr = R, S, ptr
    gp = ops.gate_nparts(S)
    d1, d2 = (torch.empty(gp * D, dtype=torch.float64, device=dev()) for _ in range(2))
    a1, a2 = torch.empty(S, D, device=dev()), torch.full((S, D), float("nan"), device=dev())
    b1 = torch.empty(S, 2 * D, device=dev()) if with_bc else None
    b2 = torch.full((S, 2 * D), float("nan"), device=dev()) if with_bc else None
    ops.gate_scatter_fwd(ref, None, None, lay, mr, gam, bet, None, a1, d1, d2, bc=b1)
    ops.att_gate_fwd(gs, q, ptr, mr, gam, bet, scale, a2, bc=b2)
    assert rel_err(a2, a1) < 2e-6
    if with_bc:
        assert rel_err(b2, b1) < 1e-5


@pytest.mark.parametrize("R,Cc", [(1, 4), (777, 256), (20000, 256), (300, 320)])
def test_softplus_bwd_sums(ops, R, Cc):
    """cartnet_softplus_bwd_sums: out bitwise cartnet_eltwise op 1, its column sums against fp64."""
    a, b = rnd(R, Cc, seed=R), rnd(R, Cc, seed=R + 1, scale=8.0)
    o1, o2 = torch.empty_like(a), torch.empty_like(a)
    s = torch.full((Cc,), 9.0, device=dev())
    ops.eltwise(1, a, b, o1)
    ops.softplus_bwd_sums(a, b, o2, s)
    assert torch.equal(o1, o2)
    assert rel_err(o1, a.double().cpu() * torch.sigmoid(b.double().cpu())) < 1e-6
    assert rel_err(s, o1.double().cpu().sum(0)) < 1e-6


@pytest.mark.parametrize("N,D,training", [(1, 8, True), (999, 256, True), (999, 256, False), (12416, 128, True)])
def test_softplus_update_bwd_apply_sums(ops, N, D, training):
    """cartnet_softplus_update_bwd_apply_sums: d_o / dx bitwise the plain apply pass; sum_do against fp64 of d_o (in
    training mode that sum is rounding noise around zero -- BatchNorm's output does not move with a constant added to its
    input -- so the comparison is absolute, against the scale of |d_o| summed)."""
    o, x, dy = rnd(N, D, seed=1), rnd(N, D, seed=2), rnd(N, D, seed=3)
    mr = torch.cat([rnd(D, seed=4, scale=0.1), rnd(D, seed=5).abs() + 0.5])
    gam, bet, sums = rnd(D, seed=6), rnd(D, seed=7), rnd(2 * D, seed=8)
    d1, d2, x1, x2 = (torch.empty(N, D, device=dev()) for _ in range(4))
    sd = torch.full((D,), 9.0, device=dev())
    ops.softplus_update_bwd_apply(o, x, dy, mr, gam, bet, sums, training, d1, None, x1)
    ops.softplus_update_bwd_apply(o, x, dy, mr, gam, bet, sums, training, d2, None, x2, sum_do=sd)
    assert torch.equal(d1, d2) and torch.equal(x1, x2)
    ref = d1.double().cpu().sum(0)
    scale = d1.double().cpu().abs().sum(0).max()
    assert float((sd.double().cpu() - ref).abs().max()) <= 1e-6 * float(scale)


@pytest.mark.parametrize("H", [8, 32, 128])
def test_cholesky_head_fwd_bwd(ops, H):
    from oracle import cartnet_ref as orc
    N = 301
    g = torch.Generator().manual_seed(H)
    mask = torch.rand(N, generator=g) < 0.55
    hid = rnd(N, H, seed=1)
    W2, b2 = rnd(6, H, seed=2, scale=0.3), rnd(6, seed=3)
    idx = torch.empty(N, dtype=torch.int32, device=dev())
    cnt = torch.zeros(1, dtype=torch.int32, device=dev())
    ops.mask_index(mask.to(dev()), idx, cnt)
    M = int(mask.sum())
    assert int(cnt.item()) == M
    exp_idx = torch.full((N,), -1, dtype=torch.int64)
    exp_idx[mask] = torch.arange(M)
    assert torch.equal(idx.cpu().long(), exp_idx)
    p6, pred = torch.empty(M, 6, device=dev()), torch.empty(M, 3, 3, device=dev())
    ops.cholesky_head_fwd(hid, idx, W2, b2, p6, pred)
    h64 = hid.double().cpu().requires_grad_(True)
    w64, bb64 = W2.double().cpu().requires_grad_(True), b2.double().cpu().requires_grad_(True)
    # oracle head = Linear(D->H) [identity here] -> SiLU -> Linear(H->6) -> softplus / L^T L
    sd = {"head.MLP.0.weight": torch.eye(H, dtype=torch.float64), "head.MLP.0.bias": torch.zeros(H, dtype=torch.float64),
          "head.MLP.2.weight": w64, "head.MLP.2.bias": bb64}
    ref = orc.cholesky_head(sd, h64, mask)
    assert rel_err(pred, ref) < TOL
    dpred = rnd(M, 3, 3, seed=4)
    (ref * dpred.double().cpu()).sum().backward()
    dhid = torch.empty(N, H, device=dev())
    nparts = ops.node_nparts(N)
    parts = torch.zeros(nparts * (7 * H + 8), device=dev())
    ops.cholesky_head_bwd(hid, idx, W2, p6, dpred.contiguous(), dhid, parts)
    assert rel_err(dhid, h64.grad) < TOL
    tot = parts.view(nparts, 7 * H + 8).double().sum(0)
    assert rel_err(tot[6 * H + 8:], h64.grad.sum(0)) < TOL
    assert rel_err(tot[:6 * H].view(6, H), w64.grad) < TOL
    assert rel_err(tot[6 * H:6 * H + 6], bb64.grad) < TOL


def test_scalar_head_fwd_bwd(ops):
    from oracle import cartnet_ref as orc
    H, Bg = 32, 9
    g = torch.Generator().manual_seed(1)
    sizes = torch.randint(1, 12, (Bg,), generator=g)
    ptr = torch.zeros(Bg + 1, dtype=torch.int64)
    ptr[1:] = torch.cumsum(sizes, 0)
    N = int(ptr[-1])
    batch = torch.repeat_interleave(torch.arange(Bg), sizes)
    hid, w2, b2 = rnd(N, H, seed=2), rnd(1, H, seed=3), rnd(1, seed=4)
    out = torch.empty(Bg, device=dev())
    ops.scalar_head_fwd(hid, w2, b2, ptr.to(dev()), out)
    h64 = hid.double().cpu().requires_grad_(True)
    w64, bb64 = w2.double().cpu().requires_grad_(True), b2.double().cpu().requires_grad_(True)
    sd = {"head.MLP.0.weight": torch.eye(H, dtype=torch.float64), "head.MLP.0.bias": torch.zeros(H, dtype=torch.float64),
          "head.MLP.2.weight": w64, "head.MLP.2.bias": bb64}
    ref = orc.scalar_head(sd, h64, batch, Bg)
    assert rel_err(out, ref) < TOL
    dout = rnd(Bg, seed=5)
    (ref * dout.double().cpu()).sum().backward()
    dhid = torch.empty(N, H, device=dev())
    nparts = ops.node_nparts(N)
    parts = torch.zeros(nparts * (2 * H + 8), device=dev())
    ops.scalar_head_bwd(hid, w2, ptr.to(dev()), batch.to(dev()), dout, dhid, parts)
    assert rel_err(dhid, h64.grad) < TOL
    tot = parts.view(nparts, 2 * H + 8).double().sum(0)
    assert rel_err(tot[H + 8:], h64.grad.sum(0)) < TOL
    assert rel_err(tot[:H], w64.grad.view(-1)) < TOL
    assert rel_err(tot[H:H + 1], bb64.grad) < TOL


def test_adam_matches_torch(ops):
    n = 10007
    p0, g0 = rnd(n, seed=1), rnd(n, seed=2)
    p_ref = torch.nn.Parameter(p0.clone().cpu())
    opt = torch.optim.Adam([p_ref], lr=1e-3)
    p, m, v = p0.clone(), torch.zeros(n, device=dev()), torch.zeros(n, device=dev())
    for step in range(1, 4):
        gstep = g0 * step
        p_ref.grad = gstep.clone().cpu()
        opt.step()
        ops.adam_step(p, gstep, m, v, 1e-3, 0.9, 0.999, 1e-8, step, 1.0)
    assert rel_err(p, p_ref.data) < 1e-6


# ---- half storage (precision 2, operands / output kept in memory as bf16: csrc/gemm_h.h)
def _bf(t):
    return t.to(torch.bfloat16)


@pytest.mark.parametrize("M", [1, 127, 300, 33000])
def test_gemm_half_storage_activation_products(ops, M):
    """The forms the model chains at precision 2 with bf16 storage: fp32 A -> bf16 C with node-term gathers (layer GEMM 1),
    silu(bf16 A) -> fp32 / bf16 C with BatchNorm sums (GEMM 2), [bf16] A with silu'(bf16) -> fp32 / bf16 C (dpre), bf16 A ->
    fp32 C with a residual (dE).  References are built from the SAME bf16-rounded operands in fp64, so what is left is the
    bf16 rounding of the MFMA operands (weights, silu values) and of the stored output."""
    K, N, G = 256, 256, 2
    tol = 2e-2
    X = rnd(M, G * K, seed=3)
    Ws = [rnd(N, K, seed=20 + g, scale=0.1) for g in range(G)]
    Bt = [w.t().contiguous() for w in Ws]
    imgs = ops.split_b([w.t() for w in Ws])
    Xs = [X[:, g * K:(g + 1) * K] for g in range(G)]
    # (1) fp32 A -> bf16 C, gather epilogue
    nn = 50
    P = rnd(nn, 4 * N, seed=7)
    g_ = torch.Generator().manual_seed(3)
    tgt = torch.sort(torch.randint(0, nn, (M,), generator=g_)).values.to(torch.int32).to(dev())
    src = torch.randint(0, nn, (M,), generator=g_).to(torch.int32).to(dev())
    Ch = torch.full((M, G * N), float("nan"), device=dev(), dtype=torch.bfloat16)
    ops.gemm(Xs, Bt, [Ch[:, :N], Ch[:, N:]], b_kstrided=True, b_split=imgs, precision=2,
             gather_i=[P[:, :N], P[:, N:2 * N]], gather_j=[P[:, 2 * N:3 * N], P[:, 3 * N:]], tgt=tgt, src=src)
    Pd = P.double()
    for g in range(G):
        ref = Xs[g].double() @ Ws[g].double().t() + Pd[tgt.long(), g * N:(g + 1) * N] + Pd[src.long(), (2 + g) * N:(3 + g) * N]
        assert rel_err(Ch[:, g * N:(g + 1) * N].float(), ref) < tol, g
    # (2) silu(bf16 A) -> fp32 C and -> bf16 C, with BatchNorm sums of group 0
    Ah = _bf(X)
    Ahs = [Ah[:, g * K:(g + 1) * K] for g in range(G)]
    tiles = ops.gemm_tiles_m(M)
    for out_dtype in (torch.float32, torch.bfloat16):
        C2 = torch.full((M, G * N), float("nan"), device=dev(), dtype=out_dtype)
        cs = torch.zeros(tiles * N, dtype=torch.float64, device=dev())
        cq = torch.zeros(tiles * N, dtype=torch.float64, device=dev())
        ops.gemm(Ahs, Bt, [C2[:, :N], C2[:, N:]], b_kstrided=True, b_split=imgs, precision=2, a_act=True,
                 colsum=[cs, None], colsq=[cq, None])
        for g in range(G):
            ref = silu64(Ahs[g].double()) @ Ws[g].double().t()
            assert rel_err(C2[:, g * N:(g + 1) * N].float(), ref) < tol, (g, out_dtype)
        ref0 = silu64(Ahs[0].double()) @ Ws[0].double().t()
        assert rel_err(cs.view(tiles, N).sum(0), ref0.sum(0)) < tol          # sums are taken before the output is rounded
    # (3) dpre: [fp32 | bf16] A, silu'(bf16 pre) -> fp32 / bf16 C
    pre = _bf(rnd(M, G * N, seed=9))
    for a_half, c_half in ((False, False), (True, False), (True, True)):
        A3 = Ah if a_half else X
        A3s = [A3[:, g * K:(g + 1) * K] for g in range(G)]
        C3 = torch.full((M, G * N), float("nan"), device=dev(), dtype=torch.bfloat16 if c_half else torch.float32)
        ops.gemm(A3s, Bt, [C3[:, :N], C3[:, N:]], b_kstrided=True, b_split=imgs, precision=2,
                 dact=[pre[:, :N], pre[:, N:]])
        for g in range(G):
            ref = (A3s[g].double() @ Ws[g].double().t()) * dsilu64(pre[:, g * N:(g + 1) * N].double())
            assert rel_err(C3[:, g * N:(g + 1) * N].float(), ref) < tol, (g, a_half, c_half)
    # (4) dE: bf16 A -> fp32 C + residual
    resid = rnd(M, N, seed=12)
    C4 = torch.full((M, N), float("nan"), device=dev())
    ops.gemm(Ahs[0], Bt[0], C4, b_kstrided=True, b_split=imgs[:1], precision=2, resid=resid)
    assert rel_err(C4, Ahs[0].double() @ Ws[0].double().t() + resid.double()) < tol


@pytest.mark.parametrize("K", [16, 100, 4099, 40003])
def test_gemm_half_storage_weight_gradients(ops, K):
    """dW = dY^T (silu?)(X) with bf16 dY and / or bf16 X: whole and ragged K (the kernel masks the rows past K itself),
    one launch and split-K slabs summed by cartnet_splitk_reduce."""
    M, N, G = 256, 256, 2
    tol = 2e-2
    dY, Xm = rnd(K, G * M, seed=5), rnd(K, G * N, seed=6)
    for a_half, b_half, act in ((False, True, True), (True, True, True), (True, False, False), (True, True, False)):
        A = _bf(dY) if a_half else dY
        B = _bf(Xm) if b_half else Xm
        As = [A[:, g * M:(g + 1) * M] for g in range(G)]
        Bs = [B[:, g * N:(g + 1) * N] for g in range(G)]
        ref = [As[g].double().t() @ (silu64(Bs[g].double()) if act else Bs[g].double()) for g in range(G)]
        scale = max(r.abs().max().item() for r in ref)
        for S in (1, 4):
            if S > 1 and K < 32:
                continue
            outs = [torch.full((M, N), float("nan"), device=dev()) for _ in range(G)]
            if S == 1:
                ops.gemm(As, Bs, outs, a_kstrided=True, b_kstrided=True, b_act=act, precision=2)
            else:
                slabs = [torch.full((S * M, N), float("nan"), device=dev()) for _ in range(G)]
                ops.gemm(As, Bs, slabs, a_kstrided=True, b_kstrided=True, b_act=act, precision=2, splitk=S)
                ops.splitk_reduce(slabs, S, outs)
            for g in range(G):
                assert (outs[g].double().cpu() - ref[g].cpu()).abs().max().item() < tol * scale, (a_half, b_half, act, S, g)


def test_gemm_half_storage_is_refused_where_no_kernel_reads_it(ops):
    X, W = rnd(300, 256, seed=1), rnd(256, 256, seed=2)
    C_ = torch.empty(300, 256, device=dev())
    with pytest.raises(RuntimeError, match="half"):      # precision 0
        ops.gemm(_bf(X), W.t().contiguous(), C_, b_kstrided=True, b_split=ops.pack_b([W.t()]))
    with pytest.raises(RuntimeError, match="half"):      # no weight image
        ops.gemm(_bf(X), W.t().contiguous(), C_, b_kstrided=True, precision=2)


@pytest.mark.parametrize("N,nkeys", [(0, 5), (1, 1), (63, 3), (64, 119), (65, 119), (1023, 7), (12416, 119), (40000, 512)])
def test_sort_by_key_is_a_stable_counting_sort(ops, N, nkeys):
    """cartnet_sort_by_key (16 waves, ballot ranking): equal keys keep their input order at every size, incl. ranges
    that end inside a 64-item chunk, an empty input, a single key, and out-of-table keys (clamped and reported)."""
    g = torch.Generator().manual_seed(N + nkeys)
    z = torch.randint(0, nkeys, (N,), generator=g)
    if N > 3:
        z[::7] = z[0]                                  # long runs of one key across chunk and wave boundaries
    perm, zptr, status = ops.sort_by_key(z.to(dev()), nkeys)
    assert int(status.item()) == 0
    assert torch.equal(perm[:N].cpu().long(), torch.argsort(z, stable=True))
    assert torch.equal(zptr.cpu().long(), torch.cat([torch.zeros(1, dtype=torch.long),
                                                    torch.cumsum(torch.bincount(z, minlength=nkeys), 0)]))
    if N > 10:
        bad = z.clone()
        bad[3], bad[N - 2] = -1, nkeys + 4
        perm, zptr, status = ops.sort_by_key(bad.to(dev()), nkeys)
        assert int(status.item()) == 16
        assert torch.equal(perm[:N].cpu().long(), torch.argsort(bad.clamp(0, nkeys - 1), stable=True))


def test_colsum_finalize_f32_matches_fp64_column_sums(ops):
    for nparts, N in [(1, 5), (31, 33), (256, 904), (300, 1000)]:
        parts = rnd(nparts, N, seed=nparts)
        out = torch.full((N,), float("nan"), device=dev())
        ops.colsum_finalize(parts.contiguous().view(-1), nparts, out)
        assert rel_err(out, parts.double().sum(0)) < 1e-6


# ---- inference-mode fusion: second Linears + gate + per-target sums + edge residual in one kernel (csrc/gemm_f32gate.hip)
@pytest.mark.parametrize("D,degs", [(256, "random"), (512, "random"), (256, "long"), (256, "empty")])
def test_gate_gemm_eval_matches_the_two_kernel_path(ops, D, degs):
    """cartnet_gate_gemm_eval against fp64 and against the unfused path (cartnet_gemm + cartnet_gate_scatter_fwd with the
    same running statistics): ragged last tile, targets that straddle tile boundaries, a target spanning three tiles
    (> 256 edges), atoms without edges (zero rows), with and without the envelope."""
    g = torch.Generator().manual_seed(D + len(degs))
    if degs == "random":
        dl = torch.randint(0, 31, (700,), generator=g).tolist()
    elif degs == "long":
        dl = [3, 300, 0, 0, 129, 1, 127, 128, 2, 500, 7]
    else:
        dl = [0, 0, 5, 0, 0]
    ei, ptr = _graph_with_degrees(dl, seed=4)
    N, E = len(dl), ei.shape[1]
    lay = ops.GraphLayout(ei.to(dev()), N, ptr.to(dev()), need_csc=False)
    pre, e_in = rnd(E, 2 * D, seed=1), rnd(E, D, seed=2)
    W2g, W2a = rnd(D, D, seed=3, scale=0.08), rnd(D, D, seed=4, scale=0.08)
    bg, ba = rnd(D, seed=5), rnd(D, seed=6)
    mean = rnd(D, seed=7, scale=0.3)
    rstd = (1.0 / torch.sqrt(0.5 + torch.rand(D, generator=g))).to(dev())
    mr = torch.cat([mean, rstd]).contiguous()
    gamma, beta = rnd(D, seed=8) * 0.3 + 1.0, rnd(D, seed=9) * 0.2
    imgs = ops.pack_b([W2g.t(), W2a.t()])
    for env in (torch.rand(E, generator=g).to(dev()), None):
        e_out = torch.full((E, D), float("nan"), device=dev())
        aggr = torch.full((N, D), float("nan"), device=dev())
        ops.gate_gemm_eval(pre, imgs[0], imgs[1], bg, ba, mr, gamma, beta, env, e_in, lay, e_out, aggr)
        p64 = pre.double().cpu()
        h = silu64(p64)
        g64 = h[:, :D] @ W2g.double().cpu().t() + bg.double().cpu()
        s64 = h[:, D:] @ W2a.double().cpu().t() + ba.double().cpu()
        sig = torch.sigmoid((g64 - mean.double().cpu()) * rstd.double().cpu() * gamma.double().cpu() + beta.double().cpu())
        if env is not None:
            sig = sig * env.double().cpu()[:, None]
        ref_e = e_in.double().cpu() + sig
        ref_a = torch.zeros(N, D, dtype=torch.float64).index_add_(0, ei[1], sig * s64)
        if E > 0:
            assert rel_err(e_out, ref_e) < TOL
        assert torch.isfinite(aggr).all()
        assert rel_err(aggr, ref_a) < TOL
        zero_rows = torch.tensor([d == 0 for d in dl])
        assert bool((aggr.cpu()[zero_rows] == 0).all())
    # and run-to-run bitwise
    a2 = torch.empty_like(aggr)
    e2 = torch.empty_like(e_out)
    ops.gate_gemm_eval(pre, imgs[0], imgs[1], bg, ba, mr, gamma, beta, None, e_in, lay, e2, a2)
    assert torch.equal(a2, aggr) and (E == 0 or torch.equal(e2, e_out))


@pytest.mark.parametrize("precision", [0, 1])
@pytest.mark.parametrize("with_env", [True, False])
def test_gate_backward_sums_without_the_statistics_pass(ops, with_env, precision):
    """Round 5: sum(dbn) and sum(dbn ghat) of the gate's BatchNorm backward (models/cartnet.py:238) from three places
    where the operands are in registers anyway -- per-target sums of the forward gate kernel (bc), the node update's
    apply pass (daggr * bc) and the epilogue of the dE product that writes de_out (CartnetGemmArgs.gst_*) -- against the
    statistics pass they replace and against the fp64 formula.  D = 256, E >= 64 row tiles: the fused kernel's range."""
    D = 256
    ei, ptr = _random_graph_batch(30, 50, 80, 14, seed=77)
    N, E = int(ptr[-1]), ei.shape[1]
    assert (E + 127) // 128 >= 96                              # the bf16x3 kernel's range (64 row tiles at fp32)
    lay = ops.GraphLayout(ei.to(dev()), N, ptr.to(dev()))
    gs, e_in = rnd(E, 2 * D, seed=1), rnd(E, D, seed=2)
    env = torch.rand(E, generator=torch.Generator().manual_seed(3)).to(dev()) if with_env else None
    g64 = gs[:, :D].double().cpu()
    bmean, brstd = g64.mean(0), torch.rsqrt(g64.var(0, unbiased=False) + 1e-5)
    mean_rstd = torch.cat([bmean, brstd]).float().to(dev()).contiguous()
    gamma, beta = rnd(D, seed=6), rnd(D, seed=7)
    e_out, aggr, aggr2 = (torch.empty(E, D, device=dev()), torch.empty(N, D, device=dev()), torch.empty(N, D, device=dev()))
    e_out2 = torch.empty(E, D, device=dev())
    nparts = ops.gate_nparts(N)
    ps, pq, ps2, pq2 = (torch.zeros(nparts * D, device=dev(), dtype=torch.float64) for _ in range(4))
    bc = torch.full((N, 2 * D), 7.0, device=dev())
    ops.gate_scatter_fwd(gs, e_in, env, lay, mean_rstd, gamma, beta, e_out, aggr, ps, pq, bc=bc)
    ops.gate_scatter_fwd(gs, e_in, env, lay, mean_rstd, gamma, beta, e_out2, aggr2, ps2, pq2)
    assert torch.equal(e_out, e_out2) and torch.equal(aggr, aggr2) and torch.equal(ps, ps2)      # the old outputs, bit for bit
    # fp64 formula
    c = lambda t: t.detach().double().cpu()
    ev = c(env)[:, None] if with_env else 1.0
    ghat = (g64 - bmean) * brstd
    z = torch.sigmoid(ghat * c(gamma) + c(beta))
    w = ev * z * (1 - z)
    s64 = c(gs)[:, D:]
    tgt = ei[1]
    B_ref = torch.zeros(N, D, dtype=torch.float64).index_add_(0, tgt, s64 * w)
    C_ref = torch.zeros(N, D, dtype=torch.float64).index_add_(0, tgt, s64 * w * ghat)
    assert rel_err(bc[:, :D], B_ref) < TOL and rel_err(bc[:, D:], C_ref) < TOL
    # node share: daggr comes out of the node update's apply pass, which also leaves sum_t daggr[t] bc[t]
    aggr_in, dx_out = rnd(N, D, seed=11), rnd(N, D, seed=12)
    mr2 = torch.cat([aggr_in.mean(0), torch.rsqrt(aggr_in.var(0, unbiased=False) + 1e-5)]).contiguous()
    g2, b2 = rnd(D, seed=13), rnd(D, seed=14)
    npn = ops.node_nparts(N)
    pa, pb = (torch.zeros(npn * D, device=dev(), dtype=torch.float64) for _ in range(2))
    ops.node_update_bwd_stats(aggr_in, dx_out, mr2, g2, b2, pa, pb)
    sums2 = torch.empty(2 * D, device=dev())
    ops.colsum_finalize([pa, pb], npn, [sums2[:D], sums2[D:]])
    daggr, daggr_plain = torch.empty(N, D, device=dev()), torch.empty(N, D, device=dev())
    na, nb = (torch.full((npn * D,), 5.0, device=dev(), dtype=torch.float64) for _ in range(2))
    ops.node_update_bwd_apply(aggr_in, dx_out, mr2, g2, b2, sums2, True, daggr, bc=bc, parts_a=na, parts_b=nb)
    ops.node_update_bwd_apply(aggr_in, dx_out, mr2, g2, b2, sums2, True, daggr_plain)
    assert torch.equal(daggr, daggr_plain)
    node_a, node_b = na.view(npn, D).sum(0).cpu(), nb.view(npn, D).sum(0).cpu()
    assert rel_err(node_a, (c(daggr) * c(bc[:, :D])).sum(0)) < 1e-6
    assert rel_err(node_b, (c(daggr) * c(bc[:, D:])).sum(0)) < 1e-6
    # edge share: the dE product de_out = resid + dpre @ [W1; W2] (two folded K-segments of 256) with the epilogue
    dpre, resid = rnd(E, 2 * D, seed=21, scale=0.3), rnd(E, D, seed=22)
    W = rnd(2 * D, D, seed=23, scale=0.05)                    # [K = 512, N = 256]: segments W[:256], W[256:]
    img = (ops.pack_b if precision == 0 else ops.split_b)([W[:D], W[D:]])
    folded = torch.cat([t.view(-1) for t in img]).contiguous()
    de_out = torch.empty(E, D, device=dev())
    tiles = ops.gemm_tiles_m(E)
    ca, cb = (torch.full((tiles * D,), 3.0, device=dev(), dtype=torch.float64) for _ in range(2))
    ops.gemm([dpre[:, :D], dpre[:, D:]], [W[:D], W[D:]], de_out, segments=True, b_kstrided=True, resid=resid,
             b_split=img, b_split_folded=folded, colsum=ca, colsq=cb, precision=precision,
             gate_stats=(gs[:, :D], env, mean_rstd, gamma, beta))
    de_ref = c(resid) + c(dpre) @ c(W)
    assert rel_err(de_out, de_ref) < TOL
    plain = torch.empty(E, D, device=dev())
    ops.gemm([dpre[:, :D], dpre[:, D:]], [W[:D], W[D:]], plain, segments=True, b_kstrided=True, resid=resid, b_split=img,
             b_split_folded=folded, precision=precision)
    assert torch.equal(plain, de_out)                         # the statistics do not touch the product
    # the last layer's form: no edge residual (the head does not read the edge features)
    nores = torch.empty(E, D, device=dev())
    ca2, cb2 = (torch.full((tiles * D,), 3.0, device=dev(), dtype=torch.float64) for _ in range(2))
    ops.gemm([dpre[:, :D], dpre[:, D:]], [W[:D], W[D:]], nores, segments=True, b_kstrided=True, b_split=img,
             b_split_folded=folded, colsum=ca2, colsq=cb2, precision=precision,
             gate_stats=(gs[:, :D], env, mean_rstd, gamma, beta))
    assert rel_err(nores, c(dpre) @ c(W)) < TOL
    assert rel_err(ca2.view(tiles, D).sum(0).cpu(), (c(nores) * w).sum(0)) < 2e-6
    assert rel_err(cb2.view(tiles, D).sum(0).cpu(), (c(nores) * w * ghat).sum(0)) < 2e-6
    edge_a, edge_b = ca.view(tiles, D).sum(0).cpu(), cb.view(tiles, D).sum(0).cpu()
    assert rel_err(edge_a, (c(de_out) * w).sum(0)) < 2e-6
    assert rel_err(edge_b, (c(de_out) * w * ghat).sum(0)) < 2e-6
    # together == the statistics pass == the formula
    spa, spb = (torch.zeros(nparts * D, device=dev(), dtype=torch.float64) for _ in range(2))
    ops.gate_scatter_bwd_stats(gs, de_out, daggr, env, lay, mean_rstd, gamma, beta, spa, spb)
    old_a, old_b = spa.view(nparts, D).sum(0).cpu(), spb.view(nparts, D).sum(0).cpu()
    dbn = (c(daggr)[tgt] * s64 + c(de_out)) * w
    ref_a, ref_b = dbn.sum(0), (dbn * ghat).sum(0)
    scale_a, scale_b = dbn.abs().sum(0).max().item(), (dbn * ghat).abs().sum(0).max().item()
    for got_a, got_b in ((edge_a + node_a, edge_b + node_b), (old_a, old_b)):
        assert (got_a - ref_a).abs().max().item() <= 2e-6 * scale_a
        assert (got_b - ref_b).abs().max().item() <= 2e-6 * scale_b
    # a launch that cannot reach the kernel is refused, not silently computed without the sums
    small = 50 * 128
    with pytest.raises(ValueError, match="gate-statistics"):
        ops.gemm([dpre[:small, :D], dpre[:small, D:]], [W[:D], W[D:]], de_out[:small], segments=True, b_kstrided=True,
                 resid=resid[:small], b_split=img, b_split_folded=folded, colsum=ca[:50 * D], colsq=cb[:50 * D],
                 precision=precision, gate_stats=(gs[:small, :D], None, mean_rstd, gamma, beta))
