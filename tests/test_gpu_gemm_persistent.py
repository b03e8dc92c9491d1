"""The persistent fp32 activation x weight kernel (csrc/gemm_f32p.h; CartnetGemmArgs.tile_policy = 3 forces it, 256 excludes
it) through the C ABI: every compiled epilogue form against an fp64 torch evaluation of the same arithmetic
(max|delta| <= 1e-5 * max|ref|, north_star's fp32 bar) AND against the shipped second-generation kernels, which the forms
without a transcendental in the epilogue must equal bit for bit (same MFMA chain, same order of every sum).

Shapes exercise what the kernel's structure can get wrong: workgroups without a tile / with one / with an odd and an even
number of tiles (the two accumulator sets, the drain behind either loop exit), a ragged last row tile (rows dropped by the
buffer descriptors' range check, statistics masked), two and four groups and two column tiles (the workgroup -> (group,
column tile) map), column-block views (row strides wider than N)."""
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-5
D = 256


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


# M: 1 row; one full tile; 2 tiles + ragged; 257 row tiles (every workgroup of a group pair gets one, some two);
# 700 tiles + ragged (2-3 per workgroup: both loop exits)
SHAPES = [1, 128, 300, 128 * 257, 89500]


def _run(ops, pol, M, groups, N, form, seed=0):
    """One launch of `form` with tile_policy `pol`; returns the outputs and the fp64 references."""
    G = groups
    A = rnd(M, G * D, seed=seed + 1)                                   # groups read column blocks of one matrix
    W = [rnd(D, N, seed=seed + 10 + g, scale=0.06) for g in range(G)]  # [K, N]: b_kstrided
    img = ops.pack_b(W)
    out = torch.full((M, G * N), float("nan"), device=dev())
    Av = [A[:, g * D:(g + 1) * D] for g in range(G)]
    Cv = [out[:, g * N:(g + 1) * N] for g in range(G)]
    bias = [rnd(N, seed=seed + 20 + g) for g in range(G)]
    tiles = ops.gemm_tiles_m(M)
    kw = dict(b_kstrided=True, b_split=img, tile_policy=pol)
    res = {"out": out}
    ref = {}
    A64 = [a.double() for a in Av]
    W64 = [w.double() for w in W]
    if form == "plain":
        ops.gemm(Av, W, Cv, **kw)
        ref["out"] = torch.cat([A64[g] @ W64[g] for g in range(G)], 1)
    elif form == "bias":
        ops.gemm(Av, W, Cv, bias=bias, **kw)
        ref["out"] = torch.cat([A64[g] @ W64[g] + bias[g].double() for g in range(G)], 1)
    elif form in ("act", "stats", "stats_actout"):
        cs = [torch.full((tiles * N,), float("nan"), dtype=torch.float64, device=dev()) for _ in range(G)]
        cq = [torch.full((tiles * N,), float("nan"), dtype=torch.float64, device=dev()) for _ in range(G)]
        act = torch.full((M, G * D), float("nan"), device=dev())
        extra = {}
        if form != "act":
            # (the model takes the statistics of the first group only: a missing statistic must simply be dropped)
            extra.update(colsum=[cs[0]] + [None] * (G - 1), colsq=[cq[0]] + [None] * (G - 1))
            res.update(cs=cs[0], cq=cq[0])
        if form == "stats_actout":
            extra.update(a_act_out=[act[:, g * D:(g + 1) * D] for g in range(G)])
            res["act"] = act
        ops.gemm(Av, W, Cv, a_act=True, bias=bias, **extra, **kw)
        v = [silu64(A64[g]) @ W64[g] + bias[g].double() for g in range(G)]
        ref["out"] = torch.cat(v, 1)
        if form != "act":
            pad = tiles * 128 - M
            v0 = torch.cat([v[0], torch.zeros(pad, N, dtype=torch.float64, device=dev())]).view(tiles, 128, N)
            ref["cs"] = v0.sum(1).reshape(-1)
            ref["cq"] = (v0 * v0).sum(1).reshape(-1)
        if form == "stats_actout":
            ref["act"] = silu64(A.double())
    elif form in ("dpre", "dpre_colsum"):
        pre = rnd(M, G * N, seed=seed + 30)
        Pv = [pre[:, g * N:(g + 1) * N] for g in range(G)]
        extra = {}
        if form == "dpre_colsum":
            cs = [torch.full((tiles * N,), float("nan"), dtype=torch.float64, device=dev()) for _ in range(G)]
            extra["colsum"] = cs
        ops.gemm(Av, W, Cv, dact=Pv, **extra, **kw)
        if form == "dpre_colsum":
            res["cs_all"] = torch.stack(cs)
        v = [(A64[g] @ W64[g]) * dsilu64(Pv[g].double()) for g in range(G)]
        ref["out"] = torch.cat(v, 1)
        if form == "dpre_colsum":
            pad = tiles * 128 - M
            ref["cs_all"] = torch.stack([torch.cat([x, torch.zeros(pad, N, dtype=torch.float64, device=dev())])
                                         .view(tiles, 128, N).sum(1).reshape(-1) for x in v])
    elif form == "gather":
        # the layer's first product: + node terms of the edge's target and source atoms (tables of `atoms` rows, column
        # blocks of one [atoms, 2 G N] matrix as in the model; every atom index occurs, also 0 and atoms - 1)
        atoms = max(2, M // 14)
        P = rnd(atoms, 2 * G * N, seed=seed + 40)
        gen = torch.Generator().manual_seed(seed + 41)
        tgt = torch.sort(torch.randint(0, atoms, (M,), generator=gen)).values.to(torch.int32).to(dev())
        src = torch.randint(0, atoms, (M,), generator=gen).to(torch.int32).to(dev())
        if M > 1:
            tgt[0], tgt[-1], src[0], src[-1] = 0, atoms - 1, atoms - 1, 0
        gi = [P[:, g * N:(g + 1) * N] for g in range(G)]
        gj = [P[:, (G + g) * N:(G + g + 1) * N] for g in range(G)]
        ops.gemm(Av, W, Cv, bias=bias, gather_i=gi, gather_j=gj, tgt=tgt, src=src, **kw)
        ref["out"] = torch.cat([A64[g] @ W64[g] + bias[g].double() + gi[g].double()[tgt.long()] + gj[g].double()[src.long()]
                                for g in range(G)], 1)
    else:
        raise AssertionError(form)
    torch.cuda.synchronize()
    return res, ref


@pytest.mark.parametrize("form", ["plain", "bias", "act", "stats", "stats_actout", "dpre", "dpre_colsum", "gather"])
@pytest.mark.parametrize("M", SHAPES)
def test_every_form_against_fp64_and_the_shipped_kernels(ops, form, M):
    new, ref = _run(ops, 3, M, 2, D, form)
    old, _ = _run(ops, 256, M, 2, D, form)
    for k in ref:
        assert not torch.isnan(new[k]).any(), f"{form} M={M}: {k} has elements the kernel never wrote"
        # column statistics are sums of up to 128 values: the bar is relative to the largest sum
        assert rel_err(new[k], ref[k]) < (TOL if k in ("out", "act") else 2e-5), (form, M, k)
    bitwise = form in ("plain", "bias", "act", "stats", "stats_actout", "gather")
    for k in ref:
        if bitwise:
            assert torch.equal(new[k], old[k]), f"{form} M={M}: {k} differs from the shipped kernel"
        else:          # hardware exp / rcp in the epilogue, fused differently: one rounding
            assert rel_err(new[k], old[k]) < 2e-6, (form, M, k)


@pytest.mark.parametrize("groups,N", [(1, 512), (4, 256), (1, 1024), (2, 512), (1, 256)])
def test_groups_and_column_tiles(ops, groups, N):
    M = 128 * 70 + 33
    for form in ("bias", "stats"):
        new, ref = _run(ops, 3, M, groups, N, form, seed=3)
        old, _ = _run(ops, 256, M, groups, N, form, seed=3)
        for k in ref:
            assert not torch.isnan(new[k]).any()
            assert rel_err(new[k], ref[k]) < (TOL if k == "out" else 2e-5), (groups, N, form, k)
            assert torch.equal(new[k], old[k]), (groups, N, form, k)


def test_is_bitwise_repeatable_and_leaves_the_rest_of_the_buffers_alone(ops):
    """Column-block views: the bytes between a group's N columns and the row stride must not be touched (every store is a
    dword inside the view), and two launches give the same bits."""
    M, N, G = 128 * 300 + 77, D, 2
    A = rnd(M, G * D + 64, seed=5)
    W = [rnd(D, N, seed=6 + g, scale=0.06) for g in range(G)]
    img = ops.pack_b(W)
    outs = []
    for rep in range(2):
        out = torch.full((M, G * N + 32), 7.25, device=dev())
        ops.gemm([A[:, g * D:(g + 1) * D] for g in range(G)], W, [out[:, g * N:(g + 1) * N] for g in range(G)],
                 b_kstrided=True, b_split=img, tile_policy=3)
        torch.cuda.synchronize()
        assert torch.all(out[:, G * N:] == 7.25)
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    ref = torch.cat([A[:, g * D:(g + 1) * D].double() @ W[g].double() for g in range(G)], 1)
    assert rel_err(outs[0][:, :G * N], ref) < TOL


def test_the_library_picks_it_for_the_model_sized_launches_only(ops):
    """tile_policy 0: the edge-sized two-group layer products take the persistent kernel (same bits as policy 3), small
    launches keep the 2,768-workgroup kernels (same bits as policy 256) -- and either way the results agree."""
    for M in (300, 177140):
        a, _ = _run(ops, 0, M, 2, D, "stats", seed=9)
        b, _ = _run(ops, 3, M, 2, D, "stats", seed=9)
        c, _ = _run(ops, 256, M, 2, D, "stats", seed=9)
        for k in ("out", "cs", "cq"):
            assert torch.equal(a[k], b[k]) and torch.equal(a[k], c[k])


@pytest.mark.parametrize("M", [300, 128 * 257, 89500])
def test_forms_of_the_second_unit(ops, M):
    """csrc/gemm_f32p2.hip: silu(A) written without statistics, the softplus family with the pre-activation kept (iComformer's
    RBF branches), and the two K = 512 forms (32 K-steps per tile, slices on the first 16): all bit for bit the shipped
    kernels' results, and 1e-5 of an fp64 evaluation."""
    import torch.nn.functional as F
    A = rnd(M, 2 * D, seed=41)
    W = [rnd(D, D, seed=42 + g, scale=0.06) for g in range(2)]
    bias = [rnd(D, seed=44 + g) for g in range(2)]
    img = ops.pack_b(W)
    Av = [A[:, :D], A[:, D:]]
    W512 = rnd(2 * D, D, seed=46, scale=0.05)
    img512 = ops.pack_b([W512])
    Wseg = [rnd(D, D, seed=47 + g, scale=0.06) for g in range(2)]
    img_fold = torch.cat(ops.pack_b(Wseg))
    resid = rnd(M, D, seed=49)
    res = {}
    for pol in (3, 256):
        o = {k: torch.full((M, 2 * D), float("nan"), device=dev()) for k in ("c1", "h1", "c2", "p2", "c3", "h3", "c4")}
        ops.gemm(Av, W, [o["c1"][:, :D], o["c1"][:, D:]], b_kstrided=True, b_split=img, a_act=True, bias=bias,
                 a_act_out=[o["h1"][:, :D], o["h1"][:, D:]], tile_policy=pol)
        ops.gemm(Av, W, [o["c2"][:, :D], o["c2"][:, D:]], b_kstrided=True, b_split=img, bias=bias,
                 cpre=[o["p2"][:, :D], o["p2"][:, D:]], out_act=True, dact_kind=1, tile_policy=pol)
        ops.gemm(A, W512, o["c3"][:, :D], b_kstrided=True, b_split=img512, a_act=True, out_act=True, bias=bias[0],
                 cpre=o["c3"][:, D:], a_act_out=o["h3"], tile_policy=pol)
        ops.gemm(Av, Wseg, o["c4"][:, :D], b_kstrided=True, segments=True, resid=resid, b_split_folded=img_fold,
                 tile_policy=pol)
        torch.cuda.synchronize()
        res[pol] = o
    for k in res[3]:
        a, b = res[3][k], res[256][k]
        assert torch.equal(torch.isnan(a), torch.isnan(b)), k
        assert torch.equal(a.nan_to_num(nan=0.5), b.nan_to_num(nan=0.5)), f"M={M}: {k} differs from the shipped kernel"
    A64 = A.double()
    v1 = torch.cat([silu64(A64[:, g * D:(g + 1) * D]) @ W[g].double() + bias[g].double() for g in range(2)], 1)
    assert rel_err(res[3]["c1"], v1) < TOL and rel_err(res[3]["h1"], silu64(A64)) < TOL
    v2 = torch.cat([A64[:, g * D:(g + 1) * D] @ W[g].double() + bias[g].double() for g in range(2)], 1)
    assert rel_err(res[3]["p2"], v2) < TOL and rel_err(res[3]["c2"], F.softplus(v2)) < TOL
    v3 = silu64(A64) @ W512.double() + bias[0].double()
    assert rel_err(res[3]["c3"][:, D:], v3) < TOL and rel_err(res[3]["c3"][:, :D], silu64(v3)) < TOL
    assert rel_err(res[3]["h3"], silu64(A64)) < TOL
    v4 = A64[:, :D] @ Wseg[0].double() + A64[:, D:] @ Wseg[1].double() + resid.double()
    assert rel_err(res[3]["c4"][:, :D], v4) < TOL


@pytest.mark.parametrize("M", [300, 128 * 257, 89500])
def test_forms_of_the_third_unit(ops, M):
    """csrc/gemm_f32p3.hip (single-group products of the iComformer step): K = 512 from two folded segments without an
    epilogue operand, fp32 column sums (a bias gradient), BatchNorm statistics without SiLU on A, and the K = 512 product
    times sigmoid(pre) (softplus') with its bias gradient."""
    A = rnd(M, 2 * D, seed=51)
    W = rnd(D, D, seed=52, scale=0.06)
    img = ops.pack_b([W])
    Wseg = [rnd(D, D, seed=53 + g, scale=0.06) for g in range(2)]
    img_fold = torch.cat(ops.pack_b(Wseg))
    pre = rnd(M, D, seed=55)
    resid2 = rnd(M, D, seed=60)
    A3 = rnd(M, 3 * D, seed=56)
    Wseg3 = [rnd(D, D, seed=57 + g, scale=0.05) for g in range(3)]
    img_fold3 = torch.cat(ops.pack_b(Wseg3))
    tiles = ops.gemm_tiles_m(M)
    Av = [A[:, :D], A[:, D:]]
    res = {}
    for pol in (3, 256):
        o = {k: torch.full((M, D), float("nan"), device=dev()) for k in ("c1", "c2", "c3", "c4")}
        s = {k: torch.full((tiles * D,), float("nan"), dtype=torch.float64, device=dev()) for k in ("s2", "s3", "q3", "s4")}
        ops.gemm(Av, Wseg, o["c1"], b_kstrided=True, segments=True, b_split_folded=img_fold, tile_policy=pol)
        ops.gemm(A[:, :D], W, o["c2"], b_kstrided=True, b_split=img, colsum=s["s2"], tile_policy=pol)
        ops.gemm(A[:, :D], W, o["c3"], b_kstrided=True, b_split=img, colsum=s["s3"], colsq=s["q3"], tile_policy=pol)
        ops.gemm(Av, Wseg, o["c4"], b_kstrided=True, segments=True, b_split_folded=img_fold, dact=pre, dact_kind=1,
                 colsum=s["s4"], tile_policy=pol)
        torch.cuda.synchronize()
        res[pol] = {**o, **s}
        # K = 768: three folded segments + residual
        o["c5"] = torch.full((M, D), float("nan"), device=dev())
        ops.gemm([A3[:, g * D:(g + 1) * D] for g in range(3)], Wseg3, o["c5"], b_kstrided=True, segments=True, resid=pre,
                 b_split_folded=img_fold3, tile_policy=pol)
        # K = 512 + residual, times the activation's derivative (both families), bias gradient: two epilogue operands
        for name, kind in (("6", 0), ("7", 1)):
            o["c" + name] = torch.full((M, D), float("nan"), device=dev())
            o["s" + name] = torch.full((tiles * D,), float("nan"), dtype=torch.float64, device=dev())
            ops.gemm(Av, Wseg, o["c" + name], b_kstrided=True, segments=True, b_split_folded=img_fold, resid=resid2, dact=pre,
                     dact_kind=kind, colsum=o["s" + name], tile_policy=pol)
        torch.cuda.synchronize()
        res[pol].update({k: o[k] for k in ("c5", "c6", "s6", "c7", "s7")})
    v5 = sum(A3.double()[:, g * D:(g + 1) * D] @ Wseg3[g].double() for g in range(3)) + pre.double()
    assert rel_err(res[3]["c5"], v5) < TOL
    for k in ("c1", "c2", "c3", "s3", "q3", "c5"):     # no transcendental in the epilogue: the shipped kernels' bits
        assert torch.equal(res[3][k], res[256][k]), f"M={M}: {k} differs from the shipped kernel"
    A64 = A.double()
    pad = tiles * 128 - M

    def tile_sums(v):
        return torch.cat([v, torch.zeros(pad, D, dtype=torch.float64, device=dev())]).view(tiles, 128, D).sum(1).reshape(-1)

    v1 = A64[:, :D] @ Wseg[0].double() + A64[:, D:] @ Wseg[1].double()
    v2 = A64[:, :D] @ W.double()
    v4 = v1 * torch.sigmoid(pre.double())
    r = res[3]
    assert not any(torch.isnan(t).any() for t in r.values())
    assert rel_err(r["c1"], v1) < TOL and rel_err(r["c2"], v2) < TOL and rel_err(r["c3"], v2) < TOL and rel_err(r["c4"], v4) < TOL
    assert rel_err(r["s2"], tile_sums(v2)) < 2e-5 and rel_err(r["s3"], tile_sums(v2)) < 2e-5
    assert rel_err(r["q3"], tile_sums(v2 * v2)) < 2e-5 and rel_err(r["s4"], tile_sums(v4)) < 2e-5
    assert rel_err(r["s2"], res[256]["s2"]) < 2e-6 and rel_err(r["c4"], res[256]["c4"]) < 2e-6
    s_ = torch.sigmoid(pre.double())
    v6 = (v1 + resid2.double()) * s_ * (1 + pre.double() * (1 - s_))
    v7 = (v1 + resid2.double()) * s_
    for k, v in (("6", v6), ("7", v7)):
        assert rel_err(r["c" + k], v) < TOL and rel_err(r["s" + k], tile_sums(v)) < 2e-5
        assert rel_err(r["c" + k], res[256]["c" + k]) < 2e-6 and rel_err(r["s" + k], res[256]["s" + k]) < 2e-6


@pytest.mark.parametrize("M", [128 * 257 + 5, 89500])
@pytest.mark.parametrize("with_resid", [True, False])
def test_gate_statistics_form(ops, M, with_resid):
    """CartNet's dE product with the gate statistics of the layer below in its epilogue (CartnetGemmArgs.gst_*; KIND 530 of
    gemm_f32p.h: K = 512 from two folded segments, residual optional, envelope required): de_out bit for bit the
    second-generation kernel's, the two column sums within 5e-6 of an fp64 evaluation (and of the other kernel's); a ragged last
    row tile -- its missing rows have envelope 0 by the descriptor's range check -- and workgroups with 0 / 1 / 2 / 3 tiles."""
    dpre = rnd(M, 2 * D, seed=71, scale=0.3)
    W = rnd(2 * D, D, seed=72, scale=0.05)
    resid = rnd(M, D, seed=73) if with_resid else None
    gs = rnd(M, 2 * D, seed=74)
    env = torch.rand(M, generator=torch.Generator().manual_seed(75)).to(dev())
    g64 = gs[:, :D].double()
    mean, rstd = g64.mean(0), torch.rsqrt(g64.var(0, unbiased=False) + 1e-5)
    mean_rstd = torch.cat([mean, rstd]).float().contiguous()
    gamma, beta = rnd(D, seed=76), rnd(D, seed=77)
    img = ops.pack_b([W[:D], W[D:]])
    folded = torch.cat([t.view(-1) for t in img]).contiguous()
    tiles = ops.gemm_tiles_m(M)
    got = {}
    for pol in (3, 128):
        o = torch.full((M, D), float("nan"), device=dev())
        ca, cb = (torch.full((tiles * D,), float("nan"), device=dev(), dtype=torch.float64) for _ in range(2))
        ops.gemm([dpre[:, :D], dpre[:, D:]], [W[:D], W[D:]], o, segments=True, b_kstrided=True, resid=resid, b_split=img,
                 b_split_folded=folded, colsum=ca, colsq=cb, tile_policy=pol, gate_stats=(gs[:, :D], env, mean_rstd, gamma, beta))
        torch.cuda.synchronize()
        assert not torch.isnan(o).any() and not torch.isnan(ca).any() and not torch.isnan(cb).any()
        got[pol] = (o, ca.view(tiles, D).sum(0), cb.view(tiles, D).sum(0))
    assert torch.equal(got[3][0], got[128][0])
    v = got[3][0].double()
    ghat = (g64 - mean_rstd[:D].double()) * mean_rstd[D:].double()
    z = torch.sigmoid(ghat * gamma.double() + beta.double())
    w = env.double()[:, None] * z * (1 - z)
    ref = dpre.double() @ W.double() + (resid.double() if with_resid else 0)
    assert rel_err(got[3][0], ref) < TOL
    for k, r in ((1, (v * w).sum(0)), (2, (v * w * ghat).sum(0))):
        scale = (v * w).abs().sum(0).max() if k == 1 else (v * w * ghat).abs().sum(0).max()
        assert (got[3][k] - r).abs().max() <= 5e-6 * scale and (got[3][k] - got[128][k]).abs().max() <= 5e-6 * scale
