"""Accuracy evidence beyond well-scaled N(0,1) operands, through the C ABI.

(a) The GEMM kernels at precision 0 (fp32 MFMA) and 1 (bf16x3: every fp32 product rebuilt from six bf16 MFMA products)
    on operands that stress a split-operand scheme: rows / columns whose scales span 2^-40 .. 2^+20, K-sums that cancel
    (O(1) terms, sum ~ 0) and operands whose LOW bf16 pieces sit next to the denormal range.  The error measure is the
    componentwise one every fp32 dot product obeys,

        err_ij = |C_ij - ref_ij| / (sum_k |a_ik| |b_kj|)        in units of u = 2^-24,

    (ref in fp64), which does not care about scaling or cancellation; a correct fp32 GEMM gives O(sqrt(K)) .. O(K) u.
    Asserted: both precisions stay under BOUND_U and bf16x3 is never worse than 2x the fp32 MFMA on the same operands.
(b) fast_sigmoid / fast_silu / fast_dsilu (v_exp_f32 + bare v_rcp_f32, csrc/gemm_kernel.h:39) and the gate kernels'
    cn_sigmoid at edge values -- +-88 (exp overflow threshold), +-1e4, +-inf, NaN -- against torch's own SiLU / sigmoid.
The MSE-loss training step against the oracle's autograd (reference train/train.py:173-178) is in test_gpu_model.py.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

U = 2.0 ** -24
BOUND_U = 48.0          # componentwise error bound asserted for K <= 512, in units of u (measured: 2 .. 12)


@pytest.fixture(scope="module")
def ops():
    from cartnet_amd import ops as _ops
    from cartnet_amd import lib
    lib.load()
    return _ops


def dev():
    return torch.device("cuda:0")


def rnd(*shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g)


def comp_err_u(C, A64, B64, ref64):
    """max_ij |C - ref| / (|A| |B|)_ij in units of u; A64 [M,K], B64 [K,N] on the CPU in fp64."""
    bound = A64.abs() @ B64.abs()
    err = (C.detach().double().cpu() - ref64).abs()
    ok = bound > 0
    return (err[ok] / bound[ok]).max().item() / U


def pow2(exps):
    return torch.pow(torch.tensor(2.0, dtype=torch.float64), exps.double()).float()


def operands(case, M, K, N, seed):
    """A [M,K], B [K,N] fp32 on the CPU for one stress case."""
    A, B = rnd(M, K, seed=seed), rnd(K, N, seed=seed + 1) * 0.1
    g = torch.Generator().manual_seed(seed + 2)
    if case == "row_col_scales":          # every row of A and every column of B at its own power of two, 2^-40 .. 2^+20
        A = A * pow2(torch.randint(-40, 21, (M, 1), generator=g))
        B = B * pow2(torch.randint(-40, 21, (1, N), generator=g))
    elif case == "k_scales":              # scales vary ALONG the reduction: terms of a dot product span 2^-30 .. 2^+10
        s = torch.randint(-30, 11, (1, K), generator=g)
        A = A * pow2(s)
        B = B * pow2(-s.t() + torch.randint(-3, 4, (K, 1), generator=g))
    elif case == "cancellation":          # second half of K repeats the first with the sign flipped (+ a 2^-20 residue)
        h = K // 2
        A = torch.cat([A[:, :h], -A[:, :h] + rnd(M, h, seed=seed + 3) * 2.0 ** -20], dim=1)
        B = torch.cat([B[:h], B[:h]], dim=0)
    elif case == "low_pieces_near_denormal":   # |a| ~ 2^-100: the third bf16 piece (2^-16 a) ~ 2^-116 .. 2^-122, normal but close
        A = A * 2.0 ** -100
        B = B * 2.0 ** 60                 # products ~ 2^-40: comfortably normal in the fp32 accumulator
    elif case == "mixed_magnitude_rows":  # a few huge entries per row next to tiny ones (outlier channels)
        mask = torch.rand(M, K, generator=g) < 0.02
        A = torch.where(mask, A * 2.0 ** 18, A * 2.0 ** -12)
    else:
        raise ValueError(case)
    return A.contiguous(), B.contiguous()


CASES = ["row_col_scales", "k_scales", "cancellation", "low_pieces_near_denormal", "mixed_magnitude_rows"]


def run_nn_image(ops, A, B, precision):
    """C = A @ B through the DMA-fed activation x weight kernels (weights as a pre-arranged image)."""
    Ad, Bd = A.to(dev()), B.to(dev())
    img = (ops.pack_b if precision == 0 else ops.split_b)([Bd])
    C = torch.full((A.shape[0], B.shape[1]), float("nan"), device=dev())
    ops.gemm(Ad, Bd, C, b_kstrided=True, precision=precision, b_split=img)
    return C


def run_nt_general(ops, A, B, precision):
    """C = A @ B through the general kernel (B handed over as [N, K] rows, no image)."""
    Ad, Bt = A.to(dev()), B.t().contiguous().to(dev())
    C = torch.full((A.shape[0], B.shape[1]), float("nan"), device=dev())
    ops.gemm(Ad, Bt, C, precision=precision)
    return C


def run_tn(ops, A, B, precision):
    """C = A @ B as a weight gradient: the reduction runs over the ROWS of two k-strided operands (A^T [K,M], B [K,N])."""
    At, Bd = A.t().contiguous().to(dev()), B.to(dev())
    C = torch.full((A.shape[0], B.shape[1]), float("nan"), device=dev())
    ops.gemm(At, Bd, C, a_kstrided=True, b_kstrided=True, precision=precision)
    return C


@pytest.mark.parametrize("case", CASES)
# Shapes: at precision 1 a launch with fewer than 96 tiles of 128 x 256 runs on the narrow-tile fp32 kernels by design
# (csrc/gemm.hip choose_bn), so the bf16x3 kernels only see M >= 12,288 rows (activation x weight products) or the
# 256-wide weight-gradient form; the small shapes at the end pin that fallback (both precisions then agree bit for bit).
@pytest.mark.parametrize("form,M,K,N", [("nn_image", 33000, 256, 256), ("nn_image", 13000, 512, 256),
                                        ("nt_general", 13000, 208, 256), ("tn", 256, 6000, 256), ("tn", 256, 20001, 256),
                                        ("nn_image", 1000, 512, 256), ("tn", 128, 777, 64)])
def test_gemm_stress_operands_both_precisions(ops, case, form, M, K, N):
    run = {"nn_image": run_nn_image, "nt_general": run_nt_general, "tn": run_tn}[form]
    A, B = operands(case, M, K, N, seed=100)
    A64, B64 = A.double(), B.double()
    ref = A64 @ B64
    e = {p: comp_err_u(run(ops, A, B, p), A64, B64, ref) for p in (0, 1)}
    # the error of a K-term fp32 dot product grows at most like K u (typically sqrt(K) u); both precisions are far below
    bound = BOUND_U * max(1.0, math.sqrt(K / 512.0))
    print(f"ACCURACY {case:26s} {form:10s} M={M:6d} K={K:5d} N={N:4d}  fp32 MFMA {e[0]:7.2f} u   bf16x3 {e[1]:7.2f} u")
    assert e[0] < bound, (case, form, e)
    assert e[1] < bound, (case, form, e)
    assert e[1] <= 2.0 * e[0] + 1.0, (case, form, e)      # bf16x3 never worse than 2x the fp32 MFMA (+1 u of slack)


def test_gemm_denormal_low_pieces_are_a_documented_limit(ops):
    """|a| ~ 2^-112: the third bf16 piece of an operand falls into the denormal range.  fp32 MFMA keeps its accuracy;
    bf16x3 may lose the piece (2^-16 relative to the product instead of 2^-24) -- asserted only against that bound, and
    DESIGN.md names it: activations of this model are O(1), 30 binary orders away."""
    M, K, N = 33000, 256, 256            # enough row tiles for the bf16x3 kernel to take the launch
    A, B = rnd(M, K, seed=5) * 2.0 ** -112, rnd(K, N, seed=6) * 2.0 ** 70
    A64, B64 = A.double(), B.double()
    ref = A64 @ B64
    e0 = comp_err_u(run_nn_image(ops, A, B, 0), A64, B64, ref)
    e1 = comp_err_u(run_nn_image(ops, A, B, 1), A64, B64, ref)
    print(f"ACCURACY denormal low pieces: fp32 MFMA {e0:.2f} u, bf16x3 {e1:.2f} u")
    assert e0 < BOUND_U
    assert e1 < 2.0 ** 10, (e0, e1)      # <= 2^-14 relative to sum |a||b|: two pieces always survive


# ------------------------------------------------------------------------------------------------ activations
EDGE = [0.0, -0.0, 1.0, -1.0, 20.0, -20.0, 87.0, -87.0, 88.0, -88.0, 88.7, -88.7, 89.0, -89.0, 103.0, -103.0, 1e4, -1e4,
        3e38, -3e38, float("inf"), float("-inf"), float("nan"), 1e-30, -1e-30, 1e-40, -1e-40]


def same_special(got, want, atol, rtol=2e-6):
    got, want = got.double().cpu(), want.double().cpu()
    nan_ok = torch.isnan(got) == torch.isnan(want)
    inf_ok = torch.where(torch.isinf(want), got == want, torch.ones_like(nan_ok))
    fin = torch.isfinite(want)
    close = torch.where(fin, (got - want).abs() <= atol + rtol * want.abs(), torch.ones_like(nan_ok))
    return bool(nan_ok.all() and inf_ok.all() and close.all()), (got, want)


@pytest.mark.parametrize("precision", [0, 1])
def test_silu_epilogue_edge_values(ops, precision):
    """out_act: C = silu(A B^T + bias) with A B^T = 0 exactly and the edge values in the bias, one per column; cpre
    receives the pre-activation.  NaN / +-inf propagate like torch.nn.functional.silu (silu(-inf) = NaN in both)."""
    v = torch.tensor(EDGE + [0.0] * (64 - len(EDGE)))
    M, K, N = 130, 32, 64
    A, B = torch.zeros(M, K, device=dev()), rnd(N, K, seed=1).to(dev())
    C, Cpre = torch.empty(M, N, device=dev()), torch.empty(M, N, device=dev())
    ops.gemm(A, B, C, out_act=True, bias=v.to(dev()), cpre=Cpre, precision=precision)
    want = torch.nn.functional.silu(v.double()).float().expand(M, N)
    ok, pair = same_special(C, want, atol=1e-30)
    assert ok, pair
    assert torch.equal(torch.nan_to_num(Cpre.cpu(), nan=7.0), torch.nan_to_num(v.expand(M, N), nan=7.0))


@pytest.mark.parametrize("precision", [0, 1])
def test_dsilu_epilogue_edge_values(ops, precision):
    """dact: C = (A B^T) * silu'(pre) with A B^T = 1 exactly and the edge values in `pre`."""
    v = torch.tensor(EDGE + [0.5] * (64 - len(EDGE)))
    M, K, N = 70, 16, 64
    A = torch.zeros(M, K, device=dev())
    A[:, 0] = 1.0
    B = torch.zeros(N, K, device=dev())
    B[:, 0] = 1.0
    pre = v.expand(M, N).contiguous().to(dev())
    C = torch.empty(M, N, device=dev())
    ops.gemm(A, B, C, dact=pre, precision=precision)
    x = v.double()
    s = torch.sigmoid(x)
    want = (s * (1 + x * (1 - s))).float().expand(M, N)        # torch's silu backward formula
    ok, pair = same_special(C, want, atol=1e-30)
    assert ok, pair


@pytest.mark.parametrize("precision", [0, 1])
def test_silu_prologue_finite_edge_values(ops, precision):
    """a_act: C = silu(A) @ I.  Finite edge values only: an infinite or NaN activation times the identity's zeros is
    NaN in every column of its row -- in torch too -- so +-inf / NaN are covered by the epilogue tests above."""
    fin = [x for x in EDGE if math.isfinite(x) and abs(x) < 1e30]
    K = 64
    v = torch.tensor(fin + [0.25] * (K - len(fin)))
    M = 200
    A = v.expand(M, K).contiguous().to(dev())
    eye = torch.eye(K, device=dev())
    C = torch.empty(M, K, device=dev())
    ops.gemm(A, eye, C, a_act=True, precision=precision)
    want = torch.nn.functional.silu(v.double()).float().expand(M, K)
    ok, pair = same_special(C, want, atol=1e-30, rtol=(2e-6 if precision == 0 else 4e-6))
    assert ok, pair


def test_gate_sigmoid_edge_values(ops):
    """cn_sigmoid in the forward gate: sigma = env * sigmoid(bn(g)) with mean 0, rstd 1, gamma 1, beta 0, so that the gate
    sees the edge values themselves; sender = 1 makes the aggregated row the sum of the sigmas."""
    D = 64
    v = torch.tensor(EDGE + [0.0] * (D - len(EDGE)))
    N, deg = 5, 3
    E = N * deg
    tgt = torch.arange(N).repeat_interleave(deg)
    ei = torch.stack([(tgt + 1) % N, tgt]).to(dev())                     # sorted by target, `deg` edges each
    lay = ops.GraphLayout(ei, N, torch.tensor([0, N], dtype=torch.int64, device=dev()))
    gs = torch.cat([v.expand(E, D), torch.ones(E, D)], dim=1).contiguous().to(dev())
    e_in = torch.zeros(E, D, device=dev())
    mr = torch.cat([torch.zeros(D), torch.ones(D)]).to(dev())
    ga, be = torch.ones(D, device=dev()), torch.zeros(D, device=dev())
    e_out, aggr = torch.empty(E, D, device=dev()), torch.empty(N, D, device=dev())
    nparts = ops.gate_nparts(N)
    ps, pq = (torch.zeros(nparts * D, dtype=torch.float64, device=dev()) for _ in range(2))
    ops.gate_scatter_fwd(gs, e_in, None, lay, mr, ga, be, e_out, aggr, ps, pq)
    want = torch.sigmoid(v.double()).float()
    ok, pair = same_special(e_out, want.expand(E, D), atol=1e-30)
    assert ok, pair
    ok, pair = same_special(aggr, (want.double() * deg).float().expand(N, D), atol=1e-30, rtol=4e-6)
    assert ok, pair
