"""Host-side wrappers over the C ABI: each function validates shapes / dtypes / devices on the host (a bad
operand must never reach a hand-written kernel), then passes raw device pointers and torch's current stream.

Operands are given as 2-D torch *views* (unit stride in the last dim); the leading dimension handed to the kernel
is the view's row stride, so column blocks of a reference-shaped weight such as ``MLP_gate.0.weight[:, D:2D]`` are
used in place, without packing.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import torch

from . import lib as _l

Tensor = torch.Tensor


def profile_gemm(enable: bool, only: Optional[int] = None, every: int = 1) -> None:
    """Start / stop HIP-event timing of cartnet_gemm launches (see include/cartnet_hip.h); ``only`` restricts it to
    one variant (the ``variant`` number profile_gemm_read reports), ``every`` to every n-th qualifying launch."""
    _l.check(_l.load().cartnet_profile_gemm_only(-1 if only is None else int(only)), "cartnet_profile_gemm_only")
    _l.check(_l.load().cartnet_profile_gemm_every(int(every)), "cartnet_profile_gemm_every")
    _l.check(_l.load().cartnet_profile_gemm(int(enable)), "cartnet_profile_gemm")


def profile_gemm_read() -> dict:
    """Per-variant totals {name: {launches, flops, ms}} of the launches recorded since profiling was enabled."""
    buf = (_l.GemmProfile * 64)()
    n = _l.load().cartnet_profile_gemm_read(buf, 64)
    if n < 0:
        _l.check(1, "cartnet_profile_gemm_read")
    out = {}
    for i in range(n):
        v = buf[i].variant
        # bit 18: the persistent kernel (csrc/gemm_f32p.h) took the launch
        name = ("tn" if v & 1 else ("nn" if v & 2 else "nt")) + str(64 * ((v >> 4) & 15)) + ("p" if v & (1 << 18) else "") + \
            ("+silu(A)" if v & 4 else "") + ("+out" if v & 512 else "") + ("+silu(B)" if v & 8 else "") + \
            (f"[{'E' if v & 256 else 'N'}-rows,{'M' if v & 1 else 'K'}={16 * ((v >> 10) & 255)}]")
        out[name] = {"launches": int(buf[i].launches), "flops": float(buf[i].flops), "ms": float(buf[i].ms),
                     "variant": int(v)}
    return out


def _f32_2d(t: Tensor, name: str, half_ok: bool = False) -> None:
    """2-D fp32 CUDA tensor with unit column stride; ``half_ok``: bf16 storage is accepted too (precision-2 GEMMs)."""
    if not (torch.is_tensor(t) and t.is_cuda and t.dim() == 2 and
            (t.dtype == torch.float32 or (half_ok and t.dtype == torch.bfloat16))):
        raise ValueError(f"{name}: expected a 2-D fp32 CUDA tensor, got {type(t).__name__} "
                         f"{getattr(t, 'dtype', None)} {tuple(getattr(t, 'shape', ()))} cuda={getattr(t, 'is_cuda', None)}")
    if t.shape[1] > 1 and t.stride(1) != 1:
        raise ValueError(f"{name}: last dimension must have unit stride (strides {t.stride()})")
    if t.shape[0] > 1 and t.stride(0) < t.shape[1]:
        raise ValueError(f"{name}: overlapping rows (strides {t.stride()}, shape {tuple(t.shape)})")


def _ld(t: Tensor) -> int:
    return int(t.stride(0)) if t.shape[0] > 1 else max(int(t.stride(0)), int(t.shape[1]))


def _vec(t: Optional[Tensor], n: int, name: str, dtype=torch.float32) -> None:
    if t is None:
        return
    if not (t.is_cuda and t.dtype == dtype and t.is_contiguous() and t.numel() >= n):
        raise ValueError(f"{name}: expected contiguous {dtype} CUDA tensor with >= {n} elements, "
                         f"got {t.dtype} {tuple(t.shape)}")


def _aslist(x, n):
    if x is None:
        return [None] * n
    if torch.is_tensor(x):
        return [x]
    return list(x)


def gemm(A: Sequence[Tensor] | Tensor, B: Sequence[Tensor] | Tensor, C_out: Sequence[Tensor] | Tensor, *,
         a_kstrided: bool = False, b_kstrided: bool = False, a_act: bool = False, b_act: bool = False,
         out_act: bool = False, segments: bool = False, bias=None, gather_i=None, gather_j=None, tgt=None, src=None,
         resid=None, dact=None, cpre=None, colsum=None, colsq=None, splitk: int = 1, precision: int = 0,
         b_split=None, b_split_folded=None, a_act_out=None, tile_policy: int = 0, gate_stats=None,
         dact_kind: int = 0) -> None:
    """C[g] = epilogue(sum_s opA(A[s]) @ opB(B[s])) on the fp32 matrix cores (see include/cartnet_hip.h).

    A / B / C_out: one tensor or a list.  With ``segments=False`` the lists are independent problems (groups) of
    identical shape; with ``segments=True`` A and B list K-segments that are summed into the single output.
    Shapes: A [M,K] (or [K,M] if a_kstrided), B [N,K] (or [K,N] if b_kstrided), C [M,N]
    (for splitk > 1: C is [splitk*M, N] contiguous slabs).
    ``a_act_out`` (with ``a_act`` and ``b_split`` at precision 0): tensors shaped and strided like A that receive silu(A).
    ``tile_policy``: CartnetGemmArgs.tile_policy (0 automatic; 1 narrow tiles for grouped N = 256 products too; 3 the persistent
    kernel wherever it has the form; 128 / 256 force a tile kernel).
    ``gate_stats`` = (g [M, N] (a column block of a wider matrix is fine), env [M] or None, mean_rstd [2N], gamma [N],
    beta [N]): CartnetGemmArgs.gst_* -- colsum / colsq then receive the partial sums of v w and v w ghat (see the header);
    raises unless the launch reaches the kernel that carries that epilogue.
    ``dact_kind``: 0 = the ``dact`` factor is silu'(dact), 1 = sigmoid(dact) (softplus backward).
    """
    lib = _l.load()
    A, B, C_out = _aslist(A, 1), _aslist(B, 1), _aslist(C_out, 1)
    if len(A) != len(B) or not (1 <= len(A) <= _l.MAX_GROUPS):
        raise ValueError(f"gemm: {len(A)} A operands vs {len(B)} B operands (max {_l.MAX_GROUPS})")
    nptr = len(A)
    ngroups, nsegs = (1, nptr) if segments else (nptr, 1)
    if len(C_out) != ngroups:
        raise ValueError(f"gemm: expected {ngroups} outputs, got {len(C_out)}")
    for i, (a, b) in enumerate(zip(A, B)):
        _f32_2d(a, f"gemm A[{i}]", half_ok=True)
        _f32_2d(b, f"gemm B[{i}]", half_ok=True)

    def is_half(ts, what):          # bf16 storage of an operand: all of its tensors or none
        ts = [t for t in ts if t is not None]
        n = sum(t.dtype == torch.bfloat16 for t in ts)
        if n not in (0, len(ts)):
            raise ValueError(f"gemm {what}: fp32 and bf16 tensors mixed")
        return n > 0
    M, K = (A[0].shape[1], A[0].shape[0]) if a_kstrided else (A[0].shape[0], A[0].shape[1])
    N, Kb = (B[0].shape[1], B[0].shape[0]) if b_kstrided else (B[0].shape[0], B[0].shape[1])
    if K != Kb:
        raise ValueError(f"gemm: inner dimensions differ: A gives K={K}, B gives K={Kb}")
    lda, ldb = _ld(A[0]), _ld(B[0])
    for i in range(nptr):
        if tuple(A[i].shape) != tuple(A[0].shape) or tuple(B[i].shape) != tuple(B[0].shape) or \
                _ld(A[i]) != lda or _ld(B[i]) != ldb:
            raise ValueError("gemm: all groups / segments must share shapes and leading dimensions")
    rows_c = M * splitk
    for i, c in enumerate(C_out):
        _f32_2d(c, f"gemm C[{i}]", half_ok=True)
        if tuple(c.shape) != (rows_c, N):
            raise ValueError(f"gemm C[{i}]: expected shape {(rows_c, N)}, got {tuple(c.shape)}")
    ldc = _ld(C_out[0])
    if splitk > 1 and ldc != N:
        raise ValueError("gemm: split-K slabs must be contiguous [splitk*M, N]")
    args = _l.GemmArgs()
    args.M, args.N, args.K = int(M), int(N), int(K)
    args.lda, args.ldb, args.ldc = lda, ldb, ldc
    args.ngroups, args.nsegs, args.splitk = ngroups, nsegs, int(splitk)
    args.a_kstrided, args.b_kstrided = int(a_kstrided), int(b_kstrided)
    args.a_act, args.b_act, args.out_act = int(a_act), int(b_act), int(out_act)
    args.precision = int(precision)
    args.tile_policy = int(tile_policy)
    args.dact_kind = int(dact_kind)
    args.a_half, args.b_half, args.c_half = int(is_half(A, "A")), int(is_half(B, "B")), int(is_half(C_out, "C"))
    for i in range(nptr):
        args.A[i] = A[i].data_ptr()
        args.B[i] = B[i].data_ptr()
    if b_split is not None:
        b_split = _aslist(b_split, nptr)
        if len(b_split) != nptr or not b_kstrided or a_kstrided:
            raise ValueError("gemm: b_split needs one image per B operand, b_kstrided=True and a_kstrided=False")
        need = int((lib.cartnet_gemm_pack_b_bytes if precision == 0 else lib.cartnet_gemm_split_b_bytes)(int(K), int(N)))
        for i, t in enumerate(b_split):
            if t is None:
                continue
            if need == 0 or t.dtype != torch.uint8 or not t.is_cuda or not t.is_contiguous() or t.numel() < need:
                raise ValueError(f"gemm b_split[{i}]: expected a contiguous uint8 CUDA tensor of {need} bytes")
            args.b_split[i] = t.data_ptr()
    if b_split_folded is not None:
        need = nsegs * int((lib.cartnet_gemm_pack_b_bytes if precision == 0 else lib.cartnet_gemm_split_b_bytes)(int(K), int(N)))
        t = b_split_folded
        if nsegs < 2 or need == 0 or t.dtype != torch.uint8 or not t.is_cuda or not t.is_contiguous() or t.numel() < need:
            raise ValueError(f"gemm b_split_folded: needs segments and a contiguous uint8 CUDA tensor of {need} bytes")
        args.b_split_folded = t.data_ptr()
    for g in range(ngroups):
        if _ld(C_out[g]) != ldc:
            raise ValueError("gemm: outputs must share a leading dimension")
        args.C[g] = C_out[g].data_ptr()

    def per_group(name, val, shape, field, ldfield=None, vector=False):
        vals = _aslist(val, ngroups)
        if len(vals) != ngroups:
            raise ValueError(f"gemm {name}: expected {ngroups} entries")
        ld = None
        for g, t in enumerate(vals):
            if t is None:
                continue
            if vector:
                _vec(t, shape, f"gemm {name}[{g}]")
            else:
                _f32_2d(t, f"gemm {name}[{g}]", half_ok=(name == "dact"))
                if shape is not None and tuple(t.shape) != shape:
                    raise ValueError(f"gemm {name}[{g}]: expected shape {shape}, got {tuple(t.shape)}")
                if ld is None:
                    ld = _ld(t)
                elif _ld(t) != ld:
                    raise ValueError(f"gemm {name}: entries must share a leading dimension")
            getattr(args, field)[g] = t.data_ptr()
        if ldfield is not None and ld is not None:
            setattr(args, ldfield, ld)
        return vals

    per_group("bias", bias, N, "bias", vector=True)
    gi = per_group("gather_i", gather_i, None, "gather_i", "ldg")
    gj = per_group("gather_j", gather_j, None, "gather_j", "ldg")
    if any(t is not None for t in gi):
        _vec(tgt, M, "gemm tgt", torch.int32)
        _vec(src, M, "gemm src", torch.int32)
        if tgt is None or src is None:
            raise ValueError("gemm: gather epilogue needs tgt and src")
        for a, b in zip(gi, gj):
            if (a is None) != (b is None) or (a is not None and (a.shape[1] != N or b.shape[1] != N or
                                                                  _ld(a) != _ld(b))):
                raise ValueError("gemm: gather_i / gather_j must pair up with N columns and equal row stride")
        args.tgt, args.src = tgt.data_ptr(), src.data_ptr()
        args.gather_rows = max(t.shape[0] for t in list(gi) + list(gj) if t is not None)
    per_group("resid", resid, (M, N), "resid", "ldr")
    args.dact_half = int(is_half(per_group("dact", dact, (M, N), "dact", "ldd"), "dact"))
    cp = per_group("cpre", cpre, (M, N), "cpre")
    for t in cp:
        if t is not None and _ld(t) != ldc:
            raise ValueError("gemm: cpre must share the output's leading dimension")
    ao = per_group("a_act_out", a_act_out, (M, K), "a_act_out")
    for t in ao:
        if t is not None and (_ld(t) != lda or a_kstrided):
            raise ValueError("gemm: a_act_out must share A's row stride (and A must be k-contiguous)")
    tiles_m = (M + 127) // 128
    cs = _aslist(colsum, ngroups)
    cq = _aslist(colsq, ngroups)
    for g in range(ngroups):
        for name, t, field in (("colsum", cs[g], "colsum"), ("colsq", cq[g], "colsq")):
            if t is None:
                continue
            _vec(t, tiles_m * N, f"gemm {name}[{g}]", torch.float64)
            getattr(args, field)[g] = t.data_ptr()
    if gate_stats is not None:
        gg, genv, gmr, ggam, gbet = gate_stats
        for nm, t in (("g", gg), ("mean_rstd", gmr), ("gamma", ggam), ("beta", gbet)):      # only env may be None
            if t is None:
                raise ValueError(f"gemm gate_stats {nm}: required (only env may be None)")
        _f32_2d(gg, "gemm gate_stats g")
        if tuple(gg.shape) != (M, N):
            raise ValueError(f"gemm gate_stats g: expected shape {(M, N)}, got {tuple(gg.shape)}")
        _vec(genv, M, "gemm gate_stats env")
        _vec(gmr, 2 * N, "gemm gate_stats mean_rstd")
        _vec(ggam, N, "gemm gate_stats gamma")
        _vec(gbet, N, "gemm gate_stats beta")
        args.gst_g, args.gst_ld, args.gst_env = gg.data_ptr(), _ld(gg), _l.ptr(genv)
        args.gst_mean_rstd, args.gst_gamma, args.gst_beta = gmr.data_ptr(), ggam.data_ptr(), gbet.data_ptr()
        if not lib.cartnet_gemm_gate_stats_ok(C.byref(args)):
            raise ValueError("gemm gate_stats: this launch does not reach the kernel with the gate-statistics epilogue "
                             "(precision 0 / 1, N = 256, weight image, resid + colsum + colsq only, >= 64 row tiles; g, "
                             "mean_rstd, gamma, beta 16-byte aligned and g's row stride a multiple of 4 elements)")
    _l.check(lib.cartnet_gemm(C.byref(args), _l.stream_ptr()), "cartnet_gemm")


def pack_b(mats: Sequence[Tensor], outs: Optional[Sequence[Tensor]] = None) -> list:
    """fp32 images (cartnet_gemm_pack_b) of k-strided GEMM operands for precision-0 calls; see split_b."""
    return split_b(mats, _fp32=True, outs=outs)


def split_b(mats: Sequence[Tensor], _fp32: bool = False, outs: Optional[Sequence[Tensor]] = None) -> list:
    """bf16x3 pre-split images (cartnet_gemm_split_b) of k-strided GEMM operands: each entry is a 2-D fp32 view
    B [K, N] with arbitrary strides (``W.t()`` of a weight W [out, in] gives the forward operand).  ``outs``: image
    buffers of an earlier call to refill in place (one launch per 80 matrices for any number of them)."""
    lib = _l.load()
    fn_bytes = lib.cartnet_gemm_pack_b_bytes if _fp32 else lib.cartnet_gemm_split_b_bytes
    fn = lib.cartnet_gemm_pack_b if _fp32 else lib.cartnet_gemm_split_b
    mats = list(mats)
    given = list(outs) if outs is not None else None
    if given is not None and len(given) != len(mats):
        raise ValueError("split_b: one output buffer per matrix")
    outs = []
    n = len(mats)
    src = (C.c_void_p * n)()
    dst = (C.c_void_p * n)()
    Ks, Ns, sk, sn = (C.c_int32 * n)(), (C.c_int32 * n)(), (C.c_int32 * n)(), (C.c_int32 * n)()
    for i, m in enumerate(mats):
        if m.dim() != 2 or m.dtype != torch.float32 or not m.is_cuda:
            raise ValueError(f"split_b[{i}]: expected a 2-D fp32 CUDA tensor")
        K, N = int(m.shape[0]), int(m.shape[1])
        nbytes = int(fn_bytes(K, N))
        if nbytes == 0:
            raise ValueError(f"split_b[{i}]: K={K} must be a multiple of 16 and N={N} of 256")
        if given is not None:
            out = given[i]
            if out.dtype != torch.uint8 or not out.is_cuda or not out.is_contiguous() or out.numel() < nbytes:
                raise ValueError(f"split_b outs[{i}]: expected a contiguous uint8 CUDA tensor of {nbytes} bytes")
        else:
            out = torch.empty(nbytes, dtype=torch.uint8, device=m.device)
        outs.append(out)
        src[i], dst[i] = m.data_ptr(), out.data_ptr()
        Ks[i], Ns[i], sk[i], sn[i] = K, N, int(m.stride(0)), int(m.stride(1))
    if n:
        _l.check(fn(src, dst, Ks, Ns, sk, sn, n, _l.stream_ptr()), "cartnet_gemm_split_b / pack_b")
    return outs


def gemm_tiles_m(M: int) -> int:
    """Number of row tiles (= partial-sum rows written by the colsum epilogue)."""
    return (int(M) + 127) // 128


def _ptr_array(tensors):
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


def splitk_reduce(slabs, splitk: int, outs) -> None:
    """outs[j] = sum over the ``splitk`` slabs of slabs[j] (fixed order); up to 4 same-shaped jobs per launch."""
    slabs, outs = _aslist(slabs, 1), _aslist(outs, 1)
    if len(slabs) != len(outs) or not (1 <= len(outs) <= _l.MAX_GROUPS):
        raise ValueError("splitk_reduce: need 1..4 matching slab/out pairs")
    M, N = outs[0].shape
    ldo = _ld(outs[0])
    for sl, o in zip(slabs, outs):
        _f32_2d(o, "splitk_reduce out")
        if tuple(o.shape) != (M, N) or _ld(o) != ldo:
            raise ValueError("splitk_reduce: outputs must share shape and leading dimension")
        _vec(sl, splitk * M * N, "splitk_reduce slabs")
    _l.check(_l.load().cartnet_splitk_reduce(_ptr_array(slabs), _ptr_array(outs), len(outs), int(splitk), M, N, ldo,
                                             _l.stream_ptr()), "cartnet_splitk_reduce")


def colsum_finalize(parts, nparts: int, outs) -> None:
    """outs[j][n] = sum_p parts[j][p, n] in fixed order.  parts: fp64 partial rows (up to 8 same-shaped jobs per
    launch) or, for the head kernels, one fp32 partial matrix."""
    parts, outs = _aslist(parts, 1), _aslist(outs, 1)
    if len(parts) != len(outs) or not (1 <= len(outs) <= 8):
        raise ValueError("colsum_finalize: need 1..8 matching parts/out pairs")
    n = outs[0].numel()
    for o in outs:
        _vec(o, n, "colsum_finalize out")
        if o.numel() != n:
            raise ValueError("colsum_finalize: outputs must have equal length")
    if parts[0].dtype == torch.float64:
        for pt in parts:
            _vec(pt, nparts * n, "colsum_finalize parts", torch.float64)
        _l.check(_l.load().cartnet_colsum_finalize(_ptr_array(parts), _ptr_array(outs), len(outs), int(nparts), n,
                                                   _l.stream_ptr()), "cartnet_colsum_finalize")
    else:
        if len(parts) != 1:
            raise ValueError("colsum_finalize: fp32 partials are finalised one at a time")
        _vec(parts[0], nparts * n, "colsum_finalize parts", torch.float32)
        _l.check(_l.load().cartnet_colsum_finalize_f32(parts[0].data_ptr(), int(nparts), n, outs[0].data_ptr(),
                                                       _l.stream_ptr()), "cartnet_colsum_finalize_f32")


class GraphLayout:
    """Device-side CSR (by target) / CSC (by source) layout of one batch; built once per batch."""

    def __init__(self, edge_index: Tensor, num_nodes: int, graph_ptr: Optional[Tensor], need_csc: bool = True):
        if not (edge_index.is_cuda and edge_index.dtype == torch.int64 and edge_index.dim() == 2 and
                edge_index.shape[0] == 2):
            raise ValueError("edge_index must be a CUDA int64 tensor of shape [2, E]")
        edge_index = edge_index.contiguous()
        dev = edge_index.device
        E, N = int(edge_index.shape[1]), int(num_nodes)
        self.E, self.N = E, N
        self.src = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
        self.tgt = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
        self.rowptr = torch.empty(N + 1, dtype=torch.int32, device=dev)
        self.colptr = torch.empty(N + 1, dtype=torch.int32, device=dev) if need_csc else None
        self.perm = torch.empty(max(E, 1), dtype=torch.int32, device=dev) if need_csc else None
        self.status = torch.empty(1, dtype=torch.int32, device=dev)
        Bg = 1
        gp = None
        if graph_ptr is not None:
            if not (graph_ptr.is_cuda and graph_ptr.dtype == torch.int64 and graph_ptr.is_contiguous()):
                raise ValueError("graph_ptr must be a contiguous CUDA int64 tensor")
            Bg = int(graph_ptr.numel()) - 1
            gp = graph_ptr.data_ptr()
        _l.check(_l.load().cartnet_csr_build(
            edge_index.data_ptr(), E, N, gp, Bg, self.src.data_ptr(), self.tgt.data_ptr(), self.rowptr.data_ptr(),
            _l.ptr(self.colptr), _l.ptr(self.perm), self.status.data_ptr(), _l.stream_ptr()), "cartnet_csr_build")

    def validate(self) -> None:
        """Host check of the device status word (one sync; call once per batch in debug / tests)."""
        raise_on_graph_status(int(self.status.item()))


def raise_on_graph_status(s: int) -> None:
    """Decode the status word cartnet_csr_build leaves on the device."""
    if s:
        why = []
        if s & 1:
            why.append("edge_index[1] is not sorted ascending")
        if s & 2:
            why.append("an index is outside [0, N)")
        if s & 4:
            why.append("an edge connects two different crystals")
        if s & 8:
            why.append("a crystal has more than 8192 atoms")
        if s & 16:
            why.append("an atomic number is outside [0, 119) (the embedding table, cartnet.py:113)")
        if s & 32:
            why.append("a batch id is outside [0, num_graphs)")
        raise ValueError("invalid graph: " + "; ".join(why))


def edge_features(cart_dist: Tensor, cart_dir: Optional[Tensor], means: Tensor, betas: Tensor, invariant: bool,
                  radius: float, env_radius: float, feat: Tensor, env: Optional[Tensor]) -> None:
    E = int(cart_dist.numel())
    R = int(means.numel())
    _vec(cart_dist, E, "edge_features cart_dist")
    _vec(means, R, "edge_features means")
    _vec(betas, R, "edge_features betas")
    if not invariant:
        if cart_dir is None or not (cart_dir.is_cuda and cart_dir.dtype == torch.float32 and
                                    cart_dir.is_contiguous() and tuple(cart_dir.shape) == (E, 3)):
            raise ValueError("edge_features: cart_dir must be a contiguous fp32 CUDA tensor [E,3]")
    _f32_2d(feat, "edge_features feat")
    if feat.shape[0] != E or not feat.is_contiguous():
        raise ValueError("edge_features: feat must be contiguous [E, ldf]")
    _vec(env, E, "edge_features env")
    _l.check(_l.load().cartnet_edge_features(
        cart_dist.data_ptr(), _l.ptr(cart_dir), means.data_ptr(), betas.data_ptr(), E, R, int(invariant),
        float(radius), float(env_radius), feat.data_ptr(), int(feat.shape[1]), _l.ptr(env), _l.stream_ptr()),
        "cartnet_edge_features")


def node_embed(z, batch, temperature, emb, wt, bt, bias, x0: Tensor, status: Optional[Tensor] = None) -> None:
    """x0 = emb[z] + temperature projection + bias (cartnet_node_embed).  Atomic numbers / batch ids outside their
    tables are clamped and reported in ``status`` (int32 [1], bits 16 / 32; decode with raise_on_graph_status)."""
    _f32_2d(x0, "node_embed x0")
    N, Cc = x0.shape
    if not x0.is_contiguous():
        raise ValueError("node_embed: x0 must be contiguous")
    if z is not None:
        _vec(z, N, "node_embed z", torch.int64)
    if batch is not None:
        _vec(batch, N, "node_embed batch", torch.int64)
    if emb is not None:
        _f32_2d(emb, "node_embed emb")
        if emb.shape[1] != Cc or not emb.is_contiguous():
            raise ValueError("node_embed: embedding table must be contiguous [n_types, C]")
    _vec(wt, Cc, "node_embed wt")
    _vec(bt, Cc, "node_embed bt")
    _vec(bias, Cc, "node_embed bias")
    Bg = 0
    if temperature is not None:
        if not (temperature.is_cuda and temperature.dtype == torch.float32 and temperature.dim() == 1 and
                temperature.is_contiguous() and temperature.numel() >= 1):
            raise ValueError("node_embed: temperature must be a contiguous fp32 CUDA vector [Bg]")
        Bg = int(temperature.numel())
    if status is not None:
        _vec(status, 1, "node_embed status", torch.int32)
    n_types = int(emb.shape[0]) if emb is not None else 0
    _l.check(_l.load().cartnet_node_embed(_l.ptr(z), _l.ptr(batch), _l.ptr(temperature), _l.ptr(emb), _l.ptr(wt),
                                          _l.ptr(bt), _l.ptr(bias), N, Cc, n_types, Bg, _l.ptr(status), x0.data_ptr(),
                                          _l.stream_ptr()),
             "cartnet_node_embed")


def node_nparts(N: int) -> int:
    return int(_l.load().cartnet_node_nparts(int(N)))


def gate_nparts(N: int) -> int:
    return int(_l.load().cartnet_gate_scatter_nparts(int(N)))


def node_embed_bwd(batch, temperature, dx0: Tensor, parts_w, parts_b) -> None:
    _f32_2d(dx0, "node_embed_bwd dx0")
    N, Cc = dx0.shape
    if not dx0.is_contiguous():
        raise ValueError("node_embed_bwd: dx0 must be contiguous")
    npart = node_nparts(N)
    _vec(parts_w, npart * Cc, "node_embed_bwd parts_w", torch.float64)
    _vec(parts_b, npart * Cc, "node_embed_bwd parts_b", torch.float64)
    if batch is not None:
        _vec(batch, N, "node_embed_bwd batch", torch.int64)
    if temperature is not None and batch is None:
        raise ValueError("node_embed_bwd: temperature needs batch")
    Bg = int(temperature.numel()) if temperature is not None else 0
    _l.check(_l.load().cartnet_node_embed_bwd(_l.ptr(batch), _l.ptr(temperature), dx0.data_ptr(), N, Cc, Bg,
                                              parts_w.data_ptr(), parts_b.data_ptr(), _l.stream_ptr()),
             "cartnet_node_embed_bwd")


def sort_by_key(keys: Tensor, nkeys: int):
    """Stable counting sort on the device: returns (perm int32 [N], ptr int32 [nkeys+1], status int32 [1])."""
    N = int(keys.numel())
    _vec(keys, N, "sort_by_key keys", torch.int64)
    dev = keys.device
    perm = torch.empty(max(N, 1), dtype=torch.int32, device=dev)
    ptr_ = torch.empty(nkeys + 1, dtype=torch.int32, device=dev)
    status = torch.empty(1, dtype=torch.int32, device=dev)
    _l.check(_l.load().cartnet_sort_by_key(keys.data_ptr(), N, int(nkeys), perm.data_ptr(), ptr_.data_ptr(),
                                           status.data_ptr(), _l.stream_ptr()), "cartnet_sort_by_key")
    return perm, ptr_, status


def segment_sum_long(rows: Tensor, ptr_: Tensor, perm: Optional[Tensor], total: int, out: Tensor) -> None:
    """out[s] = sum of rows[perm[p]] for p in [ptr[s], ptr[s+1]) -- for few, uneven segments."""
    _f32_2d(rows, "segment_sum_long rows")
    _f32_2d(out, "segment_sum_long out")
    nseg, W = out.shape
    if rows.shape[1] != W or rows.shape[0] < total:
        raise ValueError("segment_sum_long: shape mismatch")
    _vec(ptr_, nseg + 1, "segment_sum_long ptr", torch.int32)
    if perm is not None:
        _vec(perm, total, "segment_sum_long perm", torch.int32)
    tmp = torch.empty((max(total, 1), W), dtype=torch.float32, device=rows.device)
    _l.check(_l.load().cartnet_segment_sum_long(rows.data_ptr(), _ld(rows), ptr_.data_ptr(), _l.ptr(perm), nseg,
                                                int(total), W, tmp.data_ptr(), out.data_ptr(), _ld(out),
                                                _l.stream_ptr()), "cartnet_segment_sum_long")



def segment_sum_chunked(rows: Tensor, ptr_: Tensor, perm: Optional[Tensor], total: int, out: Tensor) -> None:
    """segment_sum_long with numbered partial rows (cartnet_segment_sum_chunked): the workspace is
    cartnet_segment_chunked_rows(nseg, total) rows instead of ``total`` -- sums over all edges per crystal."""
    _f32_2d(rows, "segment_sum_chunked rows")
    _f32_2d(out, "segment_sum_chunked out")
    nseg, W = out.shape
    if rows.shape[1] < W or rows.shape[0] < total:
        raise ValueError("segment_sum_chunked: shape mismatch")
    _vec(ptr_, nseg + 1, "segment_sum_chunked ptr", torch.int32)
    if perm is not None:
        _vec(perm, total, "segment_sum_chunked perm", torch.int32)
    lib = _l.load()
    tmp = torch.empty((int(lib.cartnet_segment_chunked_rows(nseg, int(total))), W), dtype=torch.float32, device=rows.device)
    _l.check(lib.cartnet_segment_sum_chunked(rows.data_ptr(), _ld(rows), ptr_.data_ptr(), _l.ptr(perm), nseg, int(total), W,
                                             tmp.data_ptr(), out.data_ptr(), _ld(out), _l.stream_ptr()),
             "cartnet_segment_sum_chunked")

def segment_sum_chunked_fold3(rows: Tensor, ptr_: Tensor, total: int, out: Tensor, fold_out: Tensor) -> None:
    """cartnet_segment_sum_chunked_fold3: the chunked per-segment sums of rows [total, W] (no permutation) and, from the same
    read, fold_out[p] = (rows[p, :W/3] + rows[p, W/3:2W/3]) + rows[p, 2W/3:]  ([total, W/3] contiguous)."""
    _f32_2d(rows, "segment_sum_chunked_fold3 rows")
    _f32_2d(out, "segment_sum_chunked_fold3 out")
    _f32_2d(fold_out, "segment_sum_chunked_fold3 fold_out")
    nseg, W = out.shape
    if rows.shape[1] != W or rows.shape[0] < total or W % 12 or tuple(fold_out.shape) != (int(total), W // 3) or \
            not fold_out.is_contiguous():
        raise ValueError("segment_sum_chunked_fold3: rows [total, W], W % 12 == 0, fold_out contiguous [total, W / 3]")
    _vec(ptr_, nseg + 1, "segment_sum_chunked_fold3 ptr", torch.int32)
    lib = _l.load()
    tmp = torch.empty((int(lib.cartnet_segment_chunked_rows(nseg, int(total))), W), dtype=torch.float32, device=rows.device)
    _l.check(lib.cartnet_segment_sum_chunked_fold3(rows.data_ptr(), _ld(rows), ptr_.data_ptr(), nseg, int(total), W,
                                                   tmp.data_ptr(), out.data_ptr(), _ld(out), fold_out.data_ptr(),
                                                   _l.stream_ptr()), "cartnet_segment_sum_chunked_fold3")


def bn_finalize(parts_sum, parts_sq, nparts: int, count: int, Cc: int, eps: float, momentum: float, training: bool,
                running_mean, running_var, nbt, mean_rstd: Tensor) -> None:
    _vec(mean_rstd, 2 * Cc, "bn_finalize mean_rstd")
    if training:
        _vec(parts_sum, nparts * Cc, "bn_finalize parts_sum", torch.float64)
        _vec(parts_sq, nparts * Cc, "bn_finalize parts_sq", torch.float64)
    _vec(running_mean, Cc, "bn_finalize running_mean")
    _vec(running_var, Cc, "bn_finalize running_var")
    if nbt is not None:
        _vec(nbt, 1, "bn_finalize num_batches_tracked", torch.int64)
    _l.check(_l.load().cartnet_bn_finalize(_l.ptr(parts_sum), _l.ptr(parts_sq), int(nparts), int(count), int(Cc),
                                           float(eps), float(momentum), int(training), _l.ptr(running_mean),
                                           _l.ptr(running_var), _l.ptr(nbt), mean_rstd.data_ptr(), None, 0, 0,
                                           _l.stream_ptr()),
             "cartnet_bn_finalize")


def _edge_rows(t: Tensor, E: int, W: int, name: str) -> None:
    _f32_2d(t, name)
    if tuple(t.shape) != (E, W) or not t.is_contiguous():
        raise ValueError(f"{name}: expected contiguous [{E},{W}], got {tuple(t.shape)}")


def gate_scatter_fwd(gs, e_in, env, layout: GraphLayout, mean_rstd, gamma, beta, e_out, aggr, parts_sum,
                     parts_sq, bc=None) -> None:
    """``bc`` [N, 2D] (optional): cartnet_gate_scatter_fwd_bc -- per target also sum s w | sum s w ghat for the backward pass."""
    E, N = layout.E, layout.N
    D = int(aggr.shape[1])
    _edge_rows(gs, E, 2 * D, "gate_scatter_fwd gs")
    if (e_in is None) != (e_out is None):
        raise ValueError("gate_scatter_fwd: e_in and e_out must both be given or both be None")
    if e_in is not None:
        _edge_rows(e_in, E, D, "gate_scatter_fwd e_in")
        _edge_rows(e_out, E, D, "gate_scatter_fwd e_out")
    _edge_rows(aggr, N, D, "gate_scatter_fwd aggr")
    _vec(env, E, "gate_scatter_fwd env")
    _vec(mean_rstd, 2 * D, "gate_scatter_fwd mean_rstd")
    _vec(gamma, D, "gate_scatter_fwd gamma")
    _vec(beta, D, "gate_scatter_fwd beta")
    npart = gate_nparts(N)
    _vec(parts_sum, npart * D, "gate_scatter_fwd parts_sum", torch.float64)
    _vec(parts_sq, npart * D, "gate_scatter_fwd parts_sq", torch.float64)
    if bc is not None:
        _edge_rows(bc, N, 2 * D, "gate_scatter_fwd bc")
        _l.check(_l.load().cartnet_gate_scatter_fwd_bc(gs.data_ptr(), _l.ptr(e_in), _l.ptr(env), layout.rowptr.data_ptr(),
                                                       mean_rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), N, D,
                                                       _l.ptr(e_out), aggr.data_ptr(), parts_sum.data_ptr(),
                                                       parts_sq.data_ptr(), bc.data_ptr(), _l.stream_ptr()),
                 "cartnet_gate_scatter_fwd_bc")
        return
    _l.check(_l.load().cartnet_gate_scatter_fwd(gs.data_ptr(), _l.ptr(e_in), _l.ptr(env), layout.rowptr.data_ptr(),
                                                mean_rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), N, D,
                                                _l.ptr(e_out), aggr.data_ptr(), parts_sum.data_ptr(),
                                                parts_sq.data_ptr(), None, _l.stream_ptr()),
             "cartnet_gate_scatter_fwd")


def gate_gemm_eval(pre: Tensor, img_gate: Tensor, img_aggr: Tensor, bias_gate: Tensor, bias_aggr: Tensor, mean_rstd: Tensor,
                   gamma: Tensor, beta: Tensor, env: Optional[Tensor], e_in: Tensor, layout: GraphLayout, e_out: Tensor,
                   aggr: Tensor) -> None:
    """Inference-mode fusion of a layer's second Linears with the gate (cartnet_gate_gemm_eval): pre [E, 2D] -> e_out
    [E, D] = e_in + sigma and aggr [N, D] = per-target sums of sigma * sender; the images come from ``pack_b([W2g.t(),
    W2a.t()])``; mean_rstd [2D] holds the edge BatchNorm's running mean and 1 / sqrt(running_var + eps)."""
    lib = _l.load()
    E, N = layout.E, layout.N
    D = int(aggr.shape[1])
    _edge_rows(pre, E, 2 * D, "gate_gemm_eval pre")
    _edge_rows(e_in, E, D, "gate_gemm_eval e_in")
    _edge_rows(e_out, E, D, "gate_gemm_eval e_out")
    _edge_rows(aggr, N, D, "gate_gemm_eval aggr")
    for name, t in (("bias_gate", bias_gate), ("bias_aggr", bias_aggr), ("gamma", gamma), ("beta", beta)):
        _vec(t, D, "gate_gemm_eval " + name)
    _vec(mean_rstd, 2 * D, "gate_gemm_eval mean_rstd")
    _vec(env, E, "gate_gemm_eval env")
    need = int(lib.cartnet_gemm_pack_b_bytes(D, D))
    for name, t in (("img_gate", img_gate), ("img_aggr", img_aggr)):
        if need == 0 or t.dtype != torch.uint8 or not t.is_cuda or not t.is_contiguous() or t.numel() < need:
            raise ValueError(f"gate_gemm_eval {name}: expected a contiguous uint8 CUDA tensor of {need} bytes (pack_b)")
    if D % 256 != 0:
        raise ValueError("gate_gemm_eval: D must be a multiple of 256")
    bnd = torch.empty(max(1, int(lib.cartnet_gate_gemm_eval_workspace(E, D)) // 4), dtype=torch.float32, device=aggr.device)
    a = _l.GateGemmArgs()
    a.pre, a.ldp = pre.data_ptr(), _ld(pre)
    a.img_gate, a.img_aggr = img_gate.data_ptr(), img_aggr.data_ptr()
    a.bias_gate, a.bias_aggr = bias_gate.data_ptr(), bias_aggr.data_ptr()
    a.mean_rstd, a.gamma, a.beta = mean_rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr()
    a.env = _l.ptr(env)
    a.e_in, a.e_out = e_in.data_ptr(), e_out.data_ptr()
    a.tgt, a.rowptr = layout.tgt.data_ptr(), layout.rowptr.data_ptr()
    a.aggr, a.bnd = aggr.data_ptr(), bnd.data_ptr()
    a.E, a.N, a.D = E, N, D
    _l.check(lib.cartnet_gate_gemm_eval(C.byref(a), _l.stream_ptr()), "cartnet_gate_gemm_eval")


def gate_scatter_bwd_stats(gs, de_out, daggr, env, layout: GraphLayout, mean_rstd, gamma, beta, parts_a,
                           parts_b) -> None:
    E, N = layout.E, layout.N
    D = int(daggr.shape[1])
    _edge_rows(gs, E, 2 * D, "gate_scatter_bwd_stats gs")
    if de_out is not None:
        _edge_rows(de_out, E, D, "gate_scatter_bwd_stats de_out")
    _edge_rows(daggr, N, D, "gate_scatter_bwd_stats daggr")
    _vec(env, E, "gate_scatter_bwd_stats env")
    _vec(mean_rstd, 2 * D, "mean_rstd")
    _vec(gamma, D, "gamma")
    _vec(beta, D, "beta")
    npart = gate_nparts(N)
    _vec(parts_a, npart * D, "parts_a", torch.float64)
    _vec(parts_b, npart * D, "parts_b", torch.float64)
    _l.check(_l.load().cartnet_gate_scatter_bwd_stats(
        gs.data_ptr(), _l.ptr(de_out), daggr.data_ptr(), _l.ptr(env), layout.rowptr.data_ptr(),
        mean_rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), N, D, parts_a.data_ptr(), parts_b.data_ptr(), None,
        _l.stream_ptr()), "cartnet_gate_scatter_bwd_stats")


def gate_scatter_bwd_apply(gs, de_out, daggr, env, layout: GraphLayout, mean_rstd, gamma, beta, sums, training: bool,
                           parts_dg, parts_ds) -> None:
    E, N = layout.E, layout.N
    D = int(daggr.shape[1])
    _edge_rows(gs, E, 2 * D, "gate_scatter_bwd_apply gs")
    if de_out is not None:
        _edge_rows(de_out, E, D, "gate_scatter_bwd_apply de_out")
    _edge_rows(daggr, N, D, "gate_scatter_bwd_apply daggr")
    _vec(env, E, "env")
    _vec(mean_rstd, 2 * D, "mean_rstd")
    _vec(gamma, D, "gamma")
    _vec(beta, D, "beta")
    _vec(sums, 2 * D, "sums")
    npart = gate_nparts(N)
    _vec(parts_dg, npart * D, "parts_dg", torch.float64)
    _vec(parts_ds, npart * D, "parts_ds", torch.float64)
    _l.check(_l.load().cartnet_gate_scatter_bwd_apply(
        gs.data_ptr(), _l.ptr(de_out), daggr.data_ptr(), _l.ptr(env), layout.rowptr.data_ptr(),
        mean_rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), sums.data_ptr(), E, int(training), N, D,
        parts_dg.data_ptr(), parts_ds.data_ptr(), None, _l.stream_ptr()), "cartnet_gate_scatter_bwd_apply")


def segment_sum(rows: Tensor, ptr_: Tensor, perm: Optional[Tensor], out: Tensor) -> None:
    _f32_2d(rows, "segment_sum rows")
    _f32_2d(out, "segment_sum out")
    N, W = out.shape
    if rows.shape[1] != W:
        raise ValueError("segment_sum: width mismatch")
    _vec(ptr_, N + 1, "segment_sum ptr", torch.int32)
    if perm is not None:
        _vec(perm, rows.shape[0], "segment_sum perm", torch.int32)
    _l.check(_l.load().cartnet_segment_sum(rows.data_ptr(), _ld(rows), ptr_.data_ptr(), _l.ptr(perm), N, W,
                                           out.data_ptr(), _ld(out), _l.stream_ptr()), "cartnet_segment_sum")


def segment_sum_pair(rows: Tensor, layout: "GraphLayout", out_t: Tensor, out_s: Tensor, ochunk: int = 256) -> None:
    """cartnet_segment_sum_pair: by-target and by-source sums of the same rows in one launch (out_t / out_s: [N, W] views
    with one row stride, e.g. the two halves of a wider matrix).  ``ochunk`` != 256: the 256-column chunk j of a sum is
    written at column j * ochunk of its output row (out_t / out_s are then views that START at the first chunk's place and
    are at least W wide; the caller owns the rest of the row)."""
    half = rows.dtype == torch.bfloat16
    _f32_2d(rows, "segment_sum_pair rows", half_ok=True)
    _f32_2d(out_t, "segment_sum_pair out_t")
    _f32_2d(out_s, "segment_sum_pair out_s")
    N, W = out_t.shape
    W = int(rows.shape[1])
    if out_s.shape[0] != N or _ld(out_s) != _ld(out_t) or rows.shape[0] != layout.E or N != layout.N or \
            (ochunk == 256 and (out_t.shape[1] != W or out_s.shape[1] != W)):
        raise ValueError("segment_sum_pair: rows [E, W], out_t / out_s [N, W] with one row stride")
    fn = _l.load().cartnet_segment_sum_pair_h if half else _l.load().cartnet_segment_sum_pair
    _l.check(fn(rows.data_ptr(), _ld(rows), layout.rowptr.data_ptr(), layout.colptr.data_ptr(), layout.perm.data_ptr(), N, W,
                out_t.data_ptr(), out_s.data_ptr(), _ld(out_t), int(ochunk), _l.stream_ptr()), "cartnet_segment_sum_pair")


def node_update_fwd(aggr, x_in, mean_rstd, gamma, beta, x_out) -> None:
    _f32_2d(aggr, "node_update_fwd aggr")
    N, D = aggr.shape
    for name, t in (("aggr", aggr), ("x_in", x_in), ("x_out", x_out)):
        _edge_rows(t, N, D, f"node_update_fwd {name}")
    _vec(mean_rstd, 2 * D, "mean_rstd")
    _vec(gamma, D, "gamma")
    _vec(beta, D, "beta")
    _l.check(_l.load().cartnet_node_update_fwd(aggr.data_ptr(), x_in.data_ptr(), mean_rstd.data_ptr(),
                                               gamma.data_ptr(), beta.data_ptr(), N, D, x_out.data_ptr(), None,
                                               _l.stream_ptr()), "cartnet_node_update_fwd")


def node_update_bwd_stats(aggr, dx_out, mean_rstd, gamma, beta, parts_a, parts_b) -> None:
    _f32_2d(aggr, "node_update_bwd_stats aggr")
    N, D = aggr.shape
    _edge_rows(aggr, N, D, "aggr")
    _edge_rows(dx_out, N, D, "dx_out")
    _vec(mean_rstd, 2 * D, "mean_rstd")
    _vec(gamma, D, "gamma")
    _vec(beta, D, "beta")
    npart = node_nparts(N)
    _vec(parts_a, npart * D, "parts_a", torch.float64)
    _vec(parts_b, npart * D, "parts_b", torch.float64)
    _l.check(_l.load().cartnet_node_update_bwd_stats(aggr.data_ptr(), dx_out.data_ptr(), mean_rstd.data_ptr(),
                                                     gamma.data_ptr(), beta.data_ptr(), N, D, parts_a.data_ptr(),
                                                     parts_b.data_ptr(), None, _l.stream_ptr()),
             "cartnet_node_update_bwd_stats")


def node_update_bwd_apply(aggr, dx_out, mean_rstd, gamma, beta, sums, training: bool, daggr, bc=None, parts_a=None,
                          parts_b=None) -> None:
    """``bc`` [N, 2D] + ``parts_a`` / ``parts_b`` [node_nparts(N), D] fp64: cartnet_node_update_bwd_apply_bc -- also the
    column partial sums of daggr * bc[:, :D] and daggr * bc[:, D:]."""
    _f32_2d(aggr, "node_update_bwd_apply aggr")
    N, D = aggr.shape
    _edge_rows(aggr, N, D, "aggr")
    _edge_rows(dx_out, N, D, "dx_out")
    _edge_rows(daggr, N, D, "daggr")
    _vec(mean_rstd, 2 * D, "mean_rstd")
    _vec(gamma, D, "gamma")
    _vec(beta, D, "beta")
    _vec(sums, 2 * D, "sums")
    if bc is not None:
        _edge_rows(bc, N, 2 * D, "bc")
        _vec(parts_a, node_nparts(N) * D, "parts_a", torch.float64)
        _vec(parts_b, node_nparts(N) * D, "parts_b", torch.float64)
        _l.check(_l.load().cartnet_node_update_bwd_apply_bc(aggr.data_ptr(), dx_out.data_ptr(), mean_rstd.data_ptr(),
                                                            gamma.data_ptr(), beta.data_ptr(), sums.data_ptr(),
                                                            int(training), N, D, daggr.data_ptr(), bc.data_ptr(),
                                                            parts_a.data_ptr(), parts_b.data_ptr(), _l.stream_ptr()),
                 "cartnet_node_update_bwd_apply_bc")
        return
    _l.check(_l.load().cartnet_node_update_bwd_apply(aggr.data_ptr(), dx_out.data_ptr(), mean_rstd.data_ptr(),
                                                     gamma.data_ptr(), beta.data_ptr(), sums.data_ptr(),
                                                     int(training), N, D, daggr.data_ptr(), None, _l.stream_ptr()),
             "cartnet_node_update_bwd_apply")


def mask_index(mask: Tensor, out_index: Tensor, count: Optional[Tensor]) -> None:
    N = int(mask.numel())
    if not (mask.is_cuda and mask.dtype in (torch.bool, torch.uint8) and mask.is_contiguous()):
        raise ValueError("mask_index: mask must be a contiguous bool/uint8 CUDA tensor")
    _vec(out_index, N, "mask_index out_index", torch.int32)
    if count is not None:
        _vec(count, 1, "mask_index count", torch.int32)
    _l.check(_l.load().cartnet_mask_index(mask.data_ptr(), N, out_index.data_ptr(), _l.ptr(count), _l.stream_ptr()),
             "cartnet_mask_index")


def cholesky_head_fwd(hid, out_index, W2, b2, p6, pred) -> None:
    _f32_2d(hid, "cholesky_head_fwd hid")
    N, H = hid.shape
    if not hid.is_contiguous():
        raise ValueError("cholesky_head_fwd: hid must be contiguous")
    _vec(out_index, N, "out_index", torch.int32)
    _vec(W2, 6 * H, "W2")
    _vec(b2, 6, "b2")
    M = int(pred.shape[0])
    _vec(pred, M * 9, "pred")
    _vec(p6, M * 6, "p6")
    _l.check(_l.load().cartnet_cholesky_head_fwd(hid.data_ptr(), out_index.data_ptr(), W2.data_ptr(), b2.data_ptr(),
                                                 N, H, p6.data_ptr(), pred.data_ptr(), _l.stream_ptr()),
             "cartnet_cholesky_head_fwd")


def cholesky_head_bwd(hid, out_index, W2, p6, dpred, dhid, parts) -> None:
    _f32_2d(hid, "cholesky_head_bwd hid")
    N, H = hid.shape
    _edge_rows(dhid, N, H, "cholesky_head_bwd dhid")
    _vec(out_index, N, "out_index", torch.int32)
    _vec(W2, 6 * H, "W2")
    M = int(dpred.shape[0])
    _vec(dpred, M * 9, "dpred")
    _vec(p6, M * 6, "p6")
    _vec(parts, node_nparts(N) * (7 * H + 8), "parts")
    _l.check(_l.load().cartnet_cholesky_head_bwd(hid.data_ptr(), out_index.data_ptr(), W2.data_ptr(), p6.data_ptr(),
                                                 dpred.data_ptr(), N, H, dhid.data_ptr(), parts.data_ptr(),
                                                 _l.stream_ptr()), "cartnet_cholesky_head_bwd")


def scalar_head_fwd(hid, w2, b2, graph_ptr, out) -> None:
    _f32_2d(hid, "scalar_head_fwd hid")
    N, H = hid.shape
    Bg = int(out.numel())
    _vec(w2, H, "w2")
    _vec(b2, 1, "b2")
    _vec(graph_ptr, Bg + 1, "graph_ptr", torch.int64)
    _vec(out, Bg, "out")
    _l.check(_l.load().cartnet_scalar_head_fwd(hid.data_ptr(), w2.data_ptr(), b2.data_ptr(), graph_ptr.data_ptr(), Bg,
                                               H, out.data_ptr(), _l.stream_ptr()), "cartnet_scalar_head_fwd")


def scalar_head_bwd(hid, w2, graph_ptr, batch, dout, dhid, parts) -> None:
    _f32_2d(hid, "scalar_head_bwd hid")
    N, H = hid.shape
    Bg = int(dout.numel())
    _edge_rows(dhid, N, H, "dhid")
    _vec(w2, H, "w2")
    _vec(graph_ptr, Bg + 1, "graph_ptr", torch.int64)
    _vec(batch, N, "batch", torch.int64)
    _vec(dout, Bg, "dout")
    _vec(parts, node_nparts(N) * (2 * H + 8), "parts")
    _l.check(_l.load().cartnet_scalar_head_bwd(hid.data_ptr(), w2.data_ptr(), graph_ptr.data_ptr(), batch.data_ptr(),
                                               dout.data_ptr(), N, Bg, H, dhid.data_ptr(), parts.data_ptr(),
                                               _l.stream_ptr()), "cartnet_scalar_head_bwd")


def transpose(srcs: Sequence[Tensor], outs: Optional[Sequence[Tensor]] = None) -> list:
    """Returns contiguous transposes of 2-D fp32 views (up to 40 matrices per launch); ``outs``: contiguous [cols, rows]
    tensors to refill in place."""
    given = list(outs) if outs is not None else None
    outs = []
    srcs = list(srcs)
    if given is not None and len(given) != len(srcs):
        raise ValueError("transpose: one output per matrix")
    for i in range(0, len(srcs), 40):
        chunk = srcs[i:i + 40]
        dsts = []
        for j, t in enumerate(chunk):
            _f32_2d(t, "transpose src")
            if given is not None:
                d = given[i + j]
                if not (d.is_cuda and d.dtype == torch.float32 and d.is_contiguous() and
                        tuple(d.shape) == (t.shape[1], t.shape[0])):
                    raise ValueError("transpose outs: expected a contiguous fp32 CUDA tensor of the transposed shape")
                dsts.append(d)
            else:
                dsts.append(torch.empty((t.shape[1], t.shape[0]), dtype=torch.float32, device=t.device))
        n = len(chunk)
        I32 = C.c_int32 * n
        _l.check(_l.load().cartnet_transpose(
            _ptr_array(chunk), _ptr_array(dsts), I32(*[int(t.shape[0]) for t in chunk]),
            I32(*[int(t.shape[1]) for t in chunk]), I32(*[_ld(t) for t in chunk]), I32(*[int(t.shape[0]) for t in chunk]),
            n, _l.stream_ptr()), "cartnet_transpose")
        outs.extend(dsts)
    return outs


def adam_step(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step, grad_scale=1.0) -> None:
    n = int(param.numel())
    for name, t in (("param", param), ("grad", grad), ("exp_avg", exp_avg), ("exp_avg_sq", exp_avg_sq)):
        _vec(t, n, f"adam_step {name}")
    _l.check(_l.load().cartnet_adam_step(param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(),
                                         n, float(lr), float(beta1), float(beta2), float(eps), int(step),
                                         float(grad_scale), _l.stream_ptr()), "cartnet_adam_step")


# --------------------------------------------------------------------------------------------------- iComformer pieces
class SegmentLayout:
    """Minimal stand-in for GraphLayout where only (rowptr, N, E) matter: S segments of rows given by ``ptr``."""

    def __init__(self, ptr: Tensor, n_rows: int):
        _vec(ptr, ptr.numel(), "SegmentLayout ptr", torch.int32)
        self.rowptr = ptr
        self.N = int(ptr.numel()) - 1
        self.E = int(n_rows)


def rbf_expand(v: Tensor, centers: Tensor, gamma: float, out: Tensor) -> None:
    n, bins = int(v.numel()), int(centers.numel())
    _vec(v, n, "rbf_expand v")
    _vec(centers, bins, "rbf_expand centers")
    _f32_2d(out, "rbf_expand out")
    if tuple(out.shape) != (n, bins):
        raise ValueError("rbf_expand: out must be [n, bins]")
    _l.check(_l.load().cartnet_rbf_expand(v.data_ptr(), n, centers.data_ptr(), bins, float(gamma), out.data_ptr(),
                                          _ld(out), _l.stream_ptr()), "cartnet_rbf_expand")


def lattice_features(cell: Tensor, batch: Tensor, src32: Tensor, cart_dist: Tensor, cart_dir: Tensor, edge_feat: Tensor,
                     nei_len: Tensor, nei_cos: Tensor) -> None:
    E, Bg = int(cart_dist.numel()), int(cell.shape[0])
    _vec(cell, Bg * 9, "lattice_features cell")
    _vec(batch, 1, "lattice_features batch", torch.int64)
    _vec(src32, E, "lattice_features src", torch.int32)
    _vec(cart_dist, E, "cart_dist")
    _vec(cart_dir, 3 * E, "cart_dir")
    _vec(edge_feat, E, "edge_feat")
    _vec(nei_len, 3 * Bg, "nei_len")
    _vec(nei_cos, 3 * E, "nei_cos")
    _l.check(_l.load().cartnet_lattice_features(cell.data_ptr(), batch.data_ptr(), src32.data_ptr(), cart_dist.data_ptr(),
                                                cart_dir.data_ptr(), E, Bg, edge_feat.data_ptr(), nei_len.data_ptr(),
                                                nei_cos.data_ptr(), _l.stream_ptr()), "cartnet_lattice_features")


def eltwise(op: int, a: Tensor, b: Optional[Tensor], out: Tensor, scale: float = 1.0) -> None:
    """op 0: out = softplus(a); 1: out = a*sigmoid(b); 2: out = a+b; 3: out = a*scale (2-D views)."""
    _f32_2d(a, "eltwise a")
    _f32_2d(out, "eltwise out")
    if tuple(a.shape) != tuple(out.shape):
        raise ValueError("eltwise: shape mismatch")
    if b is not None:
        _f32_2d(b, "eltwise b")
        if tuple(b.shape) != tuple(a.shape):
            raise ValueError("eltwise: shape mismatch")
    rows, cols = a.shape
    _l.check(_l.load().cartnet_eltwise(int(op), a.data_ptr(), _l.ptr(b), out.data_ptr(), rows, cols, _ld(a),
                                       _ld(b) if b is not None else 0, _ld(out), float(scale), _l.stream_ptr()),
             "cartnet_eltwise")


def segment_nparts(S: int) -> int:
    return int(_l.load().cartnet_segment_nparts(int(S)))


def rowmul_fwd(key: Tensor, q: Tensor, ptr_: Tensor, scale: float, alpha: Optional[Tensor], parts_sum: Tensor,
               parts_sq: Tensor) -> None:
    """``alpha`` None: the bn_att statistics only (alpha itself is recomputed by att_gate_fwd / att_gate_bwd_apply)."""
    _f32_2d(key, "rowmul_fwd key")
    _f32_2d(q, "rowmul_fwd q")
    R, Cc = key.shape
    S = int(q.shape[0])
    if alpha is not None:
        _f32_2d(alpha, "rowmul_fwd alpha")
    if (alpha is not None and tuple(alpha.shape) != (R, Cc)) or q.shape[1] != Cc:
        raise ValueError("rowmul_fwd: shape mismatch")
    _vec(ptr_, S + 1, "rowmul_fwd ptr", torch.int32)
    npart = segment_nparts(S)
    _vec(parts_sum, npart * Cc, "rowmul_fwd parts_sum", torch.float64)
    _vec(parts_sq, npart * Cc, "rowmul_fwd parts_sq", torch.float64)
    _l.check(_l.load().cartnet_rowmul_fwd(key.data_ptr(), _ld(key), q.data_ptr(), _ld(q), ptr_.data_ptr(), S, Cc,
                                          float(scale), _l.ptr(alpha), _ld(alpha) if alpha is not None else Cc,
                                          parts_sum.data_ptr(), parts_sq.data_ptr(), _l.stream_ptr()), "cartnet_rowmul_fwd")


def rowmul_bwd(dalpha: Tensor, key: Tensor, q: Tensor, ptr_: Tensor, scale: float, dq: Tensor,
               sum_dkey: Optional[Tensor] = None, sum_dq: Optional[Tensor] = None) -> None:
    """dalpha <- dkey = dalpha * q[s] * scale in place, dq[s] = scale * sum_r dalpha[r] * key[r]; with ``sum_dkey`` /
    ``sum_dq`` ([C] each, C <= 256) also the column sums of dkey and dq from the same pass (cartnet_rowmul_bwd_sums)."""
    _f32_2d(dalpha, "rowmul_bwd dalpha")
    _f32_2d(key, "rowmul_bwd key")
    _f32_2d(q, "rowmul_bwd q")
    _f32_2d(dq, "rowmul_bwd dq")
    R, Cc = key.shape
    S = int(q.shape[0])
    if tuple(dalpha.shape) != (R, Cc) or tuple(dq.shape) != (S, Cc) or q.shape[1] != Cc:
        raise ValueError("rowmul_bwd: shape mismatch")
    _vec(ptr_, S + 1, "rowmul_bwd ptr", torch.int32)
    if (sum_dkey is None) != (sum_dq is None):
        raise ValueError("rowmul_bwd: sum_dkey and sum_dq come together")
    if sum_dkey is not None:
        _vec(sum_dkey, Cc, "rowmul_bwd sum_dkey")
        _vec(sum_dq, Cc, "rowmul_bwd sum_dq")
        npart = segment_nparts(S)
        pk, pq = (torch.empty(npart * Cc, dtype=torch.float64, device=key.device) for _ in range(2))
        _l.check(_l.load().cartnet_rowmul_bwd_sums(dalpha.data_ptr(), _ld(dalpha), key.data_ptr(), _ld(key), q.data_ptr(),
                                                   _ld(q), ptr_.data_ptr(), S, Cc, float(scale), dq.data_ptr(), _ld(dq),
                                                   pk.data_ptr(), pq.data_ptr(), _l.stream_ptr()),
                 "cartnet_rowmul_bwd_sums")
        colsum_finalize([pk, pq], npart, [sum_dkey, sum_dq])
        return
    _l.check(_l.load().cartnet_rowmul_bwd(dalpha.data_ptr(), _ld(dalpha), key.data_ptr(), _ld(key), q.data_ptr(), _ld(q),
                                          ptr_.data_ptr(), S, Cc, float(scale), dq.data_ptr(), _ld(dq),
                                          _l.stream_ptr()), "cartnet_rowmul_bwd")


def att_gate_fwd(gs: Tensor, q: Tensor, ptr_: Tensor, mean_rstd: Tensor, gamma: Tensor, beta: Tensor, scale: float,
                 aggr: Tensor, bc: Optional[Tensor] = None) -> None:
    """cartnet_att_gate_fwd: gs = [key | msg] [R, 2D]; aggr[s] = sum_r sigmoid(bn(key q[s] scale)) msg (+ bc [S, 2D])."""
    _f32_2d(gs, "att_gate_fwd gs")
    _f32_2d(q, "att_gate_fwd q")
    S, D = aggr.shape
    if gs.shape[1] != 2 * D or tuple(q.shape) != (S, D):
        raise ValueError("att_gate_fwd: shape mismatch")
    _edge_rows(aggr, S, D, "att_gate_fwd aggr")
    _vec(ptr_, S + 1, "att_gate_fwd ptr", torch.int32)
    _vec(mean_rstd, 2 * D, "mean_rstd")
    _vec(gamma, D, "gamma")
    _vec(beta, D, "beta")
    if bc is not None:
        _edge_rows(bc, S, 2 * D, "att_gate_fwd bc")
    if _ld(gs) != 2 * D:
        raise ValueError("att_gate_fwd: gs must be a contiguous [R, 2D] matrix")
    _l.check(_l.load().cartnet_att_gate_fwd(gs.data_ptr(), q.data_ptr(), _ld(q), ptr_.data_ptr(), mean_rstd.data_ptr(),
                                            gamma.data_ptr(), beta.data_ptr(), float(scale), S, D, aggr.data_ptr(),
                                            _l.ptr(bc), _l.stream_ptr()), "cartnet_att_gate_fwd")


def att_gate_bwd_apply(gs: Tensor, key: Optional[Tensor], q: Tensor, daggr: Tensor, ptr_: Tensor, mean_rstd: Tensor, gamma: Tensor,
                       beta: Tensor, sums: Tensor, count: int, training: bool, scale: float, dq: Tensor,
                       sum_dkey: Tensor, sum_dmsg: Tensor, sum_dq: Tensor) -> None:
    """cartnet_att_gate_bwd_apply + finaliser: gs = [alpha | msg] -> [dkey | dmsg] in place, dq, and the column sums of
    dkey / dmsg / dq (iComformer's attention block backward in one pass; comformer_conv.py:90-99).  ``key`` None: gs holds
    [key | msg] (the forward pass went through att_gate_fwd; alpha is recomputed)."""
    _f32_2d(gs, "att_gate_bwd_apply gs")
    if key is not None:
        _f32_2d(key, "att_gate_bwd_apply key")
    _f32_2d(q, "att_gate_bwd_apply q")
    _f32_2d(dq, "att_gate_bwd_apply dq")
    R, D = int(gs.shape[0]), int(gs.shape[1]) // 2
    S = int(q.shape[0])
    if (key is not None and tuple(key.shape) != (R, D)) or _ld(gs) != 2 * D:
        raise ValueError("att_gate_bwd_apply: key / gs shape mismatch")
    if tuple(gs.shape) != (R, 2 * D) or q.shape[1] != D or tuple(dq.shape) != (S, D):
        raise ValueError("att_gate_bwd_apply: shape mismatch")
    _edge_rows(daggr, S, D, "att_gate_bwd_apply daggr")
    _vec(ptr_, S + 1, "att_gate_bwd_apply ptr", torch.int32)
    _vec(mean_rstd, 2 * D, "mean_rstd")
    _vec(gamma, D, "gamma")
    _vec(beta, D, "beta")
    _vec(sums, 2 * D, "sums")
    for t in (sum_dkey, sum_dmsg, sum_dq):
        _vec(t, D, "att_gate_bwd_apply sums out")
    npart = segment_nparts(S)
    pk, pm, pq = (torch.empty(npart * D, dtype=torch.float64, device=gs.device) for _ in range(3))
    _l.check(_l.load().cartnet_att_gate_bwd_apply(
        gs.data_ptr(), _l.ptr(key), _ld(key) if key is not None else D, q.data_ptr(), _ld(q), daggr.data_ptr(), ptr_.data_ptr(),
        mean_rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), sums.data_ptr(), int(count), int(training), float(scale),
        S, D, dq.data_ptr(), _ld(dq), pk.data_ptr(), pm.data_ptr(), pq.data_ptr(), _l.stream_ptr()),
        "cartnet_att_gate_bwd_apply")
    colsum_finalize([pk, pm, pq], npart, [sum_dkey, sum_dmsg, sum_dq])


def softplus_update_fwd(o, x, mean_rstd, gamma, beta, y) -> None:
    _f32_2d(o, "softplus_update_fwd o")
    N, D = o.shape
    for name, t in (("o", o), ("x", x), ("y", y)):
        _edge_rows(t, N, D, f"softplus_update_fwd {name}")
    _vec(mean_rstd, 2 * D, "mean_rstd")
    _vec(gamma, D, "gamma")
    _vec(beta, D, "beta")
    _l.check(_l.load().cartnet_softplus_update_fwd(o.data_ptr(), x.data_ptr(), mean_rstd.data_ptr(), gamma.data_ptr(),
                                                   beta.data_ptr(), N, D, y.data_ptr(), _l.stream_ptr()),
             "cartnet_softplus_update_fwd")


def softplus_update_bwd_stats(o, x, dy, mean_rstd, gamma, beta, parts_a, parts_b) -> None:
    _f32_2d(o, "softplus_update_bwd_stats o")
    N, D = o.shape
    for name, t in (("o", o), ("x", x), ("dy", dy)):
        _edge_rows(t, N, D, f"softplus_update_bwd_stats {name}")
    _vec(mean_rstd, 2 * D, "mean_rstd")
    _vec(gamma, D, "gamma")
    _vec(beta, D, "beta")
    npart = segment_nparts(N)
    _vec(parts_a, npart * D, "parts_a", torch.float64)
    _vec(parts_b, npart * D, "parts_b", torch.float64)
    _l.check(_l.load().cartnet_softplus_update_bwd_stats(o.data_ptr(), x.data_ptr(), dy.data_ptr(), mean_rstd.data_ptr(),
                                                         gamma.data_ptr(), beta.data_ptr(), N, D, parts_a.data_ptr(),
                                                         parts_b.data_ptr(), _l.stream_ptr()),
             "cartnet_softplus_update_bwd_stats")


def softplus_update_bwd_apply(o, x, dy, mean_rstd, gamma, beta, sums, training: bool, d_o, dx_add, dx,
                              sum_do: Optional[Tensor] = None) -> None:
    """``sum_do`` [D]: also the column sums of d_o from the same pass (cartnet_softplus_update_bwd_apply_sums)."""
    _f32_2d(o, "softplus_update_bwd_apply o")
    N, D = o.shape
    for name, t in (("o", o), ("x", x), ("dy", dy), ("d_o", d_o), ("dx", dx)):
        _edge_rows(t, N, D, f"softplus_update_bwd_apply {name}")
    if dx_add is not None:
        _edge_rows(dx_add, N, D, "softplus_update_bwd_apply dx_add")
    _vec(mean_rstd, 2 * D, "mean_rstd")
    _vec(gamma, D, "gamma")
    _vec(beta, D, "beta")
    _vec(sums, 2 * D, "sums")
    if sum_do is not None:
        _vec(sum_do, D, "softplus_update_bwd_apply sum_do")
        npart = segment_nparts(N)
        pd = torch.empty(npart * D, dtype=torch.float64, device=o.device)
        _l.check(_l.load().cartnet_softplus_update_bwd_apply_sums(o.data_ptr(), x.data_ptr(), dy.data_ptr(),
                                                                  mean_rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                                                  sums.data_ptr(), int(training), N, D, d_o.data_ptr(),
                                                                  _l.ptr(dx_add), dx.data_ptr(), pd.data_ptr(),
                                                                  _l.stream_ptr()),
                 "cartnet_softplus_update_bwd_apply_sums")
        colsum_finalize([pd], npart, [sum_do])
        return
    _l.check(_l.load().cartnet_softplus_update_bwd_apply(o.data_ptr(), x.data_ptr(), dy.data_ptr(), mean_rstd.data_ptr(),
                                                         gamma.data_ptr(), beta.data_ptr(), sums.data_ptr(),
                                                         int(training), N, D, d_o.data_ptr(), _l.ptr(dx_add),
                                                         dx.data_ptr(), _l.stream_ptr()),
             "cartnet_softplus_update_bwd_apply")


def softplus_bwd_sums(a: Tensor, b: Tensor, out: Tensor, sum_out: Tensor) -> None:
    """out = a * sigmoid(b) (the backward of out = softplus(b)) and sum_out[c] = sum_r out[r, c] from the same pass
    (cartnet_softplus_bwd_sums + finaliser): the RBF branches of iComformer, comformer.py:93-105."""
    _f32_2d(a, "softplus_bwd_sums a")
    R, Cc = a.shape
    _f32_2d(b, "softplus_bwd_sums b")
    _f32_2d(out, "softplus_bwd_sums out")
    if tuple(b.shape) != (R, Cc) or tuple(out.shape) != (R, Cc):
        raise ValueError("softplus_bwd_sums: shape mismatch")
    _vec(sum_out, Cc, "softplus_bwd_sums sum_out")
    npart = segment_nparts(R)
    ps = torch.empty(npart * Cc, dtype=torch.float64, device=a.device)
    _l.check(_l.load().cartnet_softplus_bwd_sums(a.data_ptr(), _ld(a), b.data_ptr(), _ld(b), out.data_ptr(), _ld(out), R, Cc,
                                                 ps.data_ptr(), _l.stream_ptr()), "cartnet_softplus_bwd_sums")
    colsum_finalize([ps], npart, [sum_out])


def coldot_bc(d: Tensor, bc: Tensor, out_a: Tensor, out_b: Tensor) -> None:
    """out_a[c] = sum_r d[r, c] * bc[r, c], out_b[c] = sum_r d[r, c] * bc[r, C + c]  (cartnet_coldot_bc_partial + finaliser):
    the targets' share of the gate's BatchNorm-backward sums from the per-target sums of gate_scatter_fwd(..., bc=...)."""
    _f32_2d(d, "coldot_bc d")
    R, Cc = d.shape
    _edge_rows(bc, R, 2 * Cc, "coldot_bc bc")
    _vec(out_a, Cc, "coldot_bc out_a")
    _vec(out_b, Cc, "coldot_bc out_b")
    npart = segment_nparts(R)
    pa, pb = (torch.empty(npart * Cc, dtype=torch.float64, device=d.device) for _ in range(2))
    _l.check(_l.load().cartnet_coldot_bc_partial(d.data_ptr(), _ld(d), bc.data_ptr(), R, Cc, pa.data_ptr(), pb.data_ptr(),
                                                 _l.stream_ptr()), "cartnet_coldot_bc_partial")
    colsum_finalize([pa, pb], npart, [out_a, out_b])


def colsum(x: Tensor, out: Tensor) -> None:
    """out[c] = sum_r x[r, c] (fp64 partials, fixed order) for a 2-D fp32 view."""
    _f32_2d(x, "colsum x")
    R, Cc = x.shape
    _vec(out, Cc, "colsum out")
    npart = segment_nparts(R)
    parts = torch.empty(npart * Cc, dtype=torch.float64, device=x.device)
    _l.check(_l.load().cartnet_colsum_partial(x.data_ptr(), _ld(x), R, Cc, parts.data_ptr(), _l.stream_ptr()),
             "cartnet_colsum_partial")
    colsum_finalize(parts, npart, out)


# ------------------------------------------------------------------------------------------------ eComformer (equi)
EQUI_NS, EQUI_H1, EQUI_NW = 64, 128, 5120


def _equi_common(feat: Tensor, width: int, w: Tensor, cart_dir: Tensor, layout: "GraphLayout", name: str):
    _f32_2d(feat, f"{name} node features")
    N = int(feat.shape[0])
    if feat.shape[1] != width or not feat.is_contiguous():
        raise ValueError(f"{name}: node features must be contiguous [N, {width}]")
    E = int(layout.E)
    if tuple(w.shape) != (E, EQUI_NW) or w.dtype != torch.float32 or not w.is_cuda or not w.is_contiguous():
        raise ValueError(f"{name}: w must be a contiguous fp32 CUDA tensor [E={E}, {EQUI_NW}]")
    if tuple(cart_dir.shape) != (E, 3) or cart_dir.dtype != torch.float32 or not cart_dir.is_contiguous():
        raise ValueError(f"{name}: cart_dir must be contiguous fp32 [E, 3]")
    if layout.N != N or layout.colptr is None:
        raise ValueError(f"{name}: graph layout (with the by-source permutation) does not match the node count")
    return N, E


def _equi_out(t: Tensor, shape, name: str):
    if tuple(t.shape) != tuple(shape) or t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
        raise ValueError(f"{name}: expected a contiguous fp32 CUDA tensor {tuple(shape)}")


def equi_tp1_fwd(x0, w, cart_dir, layout: "GraphLayout", h1) -> None:
    N, _ = _equi_common(x0, EQUI_NS, w, cart_dir, layout, "equi_tp1_fwd")
    _equi_out(h1, (N, EQUI_H1), "equi_tp1_fwd h1")
    _l.check(_l.load().cartnet_equi_tp1_fwd(x0.data_ptr(), w.data_ptr(), cart_dir.data_ptr(), layout.colptr.data_ptr(),
                                            layout.perm.data_ptr(), layout.tgt.data_ptr(), N, h1.data_ptr(),
                                            _l.stream_ptr()), "cartnet_equi_tp1_fwd")


def equi_tp1_bwd(x0, w, cart_dir, layout: "GraphLayout", dh1, dw, dxe) -> None:
    N, E = _equi_common(x0, EQUI_NS, w, cart_dir, layout, "equi_tp1_bwd")
    _equi_out(dh1, (N, EQUI_H1), "equi_tp1_bwd dh1")
    _equi_out(dw, (E, EQUI_NW), "equi_tp1_bwd dw")
    _equi_out(dxe, (E, EQUI_NS), "equi_tp1_bwd dxe")
    _l.check(_l.load().cartnet_equi_tp1_bwd(x0.data_ptr(), w.data_ptr(), cart_dir.data_ptr(), layout.colptr.data_ptr(),
                                            layout.perm.data_ptr(), layout.tgt.data_ptr(), dh1.data_ptr(), N,
                                            dw.data_ptr(), dxe.data_ptr(), _l.stream_ptr()), "cartnet_equi_tp1_bwd")


def equi_tp2_fwd(h1, w, cart_dir, layout: "GraphLayout", o2) -> None:
    N, _ = _equi_common(h1, EQUI_H1, w, cart_dir, layout, "equi_tp2_fwd")
    _equi_out(o2, (N, EQUI_NS), "equi_tp2_fwd o2")
    _l.check(_l.load().cartnet_equi_tp2_fwd(h1.data_ptr(), w.data_ptr(), cart_dir.data_ptr(), layout.colptr.data_ptr(),
                                            layout.perm.data_ptr(), layout.tgt.data_ptr(), N, o2.data_ptr(),
                                            _l.stream_ptr()), "cartnet_equi_tp2_fwd")


def equi_tp2_bwd(h1, w, cart_dir, layout: "GraphLayout", do2, dw, dhe) -> None:
    N, E = _equi_common(h1, EQUI_H1, w, cart_dir, layout, "equi_tp2_bwd")
    _equi_out(do2, (N, EQUI_NS), "equi_tp2_bwd do2")
    _equi_out(dw, (E, EQUI_NW), "equi_tp2_bwd dw")
    _equi_out(dhe, (E, EQUI_H1), "equi_tp2_bwd dhe")
    _l.check(_l.load().cartnet_equi_tp2_bwd(h1.data_ptr(), w.data_ptr(), cart_dir.data_ptr(), layout.colptr.data_ptr(),
                                            layout.perm.data_ptr(), layout.tgt.data_ptr(), do2.data_ptr(), N,
                                            dw.data_ptr(), dhe.data_ptr(), _l.stream_ptr()), "cartnet_equi_tp2_bwd")


def colstats_nparts(R: int) -> int:
    return int(_l.load().cartnet_colstats_nparts(int(R)))


def colstats_partial(x: Tensor, parts_sum: Tensor, parts_sq: Tensor) -> None:
    """fp64 partial column sums / sums of squares of x [R, C] (rows of parts: colstats_nparts(R))."""
    _f32_2d(x, "colstats_partial x")
    R, Cc = x.shape
    n = colstats_nparts(R) * Cc
    _vec(parts_sum, n, "colstats parts_sum", torch.float64)
    _vec(parts_sq, n, "colstats parts_sq", torch.float64)
    _l.check(_l.load().cartnet_colstats_partial(x.data_ptr(), _ld(x), R, Cc, parts_sum.data_ptr(), parts_sq.data_ptr(),
                                                _l.stream_ptr()), "cartnet_colstats_partial")
