"""ctypes binding of libcartnet_hip.so (the C ABI declared in include/cartnet_hip.h).

The product path has no CPU or PyTorch fallback: if the shared library is missing or a call fails, this module
raises.  Tensors cross the boundary as raw device pointers + sizes; the HIP stream is torch's current stream.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# CARTNET_LIB (tools only): a diagnostic / A-B build of the same ABI next to the product library (tools/build_variant.sh)
LIB_PATH = os.environ.get("CARTNET_LIB") or os.path.join(_HERE, "libcartnet_hip.so")
MAX_GROUPS = 4
ABI_VERSION = 12         # cartnet_abi_version() of the library this binding mirrors (include/cartnet_hip.h)

_lib: Optional[C.CDLL] = None

c_f32p = C.c_void_p
c_i32p = C.c_void_p
c_i64p = C.c_void_p
c_u8p = C.c_void_p
c_stream = C.c_void_p
c_groups = C.c_void_p       # const CartnetGroups* (None = one BatchNorm group: the whole batch)


class GateGemmArgs(C.Structure):
    """CartnetGateGemmArgs (include/cartnet_hip.h)."""
    _fields_ = [("pre", C.c_void_p), ("ldp", C.c_int32), ("img_gate", C.c_void_p), ("img_aggr", C.c_void_p),
                ("bias_gate", C.c_void_p), ("bias_aggr", C.c_void_p), ("mean_rstd", C.c_void_p), ("gamma", C.c_void_p),
                ("beta", C.c_void_p), ("env", C.c_void_p), ("e_in", C.c_void_p), ("e_out", C.c_void_p),
                ("tgt", C.c_void_p), ("rowptr", C.c_void_p), ("aggr", C.c_void_p), ("bnd", C.c_void_p),
                ("E", C.c_int64), ("N", C.c_int32), ("D", C.c_int32)]


class GemmArgs(C.Structure):
    _fields_ = [
        ("A", C.c_void_p * MAX_GROUPS), ("B", C.c_void_p * MAX_GROUPS), ("C", C.c_void_p * MAX_GROUPS),
        ("cpre", C.c_void_p * MAX_GROUPS), ("bias", C.c_void_p * MAX_GROUPS),
        ("gather_i", C.c_void_p * MAX_GROUPS), ("gather_j", C.c_void_p * MAX_GROUPS),
        ("resid", C.c_void_p * MAX_GROUPS), ("dact", C.c_void_p * MAX_GROUPS),
        ("colsum", C.c_void_p * MAX_GROUPS), ("colsq", C.c_void_p * MAX_GROUPS),
        ("tgt", C.c_void_p), ("src", C.c_void_p),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("lda", C.c_int32), ("ldb", C.c_int32), ("ldc", C.c_int32), ("ldg", C.c_int32), ("ldr", C.c_int32),
        ("ldd", C.c_int32),
        ("ngroups", C.c_int32), ("nsegs", C.c_int32), ("splitk", C.c_int32),
        ("a_kstrided", C.c_int32), ("b_kstrided", C.c_int32), ("a_act", C.c_int32), ("b_act", C.c_int32),
        ("out_act", C.c_int32), ("precision", C.c_int32),
        ("b_split", C.c_void_p * MAX_GROUPS), ("b_split_folded", C.c_void_p),
        ("a_act_out", C.c_void_p * MAX_GROUPS),
        ("a_half", C.c_int32), ("b_half", C.c_int32), ("c_half", C.c_int32), ("dact_half", C.c_int32),
        ("gst_g", C.c_void_p), ("gst_env", C.c_void_p), ("gst_mean_rstd", C.c_void_p), ("gst_gamma", C.c_void_p),
        ("gst_beta", C.c_void_p), ("gst_ld", C.c_int32),
        ("tile_policy", C.c_int32),
        ("dact_kind", C.c_int32),
        ("gather_rows", C.c_int32),
    ]


MAX_LAYERS = 16


class GemmProfile(C.Structure):
    _fields_ = [("variant", C.c_int32), ("launches", C.c_int64), ("flops", C.c_double), ("ms", C.c_double)]

_LAYER_PARAM_FIELDS = ["gate0_w", "gate0_b", "gate2_w", "gate2_b", "aggr0_w", "aggr0_b", "aggr2_w", "aggr2_b",
                       "norm_w", "norm_b", "norm2_w", "norm2_b"]
# CartnetLayerParams field -> suffix of the reference state_dict key under "layers.{l}."
LAYER_PARAM_KEYS = {"gate0_w": "MLP_gate.0.weight", "gate0_b": "MLP_gate.0.bias", "gate2_w": "MLP_gate.2.weight",
                    "gate2_b": "MLP_gate.2.bias", "aggr0_w": "MLP_aggr.0.weight", "aggr0_b": "MLP_aggr.0.bias",
                    "aggr2_w": "MLP_aggr.2.weight", "aggr2_b": "MLP_aggr.2.bias", "norm_w": "norm.weight",
                    "norm_b": "norm.bias", "norm2_w": "norm2.weight", "norm2_b": "norm2.bias"}
# CartnetParams field -> reference state_dict key
PARAM_KEYS = {"embedding": "encoder.embedding.weight", "temp_w": "encoder.temperature_proj_atom.weight",
              "temp_b": "encoder.temperature_proj_atom.bias", "enc_bias": "encoder.bias",
              "atom_w": "encoder.encoder_atom.1.weight", "atom_b": "encoder.encoder_atom.1.bias",
              "edge0_w": "encoder.encoder_edge.0.weight", "edge0_b": "encoder.encoder_edge.0.bias",
              "edge2_w": "encoder.encoder_edge.2.weight", "edge2_b": "encoder.encoder_edge.2.bias",
              "head0_w": "head.MLP.0.weight", "head0_b": "head.MLP.0.bias", "head2_w": "head.MLP.2.weight",
              "head2_b": "head.MLP.2.bias"}


class LayerParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in _LAYER_PARAM_FIELDS]


class LayerBuffers(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("norm_mean", "norm_var", "norm_nbt", "norm2_mean", "norm2_var", "norm2_nbt")]


class Params(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("embedding", "temp_w", "temp_b", "enc_bias", "atom_w", "atom_b", "edge0_w",
                                          "edge0_b", "edge2_w", "edge2_b")] + \
               [("layer", LayerParams * MAX_LAYERS)] + \
               [(n, C.c_void_p) for n in ("head0_w", "head0_b", "head2_w", "head2_b")]


# CartnetAllReduceFn: int (*)(void* user, double* buf, int64_t count, void* stream)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
GRADREADY_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_void_p)       # CartnetGradReadyFn(user, bucket, stream)


class Model(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("D", "R", "L", "invariant", "use_temperature", "atom_types", "cholesky",
                                         "n_types")] + \
               [("use_envelope", C.c_int32 * MAX_LAYERS)] + \
               [(n, C.c_float) for n in ("radius", "env_radius", "bn_eps", "bn_momentum")] + \
               [("gemm_precision", C.c_int32), ("bn_group_size", C.c_int32)] + \
               [("rbf_means", C.c_void_p), ("rbf_betas", C.c_void_p), ("p", Params), ("buf", LayerBuffers * MAX_LAYERS),
                ("bn_allreduce", ALLREDUCE_FN), ("bn_allreduce_user", C.c_void_p), ("half_storage", C.c_int32),
                ("grad_ready", GRADREADY_FN), ("grad_ready_user", C.c_void_p)]


_ICF_CONV_FIELDS = ("query_w", "query_b", "key_w", "key_b", "value_w", "value_b", "edge_w", "edge_b", "concate_w",
                    "concate_b", "key0_w", "key0_b", "key2_w", "key2_b", "msg0_w", "msg0_b", "msg2_w", "msg2_b", "bn_w",
                    "bn_b", "bn_att_w", "bn_att_b")
# CartnetIcfConv field -> parameter name below "att_layers.{l}." / "edge_update_layer."
ICF_CONV_KEYS = {"query_w": "lin_query.weight", "query_b": "lin_query.bias", "key_w": "lin_key.weight", "key_b": "lin_key.bias",
                 "value_w": "lin_value.weight", "value_b": "lin_value.bias", "edge_w": "lin_edge.weight",
                 "edge_b": "lin_edge.bias", "concate_w": "lin_concate.weight", "concate_b": "lin_concate.bias",
                 "key0_w": "key_update.0.weight", "key0_b": "key_update.0.bias", "key2_w": "key_update.2.weight",
                 "key2_b": "key_update.2.bias", "msg0_w": "lin_msg_update.0.weight", "msg0_b": "lin_msg_update.0.bias",
                 "msg2_w": "lin_msg_update.2.weight", "msg2_b": "lin_msg_update.2.bias", "bn_w": "bn.weight",
                 "bn_b": "bn.bias", "bn_att_w": "bn_att.weight", "bn_att_b": "bn_att.bias"}
ICF_TOP_KEYS = {"embedding": "embedding.weight", "temp_w": "temperature_proj_atom.weight",
                "temp_b": "temperature_proj_atom.bias", "rbf_w": "rbf.1.weight", "rbf_b": "rbf.1.bias",
                "rbf_angle_w": "rbf_angle.1.weight", "rbf_angle_b": "rbf_angle.1.bias", "head0_w": "cholesky.MLP.0.weight",
                "head0_b": "cholesky.MLP.0.bias", "head2_w": "cholesky.MLP.2.weight", "head2_b": "cholesky.MLP.2.bias"}


class IcfConv(C.Structure):
    """CartnetIcfConv (include/cartnet_hip.h)."""
    _fields_ = [(n, C.c_void_p) for n in _ICF_CONV_FIELDS] + \
               [(n, C.c_void_p * 3) for n in ("key_e_w", "key_e_b", "value_e_w", "value_e_b")]


class IcfBn(C.Structure):
    _fields_ = [("mean", C.c_void_p), ("var", C.c_void_p), ("nbt", C.c_void_p)]


class IcfParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("embedding", "temp_w", "temp_b", "rbf_w", "rbf_b", "rbf_angle_w", "rbf_angle_b")] + \
               [("att", IcfConv * 4), ("edge", IcfConv)] + \
               [(n, C.c_void_p) for n in ("head0_w", "head0_b", "head2_w", "head2_b")]


class IcfModel(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("C", "n_types", "gemm_precision", "reserved")] + \
               [(n, C.c_float) for n in ("gamma_rbf", "gamma_angle", "bn_eps", "bn_momentum")] + \
               [("rbf_centers", C.c_void_p), ("rbf_angle_centers", C.c_void_p), ("p", IcfParams),
                ("att_bn", IcfBn * 4), ("att_bn_att", IcfBn * 4), ("edge_bn", IcfBn), ("edge_bn_att", IcfBn)]


class Groups(C.Structure):
    """CartnetGroups: BatchNorm groups inside one batch (include/cartnet_hip.h)."""
    _fields_ = [("node_gptr", C.c_void_p), ("edge_gptr", C.c_void_p), ("G", C.c_int32), ("edge_parts", C.c_int32),
                ("node_parts", C.c_int32)]


class BatchDesc(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("z", "batch", "graph_ptr", "edge_index", "temperature", "cart_dist",
                                          "cart_dir", "non_h_mask")] + \
               [("N", C.c_int32), ("Bg", C.c_int32), ("M", C.c_int32), ("E", C.c_int64)]


class Shard(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("atom_ptr", "edge_ptr", "y_ptr", "z", "pos", "non_h_mask", "edge_src",
                                          "edge_tgt", "cart_dist", "cart_dir", "cell", "temperature", "y")] + \
               [("y_width", C.c_int32)]


class Collated(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("x", "pos", "non_h_mask", "batch", "ptr", "edge_index", "cart_dist",
                                          "cart_dir", "cell", "temperature", "y")]


# name -> (restype, argtypes); every symbol include/cartnet_hip.h declares
PROTOTYPES = {
    "cartnet_last_error": (C.c_char_p, []),
    "cartnet_abi_version": (C.c_int, []),
    "cartnet_abi_struct_sizes": (C.c_int, [C.POINTER(C.c_size_t), C.c_int32]),
    "cartnet_gemm": (C.c_int, [C.POINTER(GemmArgs), c_stream]),
    "cartnet_gate_gemm_eval_workspace": (C.c_size_t, [C.c_int64, C.c_int32]),
    "cartnet_gate_gemm_eval": (C.c_int, [C.POINTER(GateGemmArgs), c_stream]),
    "cartnet_gemm_split_b_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "cartnet_gemm_split_b": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int32),
                                       C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32,
                                       c_stream]),
    "cartnet_gemm_pack_b_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "cartnet_gemm_pack_b": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int32),
                                      C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32,
                                      c_stream]),
    "cartnet_equi_tp1_fwd": (C.c_int, [c_f32p, c_f32p, c_f32p, c_i32p, c_i32p, c_i32p, C.c_int32, c_f32p, c_stream]),
    "cartnet_equi_tp1_bwd": (C.c_int, [c_f32p, c_f32p, c_f32p, c_i32p, c_i32p, c_i32p, c_f32p, C.c_int32, c_f32p, c_f32p,
                                       c_stream]),
    "cartnet_equi_tp2_fwd": (C.c_int, [c_f32p, c_f32p, c_f32p, c_i32p, c_i32p, c_i32p, C.c_int32, c_f32p, c_stream]),
    "cartnet_equi_tp2_bwd": (C.c_int, [c_f32p, c_f32p, c_f32p, c_i32p, c_i32p, c_i32p, c_f32p, C.c_int32, c_f32p, c_f32p,
                                       c_stream]),
    "cartnet_colstats_nparts": (C.c_int, [C.c_int32]),
    "cartnet_colstats_partial": (C.c_int, [c_f32p, C.c_int32, C.c_int32, C.c_int32, c_f32p, c_f32p, c_stream]),
    "cartnet_splitk_reduce": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_int32,
                                        C.c_int32, C.c_int32, c_stream]),
    "cartnet_colsum_finalize": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_int32,
                                          c_stream]),
    "cartnet_colsum_finalize2": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int32,
                                           C.c_int32, C.c_int32, c_stream]),
    "cartnet_colsum_finalize_f32": (C.c_int, [c_f32p, C.c_int32, C.c_int32, c_f32p, c_stream]),
    "cartnet_csr_build": (C.c_int, [c_i64p, C.c_int64, C.c_int32, c_i64p, C.c_int32, c_i32p, c_i32p, c_i32p, c_i32p,
                                    c_i32p, c_i32p, c_stream]),
    "cartnet_csc_build": (C.c_int, [c_i32p, c_i32p, c_i64p, C.c_int32, C.c_int32, C.c_int64, c_i32p, c_i32p, c_i32p,
                                    c_stream]),
    "cartnet_edge_features": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, C.c_int64, C.c_int32, C.c_int32, C.c_float,
                                        C.c_float, c_f32p, C.c_int32, c_f32p, c_stream]),
    "cartnet_node_embed": (C.c_int, [c_i64p, c_i64p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int32, C.c_int32,
                                     C.c_int32, C.c_int32, c_i32p, c_f32p, c_stream]),
    "cartnet_node_nparts": (C.c_int, [C.c_int32]),
    "cartnet_node_embed_bwd": (C.c_int, [c_i64p, c_f32p, c_f32p, C.c_int32, C.c_int32, C.c_int32, c_f32p, c_f32p,
                                         c_stream]),
    "cartnet_sort_by_key": (C.c_int, [c_i64p, C.c_int32, C.c_int32, c_i32p, c_i32p, c_i32p, c_stream]),
    "cartnet_segment_sum_long": (C.c_int, [c_f32p, C.c_int32, c_i32p, c_i32p, C.c_int32, C.c_int32, C.c_int32, c_f32p,
                                           c_f32p, C.c_int32, c_stream]),
    "cartnet_segment_chunked_rows": (C.c_int32, [C.c_int32, C.c_int32]),
    "cartnet_segment_sum_chunked": (C.c_int, [c_f32p, C.c_int32, c_i32p, c_i32p, C.c_int32, C.c_int32, C.c_int32, c_f32p,
                                              c_f32p, C.c_int32, c_stream]),
    "cartnet_segment_sum_chunked_fold3": (C.c_int, [c_f32p, C.c_int32, c_i32p, C.c_int32, C.c_int32, C.c_int32, c_f32p,
                                                    c_f32p, C.c_int32, c_f32p, c_stream]),
    "cartnet_bn_finalize": (C.c_int, [c_f32p, c_f32p, C.c_int32, C.c_int64, C.c_int32, C.c_float, C.c_float,
                                      C.c_int32, c_f32p, c_f32p, c_i64p, c_f32p, c_groups, C.c_int32, C.c_int32,
                                      c_stream]),
    "cartnet_bn_sync_gather": (C.c_int, [c_f32p, c_f32p, C.c_int32, C.c_int32, C.c_int64, c_f32p, c_f32p, c_f32p, c_stream]),
    "cartnet_bn_finalize_row": (C.c_int, [c_f32p, C.c_int32, C.c_float, C.c_float, c_f32p, c_f32p, c_i64p, c_f32p, c_stream]),
    "cartnet_bn_sync_scale": (C.c_int, [c_f32p, C.c_int32, C.c_int64, c_f32p, c_stream]),
    "cartnet_group_ptrs": (C.c_int, [c_i64p, C.c_int32, C.c_int32, c_i32p, C.c_int32, c_i32p, c_i32p, c_stream]),
    "cartnet_colstats_grouped": (C.c_int, [c_f32p, C.c_int32, C.c_int32, c_groups, c_f32p, c_f32p, c_stream]),
    "cartnet_group_sums_finalize": (C.c_int, [c_f32p, c_f32p, C.c_int32, c_groups, C.c_int32, c_f32p, c_f32p, c_f32p,
                                              c_stream]),
    "cartnet_gate_scatter_nparts": (C.c_int, [C.c_int32]),
    "cartnet_gate_scatter_fwd": (C.c_int, [c_f32p, c_f32p, c_f32p, c_i32p, c_f32p, c_f32p, c_f32p, C.c_int32,
                                           C.c_int32, c_f32p, c_f32p, c_f32p, c_f32p, c_groups, c_stream]),
    "cartnet_gate_scatter_fwd_bc": (C.c_int, [c_f32p, c_f32p, c_f32p, c_i32p, c_f32p, c_f32p, c_f32p, C.c_int32,
                                              C.c_int32, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_stream]),
    "cartnet_node_update_bwd_apply_bc": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int32, C.c_int32,
                                                   C.c_int32, c_f32p, c_f32p, c_f32p, c_f32p, c_stream]),
    "cartnet_gemm_gate_stats_ok": (C.c_int, [C.POINTER(GemmArgs)]),
    "cartnet_gate_scatter_bwd_stats": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_i32p, c_f32p, c_f32p, c_f32p,
                                                 C.c_int32, C.c_int32, c_f32p, c_f32p, c_groups, c_stream]),
    "cartnet_gate_scatter_bwd_apply": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_i32p, c_f32p, c_f32p, c_f32p,
                                                 c_f32p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, c_f32p, c_f32p,
                                                 c_groups, c_stream]),
    "cartnet_segment_sum": (C.c_int, [c_f32p, C.c_int32, c_i32p, c_i32p, C.c_int32, C.c_int32, c_f32p, C.c_int32,
                                      c_stream]),
    "cartnet_segment_sum_h": (C.c_int, [c_f32p, C.c_int32, c_i32p, c_i32p, C.c_int32, C.c_int32, c_f32p, C.c_int32,
                                        c_stream]),
    "cartnet_segment_sum_pair": (C.c_int, [c_f32p, C.c_int32, c_i32p, c_i32p, c_i32p, C.c_int32, C.c_int32, c_f32p, c_f32p,
                                           C.c_int32, C.c_int32, c_stream]),
    "cartnet_segment_sum_pair_h": (C.c_int, [c_f32p, C.c_int32, c_i32p, c_i32p, c_i32p, C.c_int32, C.c_int32, c_f32p,
                                             c_f32p, C.c_int32, C.c_int32, c_stream]),
    "cartnet_gate_scatter_fwd_h": (C.c_int, [c_f32p, c_f32p, c_f32p, c_i32p, c_f32p, c_f32p, c_f32p, C.c_int32,
                                             C.c_int32, c_f32p, c_f32p, c_f32p, c_f32p, c_groups, c_stream]),
    "cartnet_gate_scatter_bwd_stats_h": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_i32p, c_f32p, c_f32p, c_f32p,
                                                   C.c_int32, C.c_int32, c_f32p, c_f32p, c_groups, c_stream]),
    "cartnet_gate_scatter_bwd_apply_h": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_i32p, c_f32p, c_f32p, c_f32p,
                                                   c_f32p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, c_f32p, c_f32p,
                                                   c_groups, c_stream]),
    "cartnet_node_update_fwd": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int32, C.c_int32, c_f32p,
                                          c_groups, c_stream]),
    # prototype (csrc/coop_layer.hip): one cooperative launch for one layer's forward at configs[2] sizes
    "cartnet_coop_layer_workspace_floats": (C.c_size_t, [C.c_int32, C.c_int32]),
    "cartnet_coop_layer_fwd": (C.c_int, [C.c_void_p] * 15 + [C.c_int32, C.c_int32, C.c_float] + [C.c_void_p] * 8 +
                               [C.c_uint32, C.c_void_p, c_stream]),
    "cartnet_node_update_bwd_stats": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int32, C.c_int32, c_f32p,
                                                c_f32p, c_groups, c_stream]),
    "cartnet_node_update_bwd_apply": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int32, C.c_int32,
                                                C.c_int32, c_f32p, c_groups, c_stream]),
    "cartnet_mask_index": (C.c_int, [c_u8p, C.c_int32, c_i32p, c_i32p, c_stream]),
    "cartnet_cholesky_head_fwd": (C.c_int, [c_f32p, c_i32p, c_f32p, c_f32p, C.c_int32, C.c_int32, c_f32p, c_f32p,
                                            c_stream]),
    "cartnet_cholesky_head_bwd": (C.c_int, [c_f32p, c_i32p, c_f32p, c_f32p, c_f32p, C.c_int32, C.c_int32, c_f32p,
                                            c_f32p, c_stream]),
    "cartnet_scalar_head_fwd": (C.c_int, [c_f32p, c_f32p, c_f32p, c_i64p, C.c_int32, C.c_int32, c_f32p, c_stream]),
    "cartnet_scalar_head_bwd": (C.c_int, [c_f32p, c_f32p, c_i64p, c_i64p, c_f32p, C.c_int32, C.c_int32, C.c_int32,
                                          c_f32p, c_f32p, c_stream]),
    "cartnet_transpose": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int32),
                                    C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32, c_stream]),
    "cartnet_rbf_expand": (C.c_int, [c_f32p, C.c_int64, c_f32p, C.c_int32, C.c_float, c_f32p, C.c_int32, c_stream]),
    "cartnet_lattice_features": (C.c_int, [c_f32p, c_i64p, c_i32p, c_f32p, c_f32p, C.c_int64, C.c_int32, c_f32p, c_f32p,
                                           c_f32p, c_stream]),
    "cartnet_eltwise": (C.c_int, [C.c_int32, c_f32p, c_f32p, c_f32p, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                  C.c_int32, C.c_float, c_stream]),
    "cartnet_segment_nparts": (C.c_int, [C.c_int32]),
    "cartnet_rowmul_fwd": (C.c_int, [c_f32p, C.c_int32, c_f32p, C.c_int32, c_i32p, C.c_int32, C.c_int32, C.c_float,
                                     c_f32p, C.c_int32, c_f32p, c_f32p, c_stream]),
    "cartnet_rowmul_bwd": (C.c_int, [c_f32p, C.c_int32, c_f32p, C.c_int32, c_f32p, C.c_int32, c_i32p, C.c_int32,
                                     C.c_int32, C.c_float, c_f32p, C.c_int32, c_stream]),
    "cartnet_rowmul_bwd_sums": (C.c_int, [c_f32p, C.c_int32, c_f32p, C.c_int32, c_f32p, C.c_int32, c_i32p, C.c_int32,
                                          C.c_int32, C.c_float, c_f32p, C.c_int32, c_f32p, c_f32p, c_stream]),
    "cartnet_att_gate_fwd": (C.c_int, [c_f32p, c_f32p, C.c_int32, c_i32p, c_f32p, c_f32p, c_f32p, C.c_float, C.c_int32,
                                       C.c_int32, c_f32p, c_f32p, c_stream]),
    "cartnet_att_gate_bwd_apply": (C.c_int, [c_f32p, c_f32p, C.c_int32, c_f32p, C.c_int32, c_f32p, c_i32p, c_f32p, c_f32p,
                                             c_f32p, c_f32p, C.c_int64, C.c_int32, C.c_float, C.c_int32, C.c_int32, c_f32p,
                                             C.c_int32, c_f32p, c_f32p, c_f32p, c_stream]),
    "cartnet_softplus_update_fwd": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int64, C.c_int32, c_f32p,
                                              c_stream]),
    "cartnet_softplus_update_bwd_stats": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int32,
                                                    C.c_int32, c_f32p, c_f32p, c_stream]),
    "cartnet_softplus_update_bwd_apply": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int32,
                                                    C.c_int32, C.c_int32, c_f32p, c_f32p, c_f32p, c_stream]),
    "cartnet_softplus_update_bwd_apply_sums": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p,
                                                         C.c_int32, C.c_int32, C.c_int32, c_f32p, c_f32p, c_f32p, c_f32p,
                                                         c_stream]),
    "cartnet_softplus_bwd_sums": (C.c_int, [c_f32p, C.c_int32, c_f32p, C.c_int32, c_f32p, C.c_int32, C.c_int32, C.c_int32,
                                            c_f32p, c_stream]),
    "cartnet_colsum_partial": (C.c_int, [c_f32p, C.c_int32, C.c_int32, C.c_int32, c_f32p, c_stream]),
    "cartnet_coldot_bc_partial": (C.c_int, [c_f32p, C.c_int32, c_f32p, C.c_int32, C.c_int32, c_f32p, c_f32p, c_stream]),
    "cartnet_radius_graph_count": (C.c_int, [c_f32p, c_f32p, c_i64p, c_i64p, C.c_int32, C.c_int32, C.c_float, c_i32p,
                                             c_i32p, c_stream]),
    "cartnet_radius_graph_fill": (C.c_int, [c_f32p, c_f32p, c_i64p, c_i64p, c_i32p, c_i64p, C.c_int32, C.c_int32,
                                            C.c_float, C.c_int64, c_i64p, c_f32p, c_f32p, c_f32p, c_stream]),
    "cartnet_neighbor_cap_count": (C.c_int, [c_i64p, c_f32p, C.c_int32, C.c_int32, C.c_float, c_f32p, c_i32p,
                                             c_stream]),
    "cartnet_neighbor_cap_fill": (C.c_int, [c_i64p, c_i64p, c_f32p, c_f32p, c_i64p, c_f32p, c_f32p, C.c_int32,
                                            C.c_int64, C.c_int64, c_i64p, c_f32p, c_f32p, c_stream]),
    "cartnet_loss_nparts": (C.c_int32, [C.c_int64]),
    "cartnet_loss_fwd": (C.c_int, [c_f32p, c_f32p, C.c_int64, C.c_void_p, c_f32p, c_stream]),
    "cartnet_loss_bwd": (C.c_int, [c_f32p, c_f32p, C.c_int64, c_f32p, c_f32p, c_f32p, c_stream]),
    "cartnet_loss_fwd_unit": (C.c_int, [c_f32p, c_f32p, C.c_int64, c_f32p, c_f32p, c_stream]),
    "cartnet_adp_metrics": (C.c_int, [c_f32p, c_f32p, C.c_int32, c_f32p, C.c_int32, c_f32p, c_f32p, c_f32p, c_stream]),
    "cartnet_collate": (C.c_int, [C.POINTER(Shard), c_i64p, c_i64p, c_i64p, c_i64p, C.c_int32, C.c_int64, C.c_int64,
                                  C.c_int64, c_f32p, C.c_float, C.c_float, C.POINTER(Collated), c_stream]),
    "cartnet_profile_gemm": (C.c_int, [C.c_int32]),
    "cartnet_profile_gemm_only": (C.c_int, [C.c_int32]),
    "cartnet_profile_gemm_every": (C.c_int, [C.c_int32]),
    "cartnet_profile_gemm_read": (C.c_int, [C.POINTER(GemmProfile), C.c_int32]),
    "cartnet_workspace_bytes": (C.c_size_t, [C.POINTER(Model), C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32]),
    "cartnet_model_forward": (C.c_int, [C.POINTER(Model), C.POINTER(BatchDesc), C.c_void_p, C.c_size_t, C.c_int32,
                                        C.c_int32, c_f32p, c_f32p, c_f32p, c_i32p, c_stream, c_stream]),
    "cartnet_model_backward": (C.c_int, [C.POINTER(Model), C.POINTER(BatchDesc), C.c_void_p, C.c_size_t, C.c_int32,
                                         c_f32p, c_f32p, C.POINTER(Params), c_stream, c_stream]),
    "cartnet_adam_step": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, C.c_int64, C.c_float, C.c_float, C.c_float,
                                    C.c_float, C.c_int32, C.c_float, c_stream]),
    "cartnet_icomformer_workspace_bytes": (C.c_size_t, [C.POINTER(IcfModel), C.c_int32, C.c_int64, C.c_int32, C.c_int32]),
    "cartnet_icomformer_forward": (C.c_int, [C.POINTER(IcfModel), C.POINTER(BatchDesc), c_f32p, C.c_void_p, C.c_size_t,
                                             C.c_int32, c_f32p, c_f32p, c_i32p, c_stream, c_stream]),
    "cartnet_icomformer_backward": (C.c_int, [C.POINTER(IcfModel), C.POINTER(BatchDesc), C.c_void_p, C.c_size_t, C.c_int32,
                                              c_f32p, c_f32p, C.POINTER(IcfParams), c_stream, c_stream]),
}


class CartnetHipError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load the shared library (once) and attach prototypes.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CartnetHipError(
            f"{LIB_PATH} not found: build it with `python -m cartnet_amd.build` (hipcc --offload-arch=gfx950). "
            "There is no CPU fallback for the CartNet hot path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)   # AttributeError if the symbol is missing -> loud
        fn.restype = res
        fn.argtypes = args
    # the ctypes mirrors above must have the C layouts: a mismatch would otherwise show up as a device fault
    if lib.cartnet_abi_version() != ABI_VERSION:
        raise CartnetHipError(f"{LIB_PATH}: ABI version {lib.cartnet_abi_version()}, this binding is written for "
                              f"{ABI_VERSION} -- rebuild the library (python -m cartnet_amd.build)")
    mirrors = [GemmArgs, Shard, Collated, GemmProfile, Groups, LayerParams, LayerBuffers, Params, Model, BatchDesc,
               GateGemmArgs, IcfConv, IcfParams, IcfModel]
    sizes = (C.c_size_t * 16)()
    n = lib.cartnet_abi_struct_sizes(sizes, 16)
    if n != len(mirrors):
        raise CartnetHipError(f"{LIB_PATH}: {n} ABI structs, this binding mirrors {len(mirrors)} -- rebuild the library")
    for cls, sz in zip(mirrors, sizes):
        if C.sizeof(cls) != sz:
            raise CartnetHipError(f"{LIB_PATH}: sizeof({cls.__name__}) is {sz} in the library, {C.sizeof(cls)} in "
                                  "cartnet_amd/lib.py -- rebuild the library (python -m cartnet_amd.build)")
    # experiment builds only (CARTNET_BUILD_EXPERIMENTAL=1, csrc/experimental/): CARTNET_Q selects the quad kernel
    if os.environ.get("CARTNET_Q") and hasattr(lib, "cartnet_gemm_experimental_q"):
        lib.cartnet_gemm_experimental_q(int(os.environ["CARTNET_Q"]))
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().cartnet_last_error()
        raise CartnetHipError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr() -> int:
    """hipStream_t of torch's current stream on the current device.  Through torch's raw accessor when it exists (0.3 us
    against ~9 us for `torch.cuda.current_stream().cuda_stream`, which builds a Stream object and resolves the device
    three times -- with several hundred wrapper calls per iComformer step that was 3 ms of host time per step)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    """Raw device pointer of a tensor (None stays NULL)."""
    if t is None:
        return None
    return t.data_ptr()
