"""ctypes binding of libngpde_hip.so (the C ABI declared in include/ngpde.h).

The product path has NO fallback: if the shared library is missing or does not export a symbol the
import fails loudly.  Nothing here imports oracle/.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libngpde_hip.so")


class NgpdeError(RuntimeError):
    """A non-zero status from the C ABI (message from ngpde_last_error())."""

    def __init__(self, code, msg):
        super().__init__(f"[ngpde status {code}] {msg}")
        self.code = code
        self.msg = msg


class DimensionMismatch(NgpdeError, ValueError):
    """Mirrors Julia's DimensionMismatch (status NGPDE_ERR_DIMENSION_MISMATCH)."""


class ArgumentError(NgpdeError, ValueError):
    """Mirrors Julia's ArgumentError / AssertionError (status NGPDE_ERR_INVALID_ARGUMENT)."""


OK, ERR_INVALID_ARGUMENT, ERR_DIMENSION_MISMATCH, ERR_HIP, ERR_UNSUPPORTED, ERR_WORKSPACE, ERR_STATE = 0, -1, -2, -3, -4, -5, -6

ACT = {"identity": 0, "relu": 1, "tanh": 2, "sigmoid": 3, "swish": 4, "gelu": 5, "leakyrelu": 6,
       "elu": 7, "softplus": 8}
AGGR = {"+": 0, "sum": 0, "add": 0, "mean": 1, "max": 2, "min": 3, "*": 4, "mul": 4, "prod": 4}
TABLEAU = {"euler": 0, "tsit5": 1}

_vp, _i32, _i64, _sz, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_size_t, C.c_float

MLP_MAX_LAYERS = 8


class Mlp(C.Structure):
    """ngpde_mlp_t: a Dense stack (layer l: dims[l] => dims[l + 1])"""
    _fields_ = [("n_layers", _i32), ("dims", _i32 * (MLP_MAX_LAYERS + 1)), ("act", _i32 * MLP_MAX_LAYERS),
                ("weight", _vp * MLP_MAX_LAYERS), ("bias", _vp * MLP_MAX_LAYERS)]


class MlpGrad(C.Structure):
    """ngpde_mlp_grad_t"""
    _fields_ = [("dweight", _vp * MLP_MAX_LAYERS), ("dbias", _vp * MLP_MAX_LAYERS)]


class EdgeLayer(C.Structure):
    """ngpde_edge_layer_t: ExplicitEdgeConv / VMHConv / MPPDEConv as blocks + Dense stacks"""
    _fields_ = [("kind", _i32), ("aggr", _i32), ("n_state", _i32), ("state", _vp * 4), ("state_width", _i32 * 4),
                ("node_feat", _vp), ("node_feat_width", _i32), ("pos", _vp), ("pos_width", _i32),
                ("edge_feat", _vp), ("edge_feat_width", _i32), ("theta", _vp), ("theta_width", _i32),
                ("phi", Mlp), ("update", Mlp)]


class GnoLayer(C.Structure):
    """ngpde_gno_layer_t"""
    _fields_ = [("in_chs", _i32), ("out_chs", _i32), ("aggr", _i32), ("act", _i32), ("h", _vp), ("node_feat", _vp), ("node_feat_width", _i32),
                ("edge_feat", _vp), ("edge_feat_width", _i32), ("phi", Mlp), ("weight", _vp), ("bias", _vp)]


LAYER_EDGECONV, LAYER_VMH, LAYER_MPPDE = 0, 1, 2
RHS_GCN2, RHS_GAT, RHS_VMH = 1, 2, 3


class OdeDesc(C.Structure):
    """ngpde_ode_desc_t: the right-hand side of a fixed-step neural ODE, for ngpde_ode_create"""
    _fields_ = [("rhs", _i32), ("tableau", _i32), ("n_steps", _i32), ("with_backward", _i32), ("members", _i32), ("dt", C.c_double),
                ("width", _i32), ("act", _i32), ("heads", _i32), ("head_width", _i32), ("negative_slope", _f32),
                ("pos_width", _i32), ("aggr", _i32), ("pos", _vp),
                ("n_phi", _i32), ("phi_dims", _i32 * (MLP_MAX_LAYERS + 1)), ("phi_acts", _i32 * MLP_MAX_LAYERS),
                ("n_gamma", _i32), ("gamma_dims", _i32 * (MLP_MAX_LAYERS + 1)), ("gamma_acts", _i32 * MLP_MAX_LAYERS)]


class OdeWb(C.Structure):
    """ngpde_ode_wb_t"""
    _fields_ = [("weight", _vp * MLP_MAX_LAYERS), ("bias", _vp * MLP_MAX_LAYERS)]


class OdeParams(C.Structure):
    """ngpde_ode_params_t"""
    _fields_ = [("first", OdeWb), ("second", OdeWb), ("attention", _vp)]


class OdeGrads(C.Structure):
    """ngpde_ode_grads_t"""
    _fields_ = [("first", MlpGrad), ("second", MlpGrad), ("dattention", _vp)]

# name -> (restype, argtypes).  Every symbol include/ngpde.h declares must be listed here:
# tests/test_abi.py checks the header against this table and against the built library.
SIGNATURES = {
    "ngpde_version": (C.c_char_p, []),
    "ngpde_last_error": (C.c_char_p, []),
    "ngpde_graph_create": (_i32, [_i64, _i64, _vp, _vp, _i32, _i32, C.POINTER(_vp)]),
    "ngpde_graph_create_device": (_i32, [_i64, _i64, _vp, _vp, _i32, _i32, _i32, _vp, _vp, C.POINTER(_vp)]),
    "ngpde_graph_node_order": (_i32, [_vp, _vp]),
    "ngpde_radius_graph": (_i32, [_i64, _i32, _vp, _f32, _vp, _i32, _i32, _i32, _i32, _i32, _i64, _vp, _vp, C.POINTER(_i64), _vp]),
    "ngpde_knn_graph": (_i32, [_i64, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "ngpde_spatial_order": (_i32, [_i64, _i32, _vp, _vp, _i32, _i32, _vp, _vp]),
    "ngpde_graph_set_gcn_norm_device": (_i32, [_vp, _i32, _vp, _i32, _vp]),
    "ngpde_graph_array": (_i32, [_vp, _i32, _i32, _vp, _vp]),
    "ngpde_edge_layer_workspace_bytes": (_sz, [_vp, C.POINTER(EdgeLayer), _i32]),
    "ngpde_edge_layer_forward": (_i32, [_vp, C.POINTER(EdgeLayer), _i32, _vp, _vp, _sz, _vp]),
    "ngpde_edge_layer_backward": (_i32, [_vp, C.POINTER(EdgeLayer), _vp, _vp, C.POINTER(MlpGrad), C.POINTER(MlpGrad), _vp, _sz, _vp]),
    "ngpde_gno_layer_workspace_bytes": (_sz, [_vp, C.POINTER(GnoLayer), _i32]),
    "ngpde_gno_layer_forward": (_i32, [_vp, C.POINTER(GnoLayer), _i32, _vp, _vp, _sz, _vp]),
    "ngpde_gno_layer_backward": (_i32, [_vp, C.POINTER(GnoLayer), _vp, _vp, C.POINTER(MlpGrad), _vp, _vp, _vp, _sz, _vp]),
    "ngpde_rows_index": (_i32, [_i64, _i64, _i64, _i32, _vp, _vp, _vp, _i32, _vp]),
    "ngpde_comm_unique_id": (_i32, [_vp, _sz]),
    "ngpde_comm_create": (_i32, [_vp, _i32, _i32, C.POINTER(_vp)]),
    "ngpde_comm_destroy": (_i32, [_vp]),
    "ngpde_comm_info": (_i32, [_vp, C.POINTER(_i32), C.POINTER(_i32)]),
    "ngpde_comm_rccl_info": (_i32, [_vp, C.POINTER(_i32), C.POINTER(_i32)]),
    "ngpde_grad_allreduce": (_i32, [_vp, _vp, _i64, _vp]),
    "ngpde_grad_allreduce_adam": (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _f32, _f32, _f32, _f32, _i64, _vp]),
    "ngpde_node_vmh_supported": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _vp, _i32]),
    "ngpde_node_vmh_create": (_i32, [_vp, _i32, _i32, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _i32, _i32, _i32, C.c_double, _i32, C.POINTER(_vp)]),
    "ngpde_node_vmh_destroy": (_i32, [_vp]),
    "ngpde_node_vmh_tape_bytes": (_sz, [_vp]),
    "ngpde_release_cached_memory": (_sz, []),
    "ngpde_node_vmh_fault": (_i32, [_vp, _vp, C.POINTER(_i32)]),
    "ngpde_node_vmh_forward": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ngpde_node_vmh_backward": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ngpde_node_vmh_forward_saveat": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp]),
    "ngpde_node_vmh_backward_saveat": (_i32, [_vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ngpde_adam_step": (_i32, [_i64, _vp, _vp, _vp, _vp, _f32, _f32, _f32, _f32, _i64, _f32, _vp]),
    "ngpde_rprop_step": (_i32, [_i64, _vp, _vp, _vp, _vp, _f32, _f32, _f32, _f32, _f32, _vp]),
    "ngpde_graph_destroy": (_i32, [_vp]),
    "ngpde_graph_info": (_i32, [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i32)]),
    "ngpde_graph_csr_by_target": (_i32, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)]),
    "ngpde_graph_csr_by_source": (_i32, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)]),
    "ngpde_graph_set_gcn_norm": (_i32, [_vp, _i32, _vp, _i32]),
    "ngpde_gcn_workspace_bytes": (_sz, [_vp, _i32, _i32, _i32]),
    "ngpde_gcn_forward": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ngpde_gcn_backward": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ngpde_gcn_backward_ew_workspace_bytes": (_sz, [_vp, _i32, _i32]),
    "ngpde_gcn_backward_ew": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ngpde_propagate_copy_xj": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "ngpde_dense_forward": (_i32, [_i64, _i32, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "ngpde_dense_multi_forward": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ngpde_dense_pair_forward": (_i32, [_i64, _i32, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp,
                                        _i32, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "ngpde_dense_pair_backward_workspace_bytes": (_sz, [_i64, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _i32]),
    "ngpde_dense_pair_backward": (_i32, [_i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32,
                                         _vp, _vp, _vp, _sz, _vp]),
    "ngpde_dense_chain2_fused": (_i32, [_i64, _i32, _vp, _vp, _vp, _i32, _i32]),
    "ngpde_dense_chain2_forward": (_i32, [_i64, _i32, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "ngpde_dense_workspace_bytes": (_sz, [_i64, _i32, _i32]),
    "ngpde_dense_backward": (_i32, [_i64, _i32, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ngpde_edge_permute": (_i32, [_vp, _i32, _i32, _vp, _vp, _vp]),
    "ngpde_edge_combine_forward": (_i32, [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ngpde_edge_combine_backward": (_i32, [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ngpde_segment_reduce_forward": (_i32, [_vp, _i32, _i32, _vp, _vp, _vp]),
    "ngpde_segment_reduce_backward": (_i32, [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "ngpde_gno_contract_forward": (_i32, [_vp, _i32, _i32, _vp, _vp, _vp, _vp]),
    "ngpde_gno_contract_backward": (_i32, [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ngpde_gno_apply_supported": (_i32, [_i32, _i32]),
    "ngpde_gno_apply_forward": (_i32, [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "ngpde_gno_apply_backward": (_i32, [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ngpde_gat_forward": (_i32, [_vp, _i32, _i32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ngpde_gat_workspace_bytes": (_sz, [_vp, _i32]),
    "ngpde_gat_backward": (_i32, [_vp, _i32, _i32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ngpde_rk_stage_combine": (_i32, [_i64, _f32, _vp, _i32, _vp, _vp, _vp, _vp]),
    "ngpde_accumulate_many": (_i32, [_i32, _vp, _vp, _vp, _vp]),
    "ngpde_gno_message_supported": (_i32, [_i32, _i32]),
    "ngpde_gno_message_backward_from_nodes": (_i32, [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ngpde_gno_message_forward": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ngpde_ode_create": (_i32, [_vp, C.POINTER(OdeDesc), C.POINTER(_vp), C.POINTER(_i32)]),
    "ngpde_ode_destroy": (_i32, [_vp]),
    "ngpde_ode_tape_bytes": (_sz, [_vp]),
    "ngpde_ode_fault": (_i32, [_vp, _vp, C.POINTER(_i32)]),
    "ngpde_ode_forward": (_i32, [_vp, _vp, C.POINTER(OdeParams), _i32, _i32, _vp, _vp]),
    "ngpde_ode_backward": (_i32, [_vp, C.POINTER(OdeParams), _i32, _i32, _vp, _vp, C.POINTER(OdeGrads), _vp]),
    "ngpde_gno_gform_supported": (_i32, [_i32, _i32]),
    "ngpde_gno_gform_preferred": (_i32, [_i64, _i64, _i32, _i32, _i32, _i32]),
    "ngpde_gno_gform_splits": (_i32, [_i64, _i32, _i32, _i32]),
    "ngpde_gno_gform_aggregate": (_i32, [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ngpde_gno_gform_transform": (_i32, [_i64, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "ngpde_gat_layer_supported": (_i32, [_vp, _i32, _i32, _i32]),
    "ngpde_gat_layer_workspace_bytes": (_sz, [_vp, _i32, _i32]),
    "ngpde_gat_layer_forward": (_i32, [_vp, _i32, _i32, _i32, _f32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ngpde_gat_layer_backward": (_i32, [_vp, _i32, _i32, _i32, _f32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ngpde_bias_act_forward": (_i32, [_i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ngpde_bias_act_workspace_bytes": (_sz, [_i32]),
    "ngpde_bias_act_backward": (_i32, [_i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ngpde_edge_mlp_supported": (_i32, [_vp, _i32, _i32, _vp]),
    "ngpde_edge_mlp_forward": (_i32, [_vp, _i32, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp]),
    "ngpde_edge_mlp_backward_supported": (_i32, [_vp, _i32, _i32, _vp, _i32]),
    "ngpde_edge_mlp_backward_needs_edge_buffer": (_i32, [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _i32]),
    "ngpde_edge_mlp_backward_workspace_bytes": (_sz, [_vp, _i32, _i32, _vp]),
    "ngpde_edge_mlp_backward": (_i32, [_vp, _i32, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp,
                                        _vp, _sz, _vp]),
    "ngpde_activation_forward": (_i32, [_i64, _i32, _vp, _vp, _vp]),
    "ngpde_spectral_weights": (_i32, [_i64, _i32, _vp, _vp, _vp]),
    "ngpde_node_gcn2_create": (_i32, [_vp, _i32, _i32, _i32, _i32, _f32, _i32, C.POINTER(_vp)]),
    "ngpde_node_gcn2_create_batch": (_i32, [_vp, _i32, _i32, _i32, _i32, _i32, _f32, _i32, C.POINTER(_vp)]),
    "ngpde_node_destroy": (_i32, [_vp]),
    "ngpde_node_tape_bytes": (_sz, [_vp]),
    "ngpde_node_gcn2_forward": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ngpde_node_gcn2_backward": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ngpde_node_launch_count": (_i32, [_vp, C.POINTER(_i32), C.POINTER(_i32)]),
    "ngpde_node_flags": (_i32, [_vp, C.POINTER(_i32)]),
    "ngpde_hub_partition_host": (_i32, [_i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ngpde_batch_pad_host": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp]),
    "ngpde_node_fault": (_i32, [_vp, _vp, C.POINTER(_i32)]),
    "ngpde_row_blocks_gather": (_i32, [_i32, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp]),
    "ngpde_row_blocks_scatter": (_i32, [_i32, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp]),
    "ngpde_transpose": (_i32, [_i32, _i32, _vp, _vp, _vp]),
    "ngpde_rows_scale": (_i32, [_i64, _i32, _vp, _vp, _vp, _vp]),
    "ngpde_node_pipeline_stats": (_i32, [_vp, _vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    "ngpde_node_generation": (_i32, [_vp, C.POINTER(C.c_uint64), C.POINTER(_i32)]),
    "ngpde_node_expect_generation": (_i32, [_vp, C.c_uint64]),
    "ngpde_node_profile": (_i32, [_vp, _i32, C.POINTER(_f32), C.POINTER(_i32), _vp]),
    "ngpde_node_gat_supported": (_i32, [_vp, _i32, _i32, _i32]),
    "ngpde_node_gat_create": (_i32, [_vp, _i32, _i32, _f32, _i32, _i32, _i32, C.c_double, _i32, C.POINTER(_vp)]),
    "ngpde_node_gat_create_batch": (_i32, [_vp, _i32, _i32, _i32, _f32, _i32, _i32, _i32, C.c_double, _i32, C.POINTER(_vp)]),
    "ngpde_node_gat_destroy": (_i32, [_vp]),
    "ngpde_node_gat_tape_bytes": (_sz, [_vp]),
    "ngpde_node_gat_fault": (_i32, [_vp, _vp, C.POINTER(_i32)]),
    "ngpde_node_gat_forward": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ngpde_node_gat_backward": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
}

_lib = None


def load():
    """Load the shared library (once).  Raises ImportError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C neuralgraphpde.jl_amd/csrc`.  There is no CPU fallback for the hot path.")
    import torch  # noqa: F401  -- makes torch's libamdhip64.so the process's HIP runtime before ours resolves it
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise ImportError(f"{LIB_PATH} does not export {name}; rebuild the library") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status):
    if status == OK:
        return
    msg = load().ngpde_last_error().decode("utf-8", "replace")
    if status == ERR_DIMENSION_MISMATCH:
        raise DimensionMismatch(status, msg)
    if status == ERR_INVALID_ARGUMENT:
        raise ArgumentError(status, msg)
    raise NgpdeError(status, msg)


def ptr(t):
    """Device (or host) address of a torch tensor / numpy array, or None."""
    if t is None:
        return None
    if hasattr(t, "data_ptr"):
        return t.data_ptr()
    return t.ctypes.data


_raw_stream = None


def current_stream():
    """hipStream_t of torch's current stream on the current device.  Through torch's raw accessors where this build has them:
    torch.cuda.current_stream() builds a Stream object and resolves the device by name -- 25-30 us per call, more than the
    one-launch GAT layer takes on the device"""
    global _raw_stream
    import torch
    if _raw_stream is None:
        get, dev = getattr(torch._C, "_cuda_getCurrentRawStream", None), getattr(torch._C, "_cuda_getDevice", None)
        _raw_stream = (lambda: get(dev())) if (get and dev) else (lambda: torch.cuda.current_stream().cuda_stream)
    return _raw_stream()


# ---- destruction of library objects from Python finalisers -----------------------------------------------------------------
# A finaliser can run at ANY allocation -- also in the middle of a HIP-graph capture (NeuralODE(capture=True), bench.py's
# graph-replayed timings), where the hipFree inside ngpde_graph_destroy / ngpde_node_destroy is not a legal call: the capture is
# invalidated and torch aborts the process.  Finalisers therefore hand their pointer to destroy_later(), which frees at once
# when no capture is in progress on this thread and otherwise parks it until the next call that finds none.
_pending_destroy = []


def _capturing():
    try:
        import torch
        return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()
    except Exception:
        return False


def flush_destroy():
    if not _pending_destroy or _capturing():
        return
    lib = load()
    while _pending_destroy:
        fn, ptr = _pending_destroy.pop()
        try:
            getattr(lib, fn)(ptr)
        except Exception:
            pass


def destroy_later(fn, ptr):
    """fn: "ngpde_graph_destroy" | "ngpde_node_destroy" | "ngpde_ode_destroy"; ptr: the handle"""
    _pending_destroy.append((fn, ptr))
    flush_destroy()
