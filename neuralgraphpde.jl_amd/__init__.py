"""neuralgraphpde.jl_amd -- MI355X-native message-passing hot path of NeuralGraphPDE.jl behind the
reference's Lux explicit-layer API.  Import it as `ngpde_amd` (the directory name contains a dot,
so the repo-root shim ngpde_amd.py loads it under that name).

Everything that computes goes through libngpde_hip.so (include/ngpde.h); there is no CPU fallback.
"""
from . import _lib
from ._lib import ArgumentError, DimensionMismatch, NgpdeError
from .graphs import EMPTYGRAPH, GNNGraph, batch, knn_graph, radius_graph, rand_graph
from .utils import drop, updategraph, wrapgraph
from .layers import (AbstractExplicitLayer, AbstractGNNContainerLayer, AbstractGNNLayer, Chain, Dense,
                     GCNConv, apply, glorot_normal, glorot_uniform, setup, to_device, zeros32)
from .node import NeuralODE
from .layers_mp import ExplicitEdgeConv, GATConv, GNOConv, MPPDEConv, SpectralConv, VMHConv
from . import dist, optim, synth


def release_cached_memory():
    """device memory the library keeps parked for re-use (the tapes of destroyed NeuralODE(VMHConv) plans) back to the device; bytes released
    (the analogue of CUDA.reclaim() / torch.cuda.empty_cache() for the library's own allocations)"""
    _lib.flush_destroy()
    return int(_lib.load().ngpde_release_cached_memory())

__all__ = [
    "AbstractExplicitLayer", "AbstractGNNLayer", "AbstractGNNContainerLayer", "GCNConv", "Dense", "Chain", "NeuralODE",
    "ExplicitEdgeConv", "VMHConv", "MPPDEConv", "GNOConv", "SpectralConv", "GATConv",
    "setup", "apply", "to_device", "updategraph", "wrapgraph", "drop", "GNNGraph", "EMPTYGRAPH", "rand_graph",
    "batch", "radius_graph", "knn_graph", "glorot_uniform", "glorot_normal", "zeros32", "NgpdeError", "DimensionMismatch", "ArgumentError",
]
