"""Host-side mirror of the reference's Lux explicit-layer API for the hot path
(/root/reference/src/layers.jl): same layer names, constructor keywords, parameter / state
structure and error behaviour, with every forward evaluated by the HIP kernels behind the C ABI.

    ps, st = setup(rng, layer)          # Lux.setup
    y, st  = layer(x, ps, st)           # Lux.apply
    st     = updategraph(st, g)         # src/utils.jl:24-31

`ps` / `st` are insertion-ordered dicts standing in for Julia NamedTuples.  Arrays keep Julia's
shapes: features (D x N), weights (out x in), bias (out x 1); they are column-major views so the
kernels see [N][D] / [in][out] without copies.
"""
from __future__ import annotations

import math

import torch

from . import _lib
from . import functional as F
from .graphs import EMPTYGRAPH, GNNGraph
from .utils import wrapgraph

# ---- rng / initialisers (Lux.glorot_uniform, glorot_normal, zeros32) -------------------------------


def _gen(rng):
    if isinstance(rng, torch.Generator):
        return rng
    g = torch.Generator()
    g.manual_seed(0 if rng is None else int(rng))
    return g


def _colmajor(rows):
    """[in][out] contiguous -> (out x in) column-major view (Julia's memory order)."""
    return rows.T


def glorot_uniform(rng, out_dims, in_dims):
    lim = math.sqrt(6.0 / (in_dims + out_dims))
    return _colmajor((torch.rand(in_dims, out_dims, generator=_gen(rng)) * 2 - 1) * lim)


def glorot_normal(rng, out_dims, in_dims):
    std = math.sqrt(2.0 / (in_dims + out_dims))
    return _colmajor(torch.randn(in_dims, out_dims, generator=_gen(rng)) * std)


def zeros32(rng, *dims):
    return torch.zeros(*dims, dtype=torch.float32)


def _act_code(activation):
    name = activation if isinstance(activation, str) else getattr(activation, "__name__", str(activation))
    if name not in _lib.ACT:
        raise _lib.ArgumentError(_lib.ERR_INVALID_ARGUMENT, f"unsupported activation {activation!r}; one of {list(_lib.ACT)}")
    return name, _lib.ACT[name]


def rows_of(x):
    """(D x N) array -> float32 [N][D] contiguous tensor (no copy for column-major inputs)."""
    if not isinstance(x, torch.Tensor):
        x = torch.as_tensor(x)
    if x.dim() != 2:
        raise _lib.DimensionMismatch(_lib.ERR_DIMENSION_MISMATCH, f"DimensionMismatch: expected a (D x N) matrix, got shape {tuple(x.shape)}")
    xt = x.T
    if xt.dtype != torch.float32:
        xt = xt.to(torch.float32)
    return xt if xt.is_contiguous() else xt.contiguous()


# ---- Lux layer protocol ----------------------------------------------------------------------------


class AbstractExplicitLayer:
    def initialparameters(self, rng):
        return {}

    def initialstates(self, rng):
        return {}

    def parameterlength(self):
        return sum(int(v.numel()) for v in _leaves(self.initialparameters(0)))

    def statelength(self):
        return 0


def _leaves(d):
    for v in d.values():
        if isinstance(v, dict):
            yield from _leaves(v)
        else:
            yield v


def setup(rng, layer):
    """Lux.setup(rng, layer) -> (ps, st)"""
    rng = _gen(rng)
    return layer.initialparameters(rng), layer.initialstates(rng)


def apply(layer, x, ps, st):
    """Lux.apply(layer, x, ps, st) -> (y, st)"""
    return layer(x, ps, st)


def to_device(tree, device):
    """`ps |> device` for nested dicts of tensors (GNNGraph leaves are left alone: their features
    are packed onto the GPU lazily by the layers)."""
    if isinstance(tree, dict):
        return {k: to_device(v, device) for k, v in tree.items()}
    if isinstance(tree, torch.Tensor):
        return tree.to(device)
    return tree


class AbstractGNNLayer(AbstractExplicitLayer):
    """src/layers.jl:5 -- a layer whose graph lives in `st.graph`."""

    initialgraph = staticmethod(lambda: EMPTYGRAPH)

    def initialstates(self, rng):                      # src/layers.jl:23
        return {"graph": self.initialgraph()}

    def statelength(self):                             # src/layers.jl:24
        return 1


class AbstractGNNContainerLayer(AbstractGNNLayer):
    """src/layers.jl:12 -- a GNN layer holding sub-layers named by `layers`."""

    layers = ()

    def sublayer(self, name):
        # Python NFKC-normalises identifiers (the reference's field name U+03D5 becomes U+03C6 as an
        # attribute); parameter / state KEYS keep the reference's spelling.
        import unicodedata
        return getattr(self, unicodedata.normalize("NFKC", name))

    def initialstates(self, rng):                      # src/layers.jl:26-30: sub-layer states, then graph
        st = {name: self.sublayer(name).initialstates(rng) for name in self.layers}
        st["graph"] = self.initialgraph()
        return st

    def statelength(self):                             # src/layers.jl:32-34
        return sum(self.sublayer(name).statelength() for name in self.layers) + 1

    def initialparameters(self, rng):
        # Lux 0.4: a container with ONE sub-layer field gets that layer's parameters un-nested
        # (docs/src/devdoc.md:74-88); otherwise NamedTuple{layers}
        if len(self.layers) == 1:
            return self.sublayer(self.layers[0]).initialparameters(rng)
        return {name: self.sublayer(name).initialparameters(rng) for name in self.layers}


class Dense(AbstractExplicitLayer):
    """Lux.Dense(in => out, activation; bias=true): y = act.(W x .+ b)  [UPSTREAM Lux 0.4]."""

    def __init__(self, in_dims, out_dims=None, activation="identity", *, init_weight=glorot_uniform,
                 init_bias=zeros32, bias=True):
        if out_dims is None:
            in_dims, out_dims = in_dims
        self.in_dims, self.out_dims = int(in_dims), int(out_dims)
        self.activation, self.act = _act_code(activation)
        self.init_weight, self.init_bias, self.bias = init_weight, init_bias, bool(bias)

    def initialparameters(self, rng):
        ps = {"weight": self.init_weight(rng, self.out_dims, self.in_dims)}
        if self.bias:
            ps["bias"] = self.init_bias(rng, self.out_dims, 1)
        return ps

    def parameterlength(self):
        return self.out_dims * (self.in_dims + (1 if self.bias else 0))

    def __call__(self, x, ps, st):
        xr = rows_of(x)
        if xr.shape[1] != self.in_dims:
            raise _lib.DimensionMismatch(_lib.ERR_DIMENSION_MISMATCH,
                                         f"DimensionMismatch: Dense({self.in_dims} => {self.out_dims}) got {xr.shape[1]} rows")
        b = ps["bias"].reshape(-1) if "bias" in ps else None
        return F.dense([xr], rows_of(ps["weight"]), b, self.act).T, st

    def __repr__(self):
        return f"Dense({self.in_dims} => {self.out_dims}" + ("" if self.activation == "identity" else f", {self.activation}") + ")"


class Chain(AbstractExplicitLayer):
    """Lux.Chain: parameters / states named layer_1 ... layer_k."""

    def __init__(self, *layers):
        self.chain = list(layers)

    def names(self):
        return [f"layer_{i + 1}" for i in range(len(self.chain))]

    def initialparameters(self, rng):
        return {n: l.initialparameters(rng) for n, l in zip(self.names(), self.chain)}

    def initialstates(self, rng):
        return {n: l.initialstates(rng) for n, l in zip(self.names(), self.chain)}

    def statelength(self):
        return sum(l.statelength() for l in self.chain)

    def __call__(self, x, ps, st):
        new_st = {}
        for n, l in zip(self.names(), self.chain):
            x, new_st[n] = l(x, ps[n], st[n])
        return x, new_st


# ---- GCNConv  (src/layers.jl:147-239) ---------------------------------------------------------------


class GCNConv(AbstractGNNLayer):
    """GCNConv(in => out, activation=identity; initialgraph, init_weight, init_bias, bias=true,
    add_self_loops=true, use_edge_weight=false)

    The positional form GCNConv(in, out, ...) defaults to glorot_normal and the pair form
    GCNConv((in, out), ...) to glorot_uniform, as in the reference (src/layers.jl:178 vs :193).
    """

    def __init__(self, in_chs, out_chs=None, activation="identity", *, initialgraph=None, init_weight=None,
                 init_bias=zeros32, bias=True, add_self_loops=True, use_edge_weight=False):
        if isinstance(in_chs, (tuple, list)):           # `in => out`
            if out_chs is not None and activation == "identity":
                activation = out_chs
            in_chs, out_chs = in_chs
            default_init = glorot_uniform
        else:
            default_init = glorot_normal
        self.in_chs, self.out_chs = int(in_chs), int(out_chs)
        self.activation, self.act = _act_code(activation)
        self.init_weight = init_weight or default_init
        self.init_bias = init_bias
        self.bias = bool(bias)
        self.add_self_loops = bool(add_self_loops)
        self.use_edge_weight = bool(use_edge_weight)
        self.initialgraph = wrapgraph(initialgraph if initialgraph is not None else (lambda: EMPTYGRAPH))

    def __repr__(self):                                # src/layers.jl:158-162
        return f"GCNConv({self.in_chs} => {self.out_chs}" + ("" if self.activation == "identity" else f", {self.activation}") + ")"

    def initialparameters(self, rng):                  # src/layers.jl:164-171
        ps = {"weight": self.init_weight(rng, self.out_chs, self.in_chs)}
        if self.bias:
            ps["bias"] = self.init_bias(rng, self.out_chs, 1)
        return ps

    def parameterlength(self):                         # src/layers.jl:173-175
        return self.out_chs * (self.in_chs + 1) if self.bias else self.out_chs * self.in_chs

    def __call__(self, x, ps, st, edge_weight=None):   # src/layers.jl:200-239
        g = st["graph"]
        if edge_weight is not None:
            n_w = int(edge_weight.numel() if isinstance(edge_weight, torch.Tensor) else len(edge_weight))
            if n_w != g.num_edges:                     # :207
                raise _lib.ArgumentError(_lib.ERR_INVALID_ARGUMENT,
                                         f"Wrong number of edge weights (expected {g.num_edges} but given {n_w})")
            # The weights stay where they are (no download): the handle cache keys on their identity and version
            # (GNNGraph.handle).  A weight tensor that requires a gradient gets one: the reference differentiates through
            # e_mul_xj and the weighted degree (:224-231); ngpde_gcn_backward_ew.
            w = edge_weight.detach() if isinstance(edge_weight, torch.Tensor) else edge_weight
            norm = (self.add_self_loops, w, True)      # :224 degree uses the given weights
        elif self.use_edge_weight:
            if g.edge_weight is None:
                raise _lib.ArgumentError(_lib.ERR_INVALID_ARGUMENT, "use_edge_weight=true but the graph stores no edge weights")
            norm = (self.add_self_loops, g.edge_weight, False)   # reference quirk: unweighted degree (:224 vs :230)
        else:
            norm = (self.add_self_loops, None, False)
        xr = rows_of(x)
        if xr.shape[0] != g.num_nodes:
            raise _lib.DimensionMismatch(_lib.ERR_DIMENSION_MISMATCH,
                                         f"DimensionMismatch: x has {xr.shape[0]} columns, graph has {g.num_nodes} nodes")
        if xr.shape[1] != self.in_chs:
            raise _lib.DimensionMismatch(_lib.ERR_DIMENSION_MISMATCH,
                                         f"DimensionMismatch: x has {xr.shape[1]} rows, layer expects {self.in_chs}")
        wt = rows_of(ps["weight"])                     # (out x in) column-major -> [in][out]
        b = ps["bias"].reshape(-1) if "bias" in ps else None
        ew_leaf = edge_weight if (isinstance(edge_weight, torch.Tensor) and edge_weight.requires_grad and torch.is_grad_enabled()) else None
        if ew_leaf is not None and not ew_leaf.is_cuda:
            raise _lib.ArgumentError(_lib.ERR_INVALID_ARGUMENT, "GCNConv: an edge_weight that requires a gradient must live on the GPU")
        y = F.gcn_conv(xr, wt, b, g.handle(norm), self.act, ew_leaf)
        return y.T, st
