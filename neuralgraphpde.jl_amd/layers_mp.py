"""Edge-function layers of the reference on the native message-passing primitives
(/root/reference/src/layers.jl): ExplicitEdgeConv :84-112, VMHConv :295-332, MPPDEConv :377-422,
GNOConv :485-547, SpectralConv :633-662, and a GAT-style layer on `softmax_edge_neighbors` semantics
[GraphNeuralNetworks.jl GATConv; the reference re-exports only the primitive, src/NeuralGraphPDE.jl:7].

Each edge-function layer is ONE library call forward and one in the pullback (include/ngpde.h: ngpde_edge_layer_*,
ngpde_gno_layer_*; functional.edge_layer / gno_layer): this module only names the blocks the reference vcats
(state, fixed node features, positions, edge features, theta) and the Dense stacks.  What the library does with them --
phi's first layer evaluated at NODE level, `z_e = P[t_e] + Q[s_e] + E_e`, the fused message launch or the primitives,
the node update as a chain -- is in csrc/api_layers.hip; tests/composed.py keeps the same layers composed from the
primitives' autograd nodes as the checker.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import _lib
from . import functional as F
from .graphs import EMPTYGRAPH, GNNGraph
from .layers import (AbstractGNNContainerLayer, AbstractGNNLayer, Chain, Dense, _act_code, glorot_uniform, rows_of,
                     zeros32)
from .utils import wrapgraph


def _dense_stack(layer, ps, what):
    """ϕ / ψ / γ as a list of (Dense, params): a Dense or a Chain of Dense (every use in the reference)."""
    if isinstance(layer, Dense):
        return [(layer, ps)]
    if isinstance(layer, Chain) and layer.chain and all(isinstance(l, Dense) for l in layer.chain):
        return [(l, ps[n]) for n, l in zip(layer.names(), layer.chain)]
    raise _lib.NgpdeError(_lib.ERR_UNSUPPORTED, f"{what} must be a Dense or a Chain of Dense layers, got {layer!r}")


def _wt_b(ps):
    return rows_of(ps["weight"]), (ps["bias"].reshape(-1) if "bias" in ps else None)


def _stack_spec(stack):
    """[(weight [in][out], bias or None, activation code)] of a Dense stack: what ngpde_edge_layer_* takes"""
    return [(*_wt_b(ps), layer.act) for layer, ps in stack]


def _node_data(g, device, exclude=()):
    """vcat(values(g.ndata)...) without the keys in `exclude`, as a cached [N][sum d] float32 tensor."""
    key = ("ndata-", tuple(exclude), str(device))
    p = g._packs.get(key)
    if p is None:
        from .graphs import _as_matrix_t
        cols = [_as_matrix_t(v, g.num_nodes).to(device) for k, v in g.ndata.items() if k not in exclude]
        p = torch.cat(cols, dim=1).contiguous() if cols else torch.zeros((g.num_nodes, 0), dtype=torch.float32, device=device)
        g._packs[key] = p
    return p


def _edge_data_p(g, handle, device):
    """vcat(values(g.edata)...) permuted once into p order (CSR by target)."""
    key = ("edata-p", str(device))
    p = g._packs.get(key)
    if p is None:
        e = g.packed("edata", device)
        p = F.edge_permute(e, handle) if e.shape[1] else e
        g._packs[key] = p
    return p


def _check_nodes(xr, g):
    if xr.shape[0] != g.num_nodes:
        raise _lib.DimensionMismatch(_lib.ERR_DIMENSION_MISMATCH,
                                     f"DimensionMismatch: input has {xr.shape[0]} columns, graph has {g.num_nodes} nodes")


def _as_named(x):
    return x if isinstance(x, dict) else {"preservedname": x}     # src/layers.jl:94-96, :308-310


# ---- ExplicitEdgeConv ----------------------------------------------------------------------------------


class ExplicitEdgeConv(AbstractGNNContainerLayer):
    """ExplicitEdgeConv(ϕ; initialgraph, aggr=mean):  h'_i = aggr_j ϕ([h_i; h_j; x_j - x_i])  (src/layers.jl:84-112)"""

    layers = ("ϕ",)

    def __init__(self, ϕ, *, initialgraph=None, aggr="mean"):
        self.ϕ, self.aggr = ϕ, aggr
        self.initialgraph = wrapgraph(initialgraph if initialgraph is not None else (lambda: EMPTYGRAPH))

    def __call__(self, x, ps, st):
        g = st["graph"]
        xn = {k: rows_of(v) for k, v in _as_named(x).items()}
        dev = next(iter(xn.values())).device
        for v in xn.values():
            _check_nodes(v, g)
        pos = _node_data(g, dev, exclude=[k for k in g.ndata if k != "x"])        # xi.x
        others = _node_data(g, dev, exclude=["x"])                                # drop(xi, :x) fixed part
        y = F.edge_layer(g.handle(), _lib.LAYER_EDGECONV, self.aggr, list(xn.values()), _stack_spec(_dense_stack(self.ϕ, ps, "ϕ")),
                         node_feat=others, pos=pos)                                # propagate(message, g, aggr)  (:111)
        return y.T, st


# ---- VMHConv ---------------------------------------------------------------------------------------------


class VMHConv(AbstractGNNContainerLayer):
    """VMHConv(ϕ, γ; initialgraph, aggr=mean):  m_i = aggr_j ϕ([h_i; h_j - h_i; x_j - x_i]), h' = γ([h_i; m_i])
    (src/layers.jl:295-332)"""

    layers = ("ϕ", "γ")

    def __init__(self, ϕ, γ, *, initialgraph=None, aggr="mean"):
        self.ϕ, self.γ, self.aggr = ϕ, γ, aggr
        self.initialgraph = wrapgraph(initialgraph if initialgraph is not None else (lambda: EMPTYGRAPH))

    def __call__(self, x, ps, st):
        g = st["graph"]
        xn = {k: rows_of(v) for k, v in _as_named(x).items()}
        dev = next(iter(xn.values())).device
        for v in xn.values():
            _check_nodes(v, g)
        pos = _node_data(g, dev, exclude=[k for k in g.ndata if k != "x"])
        others = _node_data(g, dev, exclude=["x"])
        y = F.edge_layer(g.handle(), _lib.LAYER_VMH, self.aggr, list(xn.values()), _stack_spec(_dense_stack(self.ϕ, ps["ϕ"], "ϕ")),
                         _stack_spec(_dense_stack(self.γ, ps["γ"], "γ")), node_feat=others, pos=pos)     # :316-328
        return y.T, st


# ---- MPPDEConv -------------------------------------------------------------------------------------------


class MPPDEConv(AbstractGNNContainerLayer):
    """MPPDEConv(ϕ, ψ; initialgraph, aggr=mean):  m_i = aggr_j ϕ([h_i; h_j; d_i - d_j; e_ij; θ]),
    h'_i = ψ([h_i; m_i; θ])   (src/layers.jl:377-422).  θ = vcat(g.gdata) per graph; batched graphs must
    share one structure (:359-361) and be stored contiguously (:410, :418)."""

    layers = ("ϕ", "ψ")

    def __init__(self, ϕ, ψ, *, aggr="mean", initialgraph=None):
        self.ϕ, self.ψ, self.aggr = ϕ, ψ, aggr
        self.initialgraph = wrapgraph(initialgraph if initialgraph is not None else (lambda: EMPTYGRAPH))

    def __call__(self, x, ps, st):
        g = st["graph"]
        h = rows_of(x)
        dev = h.device
        _check_nodes(h, g)
        handle = g.handle()
        y = F.edge_layer(handle, _lib.LAYER_MPPDE, self.aggr, [h], _stack_spec(_dense_stack(self.ϕ, ps["ϕ"], "ϕ")),
                         _stack_spec(_dense_stack(self.ψ, ps["ψ"], "ψ")), node_feat=g.packed("ndata", dev),     # :403-405
                         edge_feat=_edge_data_p(g, handle, dev), theta=g.packed("gdata", dev))                 # :407, :397
        return y.T, st


# ---- GNOConv ---------------------------------------------------------------------------------------------


class GNOConv(AbstractGNNContainerLayer):
    """GNOConv(in => out, ϕ, activation=identity; initialgraph, init_weight, init_bias, aggr=mean, bias=true)
    m_i = aggr_j reshape(ϕ([s_i; s_j; e_ij]), out, in) h_j;  h'_i = σ(W h_i + m_i + b)   (src/layers.jl:485-547)"""

    layers = ("linear", "ϕ")

    def __init__(self, *args, initialgraph=None, init_weight=glorot_uniform, init_bias=zeros32, aggr="mean", bias=True):
        if isinstance(args[0], (tuple, list)):             # GNOConv(in => out, ϕ[, activation])      (:501)
            ch, rest = args[0], args[1:]
        else:                                              # GNOConv(in, out, ϕ[, activation])        (:494)
            ch, rest = (args[0], args[1]), args[2:]
        ϕ = rest[0]
        activation = rest[1] if len(rest) > 1 else "identity"
        self.in_chs, self.out_chs = int(ch[0]), int(ch[1])
        self.ϕ, self.aggr, self.bias = ϕ, aggr, bool(bias)
        self.linear = Dense(self.in_chs, self.out_chs, activation, init_weight=init_weight, init_bias=init_bias, bias=bias)
        self.initialgraph = wrapgraph(initialgraph if initialgraph is not None else (lambda: EMPTYGRAPH))

    def __call__(self, x, ps, st):
        g = st["graph"]
        h = rows_of(x)
        dev = h.device
        _check_nodes(h, g)
        handle = g.handle()
        lwt, lb = _wt_b(ps["linear"])
        y = F.gno_layer(handle, self.in_chs, self.out_chs, self.aggr, self.linear.act, h, lwt, lb, _stack_spec(_dense_stack(self.ϕ, ps["ϕ"], "ϕ")),
                        node_feat=g.packed("ndata", dev), edge_feat=_edge_data_p(g, handle, dev))       # :517-547
        return y.T, st


# ---- SpectralConv ------------------------------------------------------------------------------------------


class SpectralConv(AbstractGNNLayer):
    """SpectralConv(n): Fourier differentiation on n periodic points (toy layer, src/layers.jl:633-662)."""

    def __init__(self, n):
        self.n = int(n)

    def __repr__(self):
        return f"SpectralConv({self.n})"

    def initialstates(self, rng):                              # :639-648
        n = self.n
        xs = np.linspace(0.0, 2.0 * np.pi, n + 1)[1:]
        s, t = np.nonzero(~np.eye(n, dtype=bool))               # complete digraph, lexicographic (src, dst)
        diff = (xs[t] - xs[s]).astype(np.float32)
        return {"graph": GNNGraph(s, t, num_nodes=n, index_base=0, edata=diff.reshape(1, -1))}

    def initialparameters(self, rng):                           # :650
        return {}

    def __call__(self, x, ps, st):                              # :652-662
        g = st["graph"]
        vec = (x.dim() == 1)
        xr = rows_of(x.reshape(1, -1) if vec else x)
        _check_nodes(xr, g)
        key = ("spectral-w", self.n, str(xr.device))
        w = g._packs.get(key)
        if w is None:
            w = F.spectral_weights(g.packed("edata", xr.device)[:, 0].contiguous(), self.n)
            g._packs[key] = w
        y = F.propagate_sum(xr, g.handle(), w)                  # propagate(message, g, +; xj = x, e = edata.e)
        return (y.reshape(-1) if vec else y.T), st


# ---- GAT-style layer ------------------------------------------------------------------------------------------


class GATConv(AbstractGNNLayer):
    """GATConv(in => out, σ=identity; heads=1, concat=true, negative_slope=0.2, add_self_loops=true, bias=true)
    in the explicit-parameter style of this package; semantics of GraphNeuralNetworks.jl's GATConv
    (logits leakyrelu(a . [Wx_i; Wx_j]), softmax over each node's incoming edges incl. a self loop)."""

    def __init__(self, ch, activation="identity", *, heads=1, concat=True, negative_slope=0.2, add_self_loops=True,
                 bias=True, init_weight=glorot_uniform, init_bias=zeros32, initialgraph=None):
        self.in_chs, self.out_chs = int(ch[0]), int(ch[1])
        self.heads, self.concat = int(heads), bool(concat)
        self.negative_slope, self.add_self_loops, self.bias = float(negative_slope), bool(add_self_loops), bool(bias)
        self._mix = {}
        self.activation, self.act = _act_code(activation)
        self.init_weight, self.init_bias = init_weight, init_bias
        self.initialgraph = wrapgraph(initialgraph if initialgraph is not None else (lambda: EMPTYGRAPH))

    def initialparameters(self, rng):
        c, h = self.out_chs, self.heads
        ps = {"weight": self.init_weight(rng, c * h, self.in_chs), "a": self.init_weight(rng, 2 * c, h)}
        if self.bias:
            ps["bias"] = self.init_bias(rng, c * h if self.concat else c, 1)
        return ps

    def _graph(self, g):
        if not self.add_self_loops:
            return g
        sl = getattr(g, "_with_self_loops", None)
        if sl is None:
            s, t = g.edge_index(0)
            loops = np.arange(g.num_nodes, dtype=np.int64)
            sl = GNNGraph(np.concatenate([s, loops]), np.concatenate([t, loops]), num_nodes=g.num_nodes, index_base=0)
            g._with_self_loops = sl
        return sl

    def __call__(self, x, ps, st):
        g = st["graph"]
        xr = rows_of(x)
        _check_nodes(xr, g)
        gs = self._graph(g)
        handle = gs.handle()
        c, h = self.out_chs, self.heads
        b = ps["bias"].reshape(-1) if "bias" in ps else None
        if self.concat and F.gat_layer_supported(handle, xr.shape[1], h, c):
            # 64 => heads x c = 64 on a graph whose tiles fit the LDS halo (BASELINE config 3): the whole layer is one launch
            y = F.gat_layer(xr, rows_of(ps["weight"]), rows_of(ps["a"]), b, handle, h, c, self.negative_slope, self.act,
                            gs.num_edges)
            return y.T, st
        wx = F.dense([xr], rows_of(ps["weight"]), None, 0)                          # Wx = reshape(W x, c, heads, N)
        out = F.gat_aggregate(wx, rows_of(ps["a"]), handle, h, c, self.negative_slope, gs.num_edges)
        if self.concat:
            y = F.bias_act(out, None, b, self.act)
        else:                                                                       # mean over heads: a genuine linear map
            dev = xr.device
            mix = self._mix.get(str(dev))                                           # built once per device
            if mix is None:
                mix = self._mix[str(dev)] = torch.eye(c, dtype=torch.float32, device=dev).repeat(h, 1) / h
            y = F.dense([out], mix, b, self.act)
        return y.T, st
