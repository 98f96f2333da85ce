"""Edge-function layers of the reference on the native message-passing primitives
(/root/reference/src/layers.jl): ExplicitEdgeConv :84-112, VMHConv :295-332, MPPDEConv :377-422,
GNOConv :485-547, SpectralConv :633-662, and a GAT-style layer on `softmax_edge_neighbors` semantics
[GraphNeuralNetworks.jl GATConv; the reference re-exports only the primitive, src/NeuralGraphPDE.jl:7].

The reference's message closures gather `xi`/`xj`, vcat the blocks and call the message MLP `ϕ` on
the whole edge set.  The first Dense layer of `ϕ` is linear in the concatenated blocks, so it is
evaluated at NODE level (two Dense calls over N columns instead of one over E columns) and the
gathers collapse into `z_e = P[t_e] + Q[s_e] + E_e` (ngpde_edge_combine_forward); the remaining layers
of `ϕ` run on the [E][h] activations, aggregation is an atomic-free segmented reduction.  Weight-block
slicing (tiny (out x in) matrices) is done with torch views so autograd reassembles the gradients.
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


def _tail(stack, a):
    """layers 2..k of a Dense stack on row-major activations"""
    for layer, ps in stack[1:]:
        wt, b = _wt_b(ps)
        a = F.dense([a], wt, b, layer.act)
    return a


def _node_update(stack, blocks, row_divs, n):
    """a Dense stack (psi / gamma) on a virtual vcat of node-level blocks: the first two layers as one chained call"""
    l1, p1 = stack[0]
    wt1, b1 = _wt_b(p1)
    if len(stack) >= 2:
        l2, p2 = stack[1]
        wt2, b2 = _wt_b(p2)
        y = F.dense_chain2(blocks, wt1, b1, l1.act, wt2, b2, l2.act, row_divs=row_divs, n=n)
        return _tail(stack[1:], y)
    return F.dense(blocks, wt1, b1, l1.act, row_divs=row_divs, n=n)


def _message_path(g, P, Q, Et, stack, aggr):
    """aggr_e phi(...) given the node-level first-layer terms: ONE fused launch when the message MLP fits the fused
    kernel (widths <= 64, multiples of 4, <= 3 further Dense layers, tiles fit the LDS halo; max/min only without
    gradients), the primitives otherwise."""
    l1 = stack[0][0]
    ref = next(t for t in (P, Q, Et) if t is not None)
    tail = [(_wt_b(ps)[0], _wt_b(ps)[1], layer.act) for layer, ps in stack[1:]]
    needs_grad = torch.is_grad_enabled() and any(
        t is not None and t.requires_grad for t in [P, Q, Et] + [w for w, _, _ in tail] + [b for _, b, _ in tail])
    aggr_code = _lib.AGGR[aggr]
    if os.environ.get("NGPDE_NO_FUSED_EDGE") != "1" and g.num_edges > 0 and (aggr_code in (0, 1) or (aggr_code in (2, 3, 4) and not needs_grad)):
        fh = g.handle((False, None, False))          # the handle that carries the tile schedule / halo lists
        if F.edge_mlp_supported(fh, ref.shape[1], [w.shape[1] for w, _, _ in tail]):
            return F.edge_mlp_fused(P, Q, Et, fh, l1.act, aggr, g.num_nodes, g.num_edges, tail)
    handle = g.handle()
    a = F.edge_combine(P, Q, Et, handle, l1.act, g.num_edges)
    return F.segment_reduce(_tail(stack, a), handle, aggr, g.num_nodes)


def _row_offsets(wt, sizes):
    """first rows of the consecutive row blocks of the first-layer weight (the order of the message's vcat)"""
    offs, o = [], 0
    for n in sizes:
        offs.append(o)
        o += n
    if o != wt.shape[0]:
        raise _lib.DimensionMismatch(_lib.ERR_DIMENSION_MISMATCH,
                                     f"DimensionMismatch: first layer expects {wt.shape[0]} input features, the message has {o}")
    return offs


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
        if os.environ.get("NGPDE_LAYERS_COMPOSED") == "1":
            return self._composed(xn, ps, st)
        pos = _node_data(g, dev, exclude=[k for k in g.ndata if k != "x"])        # xi.x
        others = _node_data(g, dev, exclude=["x"])                                # drop(xi, :x) fixed part
        y = F.edge_layer(g.handle(), _lib.LAYER_EDGECONV, self.aggr, list(xn.values()), _stack_spec(_dense_stack(self.ϕ, ps, "ϕ")),
                         node_feat=others, pos=pos)                                # propagate(message, g, aggr)  (:111)
        return y.T, st

    def _composed(self, xn, ps, st):
        """the same layer composed from the primitives' autograd nodes (rounds 1 - 4's host path; NGPDE_LAYERS_COMPOSED=1: the
        test of ngpde_edge_layer_* against it)"""
        g = st["graph"]
        dev = next(iter(xn.values())).device
        pos = _node_data(g, dev, exclude=[k for k in g.ndata if k != "x"])        # xi.x
        others = _node_data(g, dev, exclude=["x"])                                # drop(xi, :x) fixed part
        hblocks = list(xn.values()) + ([others] if others.shape[1] else [])
        stack = _dense_stack(self.ϕ, ps, "ϕ")                                     # reference passes the whole ps (:106)
        l1, p1 = stack[0]
        wt, b = _wt_b(p1)
        dh = sum(hb.shape[1] for hb in hblocks)
        dp = pos.shape[1]
        oa, ob, oc = _row_offsets(wt, [dh, dh, dp])                               # [hi...; hj...; xj - xi]
        wA, wB = F.row_blocks(wt, [[(dh, [(oa, 1)]), (dp, [(oc, -1)])], [(dh, [(ob, 1)]), (dp, [(oc, 1)])]])   # [wa; -wc], [wb; wc]
        fan = [F.fanout(hb, 2) for hb in hblocks]         # target side / source side: their cotangents are summed in one launch
        P, Q = F.dense_pair([f[0] for f in fan] + [pos], wA, b, 0, [f[1] for f in fan] + [pos], wB, None, 0)
        y = _message_path(g, P, Q, None, stack, self.aggr)                          # propagate(message, g, aggr)  (:111)
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
        if os.environ.get("NGPDE_LAYERS_COMPOSED") == "1":
            return self._composed(xn, ps, st)
        pos = _node_data(g, dev, exclude=[k for k in g.ndata if k != "x"])
        others = _node_data(g, dev, exclude=["x"])
        y = F.edge_layer(g.handle(), _lib.LAYER_VMH, self.aggr, list(xn.values()), _stack_spec(_dense_stack(self.ϕ, ps["ϕ"], "ϕ")),
                         _stack_spec(_dense_stack(self.γ, ps["γ"], "γ")), node_feat=others, pos=pos)     # :316-328
        return y.T, st

    def _composed(self, xn, ps, st):
        """composed from the primitives' autograd nodes (NGPDE_LAYERS_COMPOSED=1)"""
        g = st["graph"]
        dev = next(iter(xn.values())).device
        pos = _node_data(g, dev, exclude=[k for k in g.ndata if k != "x"])
        others = _node_data(g, dev, exclude=["x"])
        hblocks = list(xn.values()) + ([others] if others.shape[1] else [])
        stack = _dense_stack(self.ϕ, ps["ϕ"], "ϕ")
        l1, p1 = stack[0]
        wt, b = _wt_b(p1)
        dh = sum(hb.shape[1] for hb in hblocks)
        dp = pos.shape[1]
        oa, ob, oc = _row_offsets(wt, [dh, dh, dp])                               # [hi...; (hj - hi)...; xj - xi]  (:316)
        wA, wB = F.row_blocks(wt, [[(dh, [(oa, 1), (ob, -1)]), (dp, [(oc, -1)])],   # [wa - wb; -wc]
                                   [(dh, [(ob, 1)]), (dp, [(oc, 1)])]])             # [wb; wc]
        nx = len(xn)
        fan = [F.fanout(hb, 3 if k < nx else 2) for k, hb in enumerate(hblocks)]   # target side, source side, (features:) γ
        P, Q = F.dense_pair([f[0] for f in fan] + [pos], wA, b, 0, [f[1] for f in fan] + [pos], wB, None, 0)
        m = _message_path(g, P, Q, None, stack, self.aggr)                          # :326
        gstack = _dense_stack(self.γ, ps["γ"], "γ")
        blocks = [f[2] for f in fan[:nx]] + [m]
        y = _node_update(gstack, blocks, [1] * len(blocks), m.shape[0])           # γ(vcat(values(x)..., m))  (:328)
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
        if os.environ.get("NGPDE_LAYERS_COMPOSED") == "1":
            return self._composed(h, ps, st)
        handle = g.handle()
        y = F.edge_layer(handle, _lib.LAYER_MPPDE, self.aggr, [h], _stack_spec(_dense_stack(self.ϕ, ps["ϕ"], "ϕ")),
                         _stack_spec(_dense_stack(self.ψ, ps["ψ"], "ψ")), node_feat=g.packed("ndata", dev),     # :403-405
                         edge_feat=_edge_data_p(g, handle, dev), theta=g.packed("gdata", dev))                 # :407, :397
        return y.T, st

    def _composed(self, h, ps, st):
        """composed from the primitives' autograd nodes (NGPDE_LAYERS_COMPOSED=1)"""
        g = st["graph"]
        dev = h.device
        handle = g.handle()
        N, E, G = g.num_nodes, g.num_edges, max(g.num_graphs, 1)
        d = g.packed("ndata", dev)                                                 # :403-405
        theta = g.packed("gdata", dev)                                             # :397  [G][dθ]
        e_p = _edge_data_p(g, handle, dev)                                         # :407
        dh, dd, de, dth = h.shape[1], d.shape[1], e_p.shape[1], theta.shape[1]
        if dth and (N % G or E % G):
            raise _lib.DimensionMismatch(_lib.ERR_DIMENSION_MISMATCH,
                                         "DimensionMismatch: batched graphs must have the same structure (src/layers.jl:359-361)")
        stack = _dense_stack(self.ϕ, ps["ϕ"], "ϕ")
        l1, p1 = stack[0]
        wt, b = _wt_b(p1)
        oa, ob, oc, od, oe = _row_offsets(wt, [dh, dh, dd, de, dth])               # [hi; hj; di - dj; e; θ]  (:409-410)
        tb, trd = [h], [1]
        if dd:
            tb.append(d); trd.append(1)
        if dth:
            tb.append(theta); trd.append(N // G)                                   # θ of the target's graph = the edge's graph
        # the three recombined weights in one launch: target side [wa; wc; we], source side [wb; -wc], edge features wd
        mats = F.row_blocks(wt, [[(dh, [(oa, 1)]), (dd, [(oc, 1)]), (dth, [(oe, 1)])], [(dh, [(ob, 1)]), (dd, [(oc, -1)])]] +
                            ([[(de, [(od, 1)])]] if de else []))
        # one pass over h when the shapes allow; h comes back routed through the pair so that psi's gradient w.r.t. h is added
        # inside the pair's pullback launch
        P, Q, h = F.dense_pair(tb, mats[0], b, 0, [h] + ([d] if dd else []), mats[1], None, 0, row_divs_a=trd, n=N, passthrough=True)
        Et = F.dense([e_p], mats[2], None, 0) if de else None
        m = _message_path(g, P, Q, Et, stack, self.aggr)                            # :416
        pstack = _dense_stack(self.ψ, ps["ψ"], "ψ")
        blocks, rd = [h, m], [1, 1]
        if dth:
            blocks.append(theta); rd.append(N // G)
        y = _node_update(pstack, blocks, rd, N)                                    # ψ(vcat(x, m, repeat(θ)))  (:418)
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
        if os.environ.get("NGPDE_LAYERS_COMPOSED") == "1":
            return self._composed(h, ps, st)
        handle = g.handle()
        lwt, lb = _wt_b(ps["linear"])
        y = F.gno_layer(handle, self.in_chs, self.out_chs, self.aggr, self.linear.act, h, lwt, lb, _stack_spec(_dense_stack(self.ϕ, ps["ϕ"], "ϕ")),
                        node_feat=g.packed("ndata", dev), edge_feat=_edge_data_p(g, handle, dev))       # :517-547
        return y.T, st

    def _composed(self, h, ps, st):
        """composed from the primitives' autograd nodes (NGPDE_LAYERS_COMPOSED=1)"""
        g = st["graph"]
        dev = h.device
        handle = g.handle()
        N, E = g.num_nodes, g.num_edges
        s = g.packed("ndata", dev)                                                 # :517-519
        e_p = _edge_data_p(g, handle, dev)                                         # :521
        ds, de = s.shape[1], e_p.shape[1]
        stack = _dense_stack(self.ϕ, ps["ϕ"], "ϕ")
        l1, p1 = stack[0]
        wt, b = _wt_b(p1)
        oa, ob, od = _row_offsets(wt, [ds, ds, de])                                # [si; sj; e]  (:523)
        mats = F.row_blocks(wt, ([[(ds, [(oa, 1)])], [(ds, [(ob, 1)])]] if ds else []) + ([[(de, [(od, 1)])]] if de else []))
        wa, wb = (mats[0], mats[1]) if ds else (None, None)
        wd = mats[-1] if de else None
        Et = F.dense([e_p], wd, b if not ds else None, 0) if de else None
        P = Q = None                                                               # (node-level terms: below, in one launch)
        kout = _wt_b(stack[-1][1])[0].shape[1]
        if kout != self.in_chs * self.out_chs:
            raise _lib.DimensionMismatch(_lib.ERR_DIMENSION_MISMATCH,
                                         f"DimensionMismatch: ϕ must output in_chs*out_chs = {self.in_chs * self.out_chs} rows, got {kout}")
        last, plast = stack[-1]
        kdim = _wt_b(plast)[0].shape[0]
        reassoc = (len(stack) >= 2 and last.act == 0 and os.environ.get("NGPDE_GNO_MATERIALIZE") != "1"
                   and F.gno_apply_supported(self.out_chs, kdim))
        if reassoc:
            # reassociated: K_e h_j = T_j z_e + B2 h_j with T_j = W2 (x) h_j at node level; K is never formed
            w2, b2 = _wt_b(plast)                                                  # [k][in*out], [in*out]; row r = o + out*i
            wr = F.transpose(w2).view(self.in_chs, self.out_chs * kdim)            # [in][out][k]: the transpose of [k][in * out]
        lwt, lb = _wt_b(ps["linear"])
        # h has up to three consumers (T, B2 h, W h): one fan-out node sums their cotangents in one launch.  The small node-level
        # Dense layers -- P, Q on the node coordinates, B2 h, W h -- are latency-bound launches of a few dozen workgroups each:
        # ONE launch for all of them (ngpde_dense_multi_forward)
        hT, hS = F.fanout(h, 2) if reassoc else (None, h)
        small = ([(s, wa, b, 0), (s, wb, None, 0)] if ds else []) + ([(hS, b2.view(self.in_chs, self.out_chs), None, 0)] if (reassoc and b2 is not None) else []) + [(hS, lwt, None, 0)]
        outs = F.dense_multi(small)
        if ds:
            P, Q = outs[0], outs[1]
        Wh = outs[-1]
        Bh = None
        if reassoc:
            T = F.dense([hT], wr, None, 0)
            Bh = outs[-2] if b2 is not None else None
        if (reassoc and len(stack) == 2 and E > 0 and os.environ.get("NGPDE_NO_GNO_MFMA") != "1"
                and F.gno_message_supported(self.out_chs, kdim, l1.act)):
            # two-layer phi: the per-edge input act1(P[t] + Q[s] + E) is formed inside the message launch
            agg = F.gno_message_aggregate(P, Q, Et, T, Bh, handle, l1.act, self.out_chs, kdim, E, self.aggr, N)   # :527-534
            m = None
        elif reassoc:
            z = _tail(stack[:-1], F.edge_combine(P, Q, Et, handle, l1.act, E))
            m = F.gno_apply(T, Bh, z, handle, self.out_chs, kdim)
        else:
            K = _tail(stack, F.edge_combine(P, Q, Et, handle, l1.act, E))
            m = F.gno_contract(K, h, handle, self.in_chs, self.out_chs)            # :527-530
        if m is not None:
            agg = F.segment_reduce(m, handle, self.aggr, N)                        # :534
        y = F.bias_act(agg, Wh, lb, self.linear.act)                               # σ(W x + m + b)  (:536-547)
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
