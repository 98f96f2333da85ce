"""GNNGraph -- host-side mirror of the parts of GraphNeuralNetworks.jl's GNNGraph that
NeuralGraphPDE.jl's layers touch (re-exported at /root/reference/src/NeuralGraphPDE.jl:4), plus
the cache of native derived-graph handles (CSR by target / by source) the HIP kernels use.

Indices follow the reference: `GNNGraph(s, t)` takes the 1-based COO vectors a Julia graph holds
(`index_base=0` for 0-based callers).  Features are (D x N) arrays, node n = column n.
"""
from __future__ import annotations

import ctypes as C
import os
from collections import OrderedDict

import numpy as np
import torch

from . import _lib


def _last_dim(a):
    return a.shape[-1] if hasattr(a, "shape") and len(a.shape) else 1


def _normalize(data, default, n, what):
    """GNN.jl normalize_graphdata: bare array -> {default: array}; last dim must equal n."""
    if data is None:
        return {}
    if not isinstance(data, dict):
        data = {default: data}
    out = {}
    for k, v in data.items():
        shape = tuple(v.shape)
        if len(shape) == 1 and default == "u" and n == 1:
            pass  # a vector is one graph's feature column
        elif len(shape) == 0 or shape[-1] != n:
            raise _lib.DimensionMismatch(
                _lib.ERR_DIMENSION_MISMATCH,
                f"DimensionMismatch: {what} feature '{k}' has size {shape}, last dimension must be {n}")
        out[k] = v
    return out


def _as_matrix_t(v, n, gdata=False):
    """user feature (D x n) [or vector] -> contiguous float32 torch tensor [n][D] on v's device."""
    t = v if isinstance(v, torch.Tensor) else torch.as_tensor(np.asarray(v))
    if t.dim() == 1:
        t = t.reshape(-1, 1) if (gdata and n == 1) else t.reshape(1, -1)
    return t.to(torch.float32).T.contiguous()


class _Handle:
    """Owner of one ngpde_graph_t (destroyed with the last GNNGraph copy that shares it).

    Built on the device (ngpde_graph_create_device: the COO list is uploaded once per structure as int32, the locality
    order is computed once per structure and cached -- batches reuse their members' orders); NGPDE_HOST_GRAPH_BUILD=1
    selects the host builder (ngpde_graph_create), which produces bit-identical arrays."""

    def __init__(self, g, norm):
        lib = _lib.load()
        _lib.flush_destroy()            # handles whose finaliser ran inside a HIP-graph capture
        out = C.c_void_p()
        self.lib = lib
        self.ptr = None
        self._targets, self._n_nodes, self._inv_deg = g._t0, g.num_nodes, {}
        if os.environ.get("NGPDE_HOST_GRAPH_BUILD") == "1":
            s = np.ascontiguousarray(g._s0, dtype=np.int64)
            t = np.ascontiguousarray(g._t0, dtype=np.int64)
            _lib.check(lib.ngpde_graph_create(g.num_nodes, g.num_edges, s.ctypes.data, t.ctypes.data, 0,
                                              g.num_graphs, C.byref(out)))
            self.ptr = out
            if norm is not None:
                add_self_loops, w, weighted = norm
                wp = None
                if w is not None:
                    w = np.ascontiguousarray(_to_numpy(w), dtype=np.float32)
                    wp = w.ctypes.data
                _lib.check(lib.ngpde_graph_set_gcn_norm(self.ptr, int(add_self_loops), wp, int(weighted)))
            if g._shared.get("order") is None:
                order = np.empty(g.num_nodes, dtype=np.int32)
                _lib.check(lib.ngpde_graph_node_order(self.ptr, order.ctypes.data))
                g._shared["order"] = order
            return
        if not torch.cuda.is_available():
            raise _lib.NgpdeError(_lib.ERR_HIP, "no HIP device: the derived-graph handle lives in HBM (there is no CPU fallback)")
        dev = torch.device("cuda", torch.cuda.current_device())
        coo = g._shared.get(("coo", str(dev)))
        if coo is None:
            coo = (torch.as_tensor(g._s0.astype(np.int32), device=dev), torch.as_tensor(g._t0.astype(np.int32), device=dev))
            g._shared[("coo", str(dev))] = coo
        order = g._shared.get("order")
        order_t = torch.as_tensor(order, device=dev) if order is not None else None
        _lib.check(lib.ngpde_graph_create_device(g.num_nodes, g.num_edges, _lib.ptr(coo[0]), _lib.ptr(coo[1]), 32, 0,
                                                 g.num_graphs, _lib.ptr(order_t), _lib.current_stream(), C.byref(out)))
        self.ptr = out
        if order is None:
            order = np.empty(g.num_nodes, dtype=np.int32)
            _lib.check(lib.ngpde_graph_node_order(self.ptr, order.ctypes.data))
            g._shared["order"] = order
        if norm is not None:
            add_self_loops, w, weighted = norm
            wt = None
            if w is not None:
                wt = (w.detach() if isinstance(w, torch.Tensor) else torch.as_tensor(np.asarray(w))).to(dev, torch.float32).contiguous()
            _lib.check(lib.ngpde_graph_set_gcn_norm_device(self.ptr, int(add_self_loops), _lib.ptr(wt), int(weighted),
                                                           _lib.current_stream()))

    def inv_in_degree(self, device):
        """[N][1] float32 on `device`: 1 / max(in-degree, 1) -- what a mean aggregation's pullback multiplies a node's gradient by."""
        key = str(device)
        v = self._inv_deg.get(key)
        if v is None:
            deg = np.bincount(np.asarray(self._targets, dtype=np.int64), minlength=self._n_nodes).astype(np.float32)
            v = torch.as_tensor((np.float32(1.0) / np.maximum(deg, np.float32(1.0))).reshape(-1, 1), device=device)
            self._inv_deg[key] = v
        return v

    def __del__(self):
        try:
            if self.ptr:
                _lib.destroy_later("ngpde_graph_destroy", self.ptr)     # (not inside a HIP-graph capture: see _lib.destroy_later)
                self.ptr = None
        except Exception:
            pass


class GNNGraph:
    """COO graph with node / edge / graph features.

    GNNGraph(s, t; num_nodes, ndata, edata, gdata)        -- as test/runtests.jl:11-13
    GNNGraph(g; ndata=..., edata=..., gdata=...)          -- copy with data replaced (:28, :58, :145)
    """

    def __init__(self, s=None, t=None, *, num_nodes=None, ndata=None, edata=None, gdata=None,
                 num_graphs=None, edge_weight=None, index_base=1):
        if isinstance(s, GNNGraph):
            g = s
            self._s0, self._t0 = g._s0, g._t0
            self.num_nodes, self.num_edges = g.num_nodes, g.num_edges
            self.num_graphs = g.num_graphs if num_graphs is None else num_graphs
            self.edge_weight = g.edge_weight if edge_weight is None else edge_weight
            self._handles = g._handles          # same structure -> share the native handles
            self._shared = g._shared            # ... the device copy of the COO list and the locality order
            self.ndata = g.ndata if ndata is None else _normalize(ndata, "x", self.num_nodes, "node")
            self.edata = g.edata if edata is None else _normalize(edata, "e", self.num_edges, "edge")
            self.gdata = g.gdata if gdata is None else _normalize(gdata, "u", self.num_graphs, "graph")
            self._packs = {}
            self._members = getattr(g, "_members", None)
            return
        self._members = None
        s0 = np.asarray(s, dtype=np.int64).reshape(-1) - index_base
        t0 = np.asarray(t, dtype=np.int64).reshape(-1) - index_base
        if s0.shape != t0.shape:
            raise _lib.DimensionMismatch(_lib.ERR_DIMENSION_MISMATCH, "DimensionMismatch: s and t differ in length")
        if num_nodes is None:
            num_nodes = int(max(s0.max(initial=-1), t0.max(initial=-1)) + 1)
        if s0.size and (min(s0.min(), t0.min()) < 0 or max(s0.max(), t0.max()) >= num_nodes):
            raise _lib.DimensionMismatch(_lib.ERR_DIMENSION_MISMATCH,
                                         f"DimensionMismatch: edge index outside 1:{num_nodes}")
        self._s0, self._t0 = s0, t0
        self.num_nodes, self.num_edges = int(num_nodes), int(s0.size)
        self.num_graphs = 1 if num_graphs is None else int(num_graphs)
        self.edge_weight = edge_weight
        self.ndata = _normalize(ndata, "x", self.num_nodes, "node")
        self.edata = _normalize(edata, "e", self.num_edges, "edge")
        self.gdata = _normalize(gdata, "u", self.num_graphs, "graph")
        self._handles = {}
        self._shared = {}
        self._packs = {}

    # ---- reference-visible accessors ---------------------------------------------------------
    def edge_index(self, index_base=1):
        return self._s0 + index_base, self._t0 + index_base

    def copy(self, **kw):  # src/utils.jl:8  Base.copy(g; kwargs...) = GNNGraph(g; kwargs...)
        return GNNGraph(self, **kw)

    def __repr__(self):
        return f"GNNGraph({self.num_nodes}, {self.num_edges})"

    def __eq__(self, other):
        if not isinstance(other, GNNGraph):
            return NotImplemented
        if self is other:
            return True
        if (self.num_nodes, self.num_edges, self.num_graphs) != (other.num_nodes, other.num_edges, other.num_graphs):
            return False
        if not (np.array_equal(self._s0, other._s0) and np.array_equal(self._t0, other._t0)):
            return False
        for a, b in ((self.ndata, other.ndata), (self.edata, other.edata), (self.gdata, other.gdata)):
            if list(a) != list(b):
                return False
            for k in a:
                if a[k] is b[k]:
                    continue
                if not np.array_equal(_to_numpy(a[k]), _to_numpy(b[k])):
                    return False
        return True

    __hash__ = object.__hash__

    # ---- native side -------------------------------------------------------------------------
    def handle(self, norm=None):
        """ngpde_graph_t for this structure; `norm` = (add_self_loops, edge_weight or None,
        weighted_degree) selects a handle carrying that GCN normalisation."""
        if norm is None:
            norm = (False, None, False)   # the plain handle carries the tile schedule too (the fused edge kernels walk it)
        w = norm[1]
        if w is None:
            key = (bool(norm[0]), None, bool(norm[2]))
            h = self._handles.get(key)
            if h is None:
                h = self._handles[key] = (_Handle(self, norm), norm)
            return h[0]
        # Weighted normalisations: a small LRU keyed on the weights' identity AND content version, so that a caller passing
        # `edge_weight` on every call (an ODE right-hand side: ~300 calls per solve) re-uses one handle while the weights
        # are unchanged and never accumulates handles (each is a full CSR / halo build in HBM).  The entry keeps `w` alive,
        # so data_ptr / id stay unique while it is cached.
        if isinstance(w, torch.Tensor):
            key = (bool(norm[0]), ("tensor", w.data_ptr(), w._version, tuple(w.shape), str(w.device)), bool(norm[2]))
        else:
            key = (bool(norm[0]), ("object", id(w)), bool(norm[2]))
        lru = self._shared.setdefault("weighted_handles", OrderedDict())
        h = lru.get(key)
        if h is None:
            h = lru[key] = (_Handle(self, norm), norm)
            while len(lru) > self.max_weighted_handles:
                lru.popitem(last=False)         # the evicted _Handle frees its ngpde_graph_t when its last user lets go
        else:
            lru.move_to_end(key)
        return h[0]

    max_weighted_handles = 4

    def node_order(self):
        """the locality order of this structure (int32 permutation), computed with the first handle and cached"""
        if self._shared.get("order") is None:
            self.handle()
        return self._shared["order"]

    def packed(self, which, device):
        """Concatenation of all features of one kind in NamedTuple order as a contiguous float32
        [n][sum D] tensor on `device` (vcat(values(g.ndata)...), src/layers.jl:403-407, :517-521).
        Float64 graph features (test/runtests.jl:58) are converted: the HIP path is fp32."""
        key = (which, str(device))
        p = self._packs.get(key)
        if p is None:
            data = {"ndata": self.ndata, "edata": self.edata, "gdata": self.gdata}[which]
            n = {"ndata": self.num_nodes, "edata": self.num_edges, "gdata": self.num_graphs}[which]
            cols = [_as_matrix_t(v, n, gdata=(which == "gdata")).to(device) for v in data.values()]
            p = torch.cat(cols, dim=1).contiguous() if cols else torch.zeros((n, 0), dtype=torch.float32, device=device)
            self._packs[key] = p
        return p

    def feature_dims(self, which):
        data = {"ndata": self.ndata, "edata": self.edata, "gdata": self.gdata}[which]
        n = {"ndata": self.num_nodes, "edata": self.num_edges, "gdata": self.num_graphs}[which]
        out = {}
        for k, v in data.items():
            shape = tuple(v.shape)
            out[k] = 1 if len(shape) == 1 and not (which == "gdata" and n == 1) else int(shape[0])
            if len(shape) == 1 and which == "gdata" and n == 1:
                out[k] = int(shape[0])
        return out


def _to_numpy(v):
    if isinstance(v, torch.Tensor):
        return v.detach().cpu().numpy()
    return np.asarray(v)


EMPTYGRAPH = GNNGraph([], [], num_nodes=0)   # src/layers.jl:14  rand_graph(0, 0)


def rand_graph(n, m, *, bidirected=True, seed=None):
    """[UPSTREAM GNNGraphs.rand_graph] random graph with n nodes and m directed edges (m/2 symmetric
    pairs when bidirected).  No self loops, no multi-edges."""
    rng = np.random.default_rng(seed)
    if bidirected:
        assert m % 2 == 0, "bidirected rand_graph needs an even number of edges"
    npairs = m // 2 if bidirected else m
    max_pairs = n * (n - 1) // 2 if bidirected else n * (n - 1)
    assert npairs <= max_pairs, "too many edges requested"
    chosen = set()
    while len(chosen) < npairs:
        i, j = int(rng.integers(0, n)), int(rng.integers(0, n))
        if i == j:
            continue
        if bidirected and i > j:
            i, j = j, i
        chosen.add((i, j))
    pairs = np.array(sorted(chosen), dtype=np.int64).reshape(-1, 2)
    if bidirected:
        s = np.concatenate([pairs[:, 0], pairs[:, 1]])
        t = np.concatenate([pairs[:, 1], pairs[:, 0]])
    else:
        s, t = pairs[:, 0], pairs[:, 1]
    return GNNGraph(s, t, num_nodes=n, index_base=0)


def _device_points(points):
    """(dim x n) array / tensor -> contiguous float32 [n][dim] tensor on the current HIP device"""
    if not torch.cuda.is_available():
        raise _lib.NgpdeError(_lib.ERR_HIP, "no HIP device: the neighbour search runs on the GPU (there is no CPU fallback)")
    dev = torch.device("cuda", torch.cuda.current_device())
    t = points if isinstance(points, torch.Tensor) else torch.as_tensor(np.asarray(points))
    if t.dim() == 1:
        t = t.reshape(1, -1)
    if t.dim() != 2:
        raise _lib.DimensionMismatch(_lib.ERR_DIMENSION_MISMATCH, "DimensionMismatch: points must be a (dim x n) matrix")
    return t.to(dev, torch.float32).T.contiguous(), dev


def _device_indicator(graph_indicator, n, dev):
    if graph_indicator is None:
        return None, 1
    gi = graph_indicator if isinstance(graph_indicator, torch.Tensor) else torch.as_tensor(np.asarray(graph_indicator))
    gi = gi.reshape(-1).to(dev, torch.int32).contiguous()
    if gi.numel() != n:
        raise _lib.DimensionMismatch(_lib.ERR_DIMENSION_MISMATCH,
                                     f"DimensionMismatch: graph_indicator has {gi.numel()} entries for {n} points")
    return gi, (int(gi.max().item()) if n else 1)      # 1-based ids as in GNNGraphs: the number of graphs is the largest id


def _graph_from_device_coo(s, t, n, num_graphs, dev, pts, gi, locality):
    g = GNNGraph(s.cpu().numpy(), t.cpu().numpy(), num_nodes=n, index_base=0, num_graphs=num_graphs)
    g._shared[("coo", str(dev))] = (s, t)            # the handle builder takes the device lists as they are
    if locality == "spatial" and n:
        order = torch.empty(n, dtype=torch.int32, device=dev)
        _lib.check(_lib.load().ngpde_spatial_order(n, pts.shape[1], _lib.ptr(pts), _lib.ptr(gi), num_graphs, 1, _lib.ptr(order),
                                                   _lib.current_stream()))
        g._shared["order"] = order.cpu().numpy()
    elif locality != "bfs":
        raise ValueError("locality must be 'bfs' or 'spatial'")
    return g


def radius_graph(points, r, *, graph_indicator=None, self_loops=False, dir="in", locality="bfs"):
    """[UPSTREAM GNNGraphs.radius_graph(points, r; graph_indicator, self_loops, dir), re-exported at
    /root/reference/src/NeuralGraphPDE.jl:4] every point is linked to the points within distance r (<=) of its own graph.
    points: (dim x n), dim <= 3; graph_indicator: 1-based graph id per point.  Searched on the device
    (ngpde_radius_graph); edges come ordered by point, neighbours ascending.  locality="spatial" takes the tile
    schedule from a space-filling curve through the points instead of a host traversal of the graph."""
    pts, dev = _device_points(points)
    n, dim = pts.shape
    gi, num_graphs = _device_indicator(graph_indicator, n, dev)
    lib = _lib.load()
    ne = C.c_int64(0)
    if dir not in ("in", "out"):
        raise ValueError("dir must be 'in' or 'out'")
    args = (n, dim, _lib.ptr(pts), float(r), _lib.ptr(gi), num_graphs, 1, int(bool(self_loops)), int(dir == "out"), 0)
    _lib.check(lib.ngpde_radius_graph(*args, 0, None, None, C.byref(ne), _lib.current_stream()))
    m = int(ne.value)
    s = torch.empty(m, dtype=torch.int32, device=dev)
    t = torch.empty(m, dtype=torch.int32, device=dev)
    if m:
        _lib.check(lib.ngpde_radius_graph(*args, m, _lib.ptr(s), _lib.ptr(t), C.byref(ne), _lib.current_stream()))
    return _graph_from_device_coo(s, t, n, num_graphs, dev, pts, gi, locality)


def knn_graph(points, k, *, graph_indicator=None, self_loops=False, dir="in", locality="bfs"):
    """[UPSTREAM GNNGraphs.knn_graph(points, k; graph_indicator, self_loops, dir)] every point is linked to its k nearest
    points of its own graph (itself included only with self_loops=true).  Searched on the device (ngpde_knn_graph); the
    neighbours of a point come nearest first, coincident distances by index."""
    pts, dev = _device_points(points)
    n, dim = pts.shape
    gi, num_graphs = _device_indicator(graph_indicator, n, dev)
    if dir not in ("in", "out"):
        raise ValueError("dir must be 'in' or 'out'")
    s = torch.empty(n * int(k), dtype=torch.int32, device=dev)
    t = torch.empty(n * int(k), dtype=torch.int32, device=dev)
    _lib.check(_lib.load().ngpde_knn_graph(n, dim, _lib.ptr(pts), int(k), _lib.ptr(gi), num_graphs, 1, int(bool(self_loops)),
                                           int(dir == "out"), 0, _lib.ptr(s), _lib.ptr(t), _lib.current_stream()))
    return _graph_from_device_coo(s, t, n, num_graphs, dev, pts, gi, locality)


def batch(graphs):
    """MLUtils.batch(::Vector{GNNGraph}) [UPSTREAM]: block-diagonal union; features concatenated
    along the last dimension (test/runtests.jl:92)."""
    graphs = list(graphs)
    off, ss, tt = 0, [], []
    for g in graphs:
        ss.append(g._s0 + off)
        tt.append(g._t0 + off)
        off += g.num_nodes

    def cat(dicts, n_of, gdata=False):
        out = {}
        for k in dicts[0]:
            parts = []
            for d, g in zip(dicts, graphs):
                v = d[k]
                v = v if isinstance(v, torch.Tensor) else torch.as_tensor(np.asarray(v))
                if v.dim() == 1:
                    v = v.reshape(-1, 1) if (gdata and n_of(g) == 1) else v.reshape(1, -1)
                parts.append(v)
            out[k] = torch.cat(parts, dim=-1)
        return out

    out = GNNGraph(np.concatenate(ss) if ss else [], np.concatenate(tt) if tt else [], num_nodes=off,
                   index_base=0, num_graphs=sum(g.num_graphs for g in graphs))
    out.ndata = cat([g.ndata for g in graphs], lambda g: g.num_nodes)
    out.edata = cat([g.edata for g in graphs], lambda g: g.num_edges)
    out.gdata = cat([g.gdata for g in graphs], lambda g: g.num_graphs, gdata=True)
    out._members = graphs if all(g.num_graphs == 1 for g in graphs) else None   # (solver plans look for identical members)
    if graphs and graphs[0].edge_weight is not None:
        out.edge_weight = np.concatenate([np.asarray(g.edge_weight) for g in graphs])
    # the batch's locality order = its members' cached orders, offset: no graph traversal per minibatch
    if graphs and off > 0 and torch.cuda.is_available() and os.environ.get("NGPDE_HOST_GRAPH_BUILD") != "1":
        parts, o = [], 0
        for g in graphs:
            if g.num_nodes:
                parts.append(g.node_order() + np.int32(o))
            o += g.num_nodes
        out._shared["order"] = np.concatenate(parts).astype(np.int32)
    return out
