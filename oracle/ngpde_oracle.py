"""
oracle/ngpde_oracle.py -- CPU restatement (numpy, float64 by default) of the message-passing
hot path of NeuralGraphPDE.jl.

THIS FILE IS TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and the `cpu_baseline`
leg of bench.py may import it, and only as the checker.  The product (neuralgraphpde.jl_amd/)
never imports anything from oracle/.

PARITY PINNING STATUS
  * SpectralConv is pinned by the reference's own known-answer test
    (/root/reference/test/runtests.jl:153-162: d/dx sin = cos, d/dx cos = -sin on 100 periodic
    points, sum(abs2, err) < 1f-3) and docstring residuals (src/layers.jl:590-630).  That test
    also pins the direction convention of `propagate` (xj gathered at the SOURCE, aggregated at
    the TARGET): flipping it fails the test.
  * The fixed-step Tsit5 tableau is pinned by its order conditions (row sums == c, 5th-order
    local error on u' = u).
  * Output VALUES of GCNConv / MPPDEConv / GNOConv / VMHConv / ExplicitEdgeConv / the GAT-style
    aggregation and ALL gradients are **parity unpinned**: the reference's tests only assert
    shapes and state structure (test/runtests.jl:16-151), the arithmetic lives in un-vendored
    Julia packages (GraphNeuralNetworks.jl 0.4-0.6 `propagate`, NNlib 0.8 `gather`/`scatter`/
    `batched_mul`, Lux 0.4 `Dense`), and no Julia toolchain exists in this image.  For those the
    oracle restates the published algorithm at the reference's call sites, and is cross-checked by
    (i) finite differences in float64, (ii) an independent C restatement (ngpde_oracle.c) and (iii) an independent
    torch float64 transcription of the same call sites whose gradients come from torch.autograd
    (tests/test_golden.py, on the committed vectors tests/golden/*.npz, generator tests/golden/make_golden.py).

Conventions (same as the reference): features are (D x N) arrays, node n = column n; edges are a
COO list (s_e -> t_e); `xi` = features gathered at the target t, `xj` = gathered at the source s,
messages are reduced at the target.  Indices are 0-based inside this file; `Graph(...,
index_base=1)` accepts the 1-based vectors a Julia GNNGraph holds.
"""
from __future__ import annotations

import numpy as np

# --------------------------------------------------------------------------------------------
# activations (NNlib names).  Each entry: f(z), f'(z)
# --------------------------------------------------------------------------------------------


def _sigmoid(z):
    return 1.0 / (1.0 + np.exp(-z))


_GELU_C = np.sqrt(2.0 / np.pi)


def _gelu(z):  # NNlib 0.8 gelu = tanh approximation
    return 0.5 * z * (1.0 + np.tanh(_GELU_C * (z + 0.044715 * z**3)))


def _dgelu(z):
    u = _GELU_C * (z + 0.044715 * z**3)
    th = np.tanh(u)
    return 0.5 * (1.0 + th) + 0.5 * z * (1.0 - th * th) * _GELU_C * (1.0 + 3 * 0.044715 * z * z)


ACTIVATIONS = {
    "identity": (lambda z: z, lambda z: np.ones_like(z)),
    "relu": (lambda z: np.maximum(z, 0.0), lambda z: (z > 0).astype(z.dtype)),
    "tanh": (np.tanh, lambda z: 1.0 - np.tanh(z) ** 2),
    "sigmoid": (_sigmoid, lambda z: _sigmoid(z) * (1.0 - _sigmoid(z))),
    "swish": (lambda z: z * _sigmoid(z),
              lambda z: _sigmoid(z) * (1.0 + z * (1.0 - _sigmoid(z)))),
    "gelu": (_gelu, _dgelu),
    "leakyrelu": (lambda z: np.where(z > 0, z, 0.01 * z),
                  lambda z: np.where(z > 0, 1.0, 0.01).astype(z.dtype)),
    "elu": (lambda z: np.where(z > 0, z, np.expm1(np.minimum(z, 0.0))),
            lambda z: np.where(z > 0, 1.0, np.exp(np.minimum(z, 0.0))).astype(z.dtype)),
    "softplus": (lambda z: np.logaddexp(0.0, z), _sigmoid),
}

# integer codes shared with include/ngpde.h (ngpde_act_t)
ACT_CODE = {"identity": 0, "relu": 1, "tanh": 2, "sigmoid": 3, "swish": 4, "gelu": 5,
            "leakyrelu": 6, "elu": 7, "softplus": 8}


def act(name, z):
    return ACTIVATIONS[name][0](z)


def dact(name, z):
    return ACTIVATIONS[name][1](z)


# --------------------------------------------------------------------------------------------
# graph container -- restates the parts of GNNGraph [UPSTREAM GraphNeuralNetworks.jl] that the
# reference touches: COO (s, t), num_nodes/num_edges/num_graphs, ndata/edata/gdata NamedTuples
# (here: insertion-ordered dicts), default names :x / :e / :u.
# --------------------------------------------------------------------------------------------


def _norm_data(d, default, n):
    """normalize_graphdata: bare array -> {default: array}; vectors become (1 x n) rows for
    ndata/edata and (len x 1) columns for gdata of a single graph; last dim must be n."""
    if d is None:
        return {}
    if not isinstance(d, dict):
        d = {default: d}
    out = {}
    for k, v in d.items():
        v = np.asarray(v)
        if v.ndim == 1:
            if default == "u":
                v = v.reshape(-1, 1) if n == 1 else v.reshape(1, -1)
            else:
                v = v.reshape(1, -1)
        if v.shape[-1] != n:
            raise ValueError(f"DimensionMismatch: feature '{k}' has last dim {v.shape[-1]}, expected {n}")
        out[k] = v
    return out


class Graph:
    def __init__(self, s, t, num_nodes=None, ndata=None, edata=None, gdata=None, num_graphs=1,
                 edge_weight=None, index_base=1):
        s = np.asarray(s, dtype=np.int64).reshape(-1) - index_base
        t = np.asarray(t, dtype=np.int64).reshape(-1) - index_base
        if s.shape != t.shape:
            raise ValueError("DimensionMismatch: s and t must have the same length")
        if num_nodes is None:
            num_nodes = int(max(s.max(initial=-1), t.max(initial=-1)) + 1)
        if s.size and (min(s.min(), t.min()) < 0 or max(s.max(), t.max()) >= num_nodes):
            raise ValueError("edge index out of range")
        self.s, self.t = s, t
        self.num_nodes, self.num_edges, self.num_graphs = int(num_nodes), int(s.size), int(num_graphs)
        self.ndata = _norm_data(ndata, "x", self.num_nodes)
        self.edata = _norm_data(edata, "e", self.num_edges)
        self.gdata = _norm_data(gdata, "u", self.num_graphs)
        self.edge_weight = None if edge_weight is None else np.asarray(edge_weight).reshape(-1)

    def copy(self, **kw):  # src/utils.jl:8  Base.copy(g; kwargs...) = GNNGraph(g; kwargs...)
        g = Graph.__new__(Graph)
        g.__dict__.update(self.__dict__)
        if "ndata" in kw:
            g.ndata = _norm_data(kw["ndata"], "x", g.num_nodes)
        if "edata" in kw:
            g.edata = _norm_data(kw["edata"], "e", g.num_edges)
        if "gdata" in kw:
            g.gdata = _norm_data(kw["gdata"], "u", g.num_graphs)
        return g


def batch(graphs):
    """MLUtils.batch of GNNGraphs [UPSTREAM]: block-diagonal union (test/runtests.jl:89-102)."""
    off, ss, tt = 0, [], []
    for g in graphs:
        ss.append(g.s + off)
        tt.append(g.t + off)
        off += g.num_nodes
    cat = lambda ds: {k: np.concatenate([d[k] for d in ds], axis=-1) for k in ds[0]}
    out = Graph(np.concatenate(ss), np.concatenate(tt), num_nodes=off, index_base=0,
                num_graphs=sum(g.num_graphs for g in graphs))
    out.ndata = cat([g.ndata for g in graphs])
    out.edata = cat([g.edata for g in graphs])
    out.gdata = cat([g.gdata for g in graphs])
    if graphs[0].edge_weight is not None:
        out.edge_weight = np.concatenate([g.edge_weight for g in graphs])
    return out


def add_self_loops(g):
    """[UPSTREAM] s <- [s; 1:N], t <- [t; 1:N]; existing self loops kept; weights padded with 1."""
    n = np.arange(g.num_nodes, dtype=np.int64)
    out = Graph(np.concatenate([g.s, n]), np.concatenate([g.t, n]), num_nodes=g.num_nodes,
                index_base=0, num_graphs=g.num_graphs)
    if g.edge_weight is not None:
        out.edge_weight = np.concatenate([g.edge_weight, np.ones(g.num_nodes, g.edge_weight.dtype)])
    return out


def degree_in(g, dtype, edge_weight=None):
    """[UPSTREAM] degree(g, T; dir=:in, edge_weight): count (or weight sum) of incoming edges."""
    w = np.ones(g.num_edges, dtype) if edge_weight is None else np.asarray(edge_weight, dtype)
    d = np.zeros(g.num_nodes, dtype)
    np.add.at(d, g.t, w)
    return d


# --------------------------------------------------------------------------------------------
# primitives [UPSTREAM NNlib 0.8]: gather / scatter and their pullbacks
# --------------------------------------------------------------------------------------------


def gather(X, idx):
    return X[..., idx]


def scatter(op, M, idx, n):
    """scatter(op, M, idx; dstsize=(..., n)).  `mean` of an empty neighbourhood is 0."""
    out_shape = M.shape[:-1] + (n,)
    if op in ("+", "add", "sum"):
        out = np.zeros(out_shape, M.dtype)
        np.add.at(np.moveaxis(out, -1, 0), idx, np.moveaxis(M, -1, 0))
        return out
    if op == "mean":
        out = scatter("+", M, idx, n)
        cnt = np.bincount(idx, minlength=n).astype(M.dtype)
        return out / np.where(cnt > 0, cnt, 1.0)
    if op in ("max", "min"):
        init = -np.inf if op == "max" else np.inf
        out = np.full(out_shape, init, M.dtype)
        f = np.maximum if op == "max" else np.minimum
        f.at(np.moveaxis(out, -1, 0), idx, np.moveaxis(M, -1, 0))
        return out
    if op in ("*", "mul"):
        out = np.ones(out_shape, M.dtype)
        np.multiply.at(np.moveaxis(out, -1, 0), idx, np.moveaxis(M, -1, 0))
        return out
    raise ValueError(f"unknown aggregation {op!r}")


def scatter_pullback(op, M, idx, n, out, dout):
    """d(scatter)/dM applied to dout."""
    if op in ("+", "add", "sum"):
        return gather(dout, idx)
    if op == "mean":
        cnt = np.bincount(idx, minlength=n).astype(M.dtype)
        return gather(dout / np.where(cnt > 0, cnt, 1.0), idx)
    if op in ("max", "min"):  # NNlib: every entry equal to the extremum receives the gradient
        return gather(dout, idx) * (M == gather(out, idx))
    if op in ("*", "mul"):
        # every entry receives dout times the product of the OTHER entries of its destination (what NNlib's pullback of scatter(*)
        # forms entry by entry; dout * out / M wherever no entry is zero).  With zeros: the one zero entry of a destination gets the
        # product of the rest, two zeros leave nothing.
        nz = np.where(M == 0, 1.0, M)
        p_nz = gather(scatter("*", nz, idx, n), idx)
        zeros = gather(scatter("+", (M == 0).astype(M.dtype), idx, n), idx)
        others = np.where(M != 0, np.where(zeros == 0, p_nz / nz, 0.0), np.where(zeros == 1, p_nz, 0.0))
        return gather(dout, idx) * others
    raise ValueError(op)


def propagate(f, g, aggr, xi=None, xj=None, e=None):
    """[UPSTREAM] propagate = aggregate_neighbors(g, aggr, apply_edges(f, g, xi, xj, e)).
    xi/xj may be arrays or dicts of arrays (NamedTuples)."""
    ga = lambda x, idx: None if x is None else (
        {k: gather(v, idx) for k, v in x.items()} if isinstance(x, dict) else gather(x, idx))
    m = f(ga(xi, g.t), ga(xj, g.s), e)
    return scatter(aggr, m, g.t, g.num_nodes)


# --------------------------------------------------------------------------------------------
# Lux Dense / Chain [UPSTREAM Lux 0.4]:  y = act.(W x .+ b);  params: list of dicts
# --------------------------------------------------------------------------------------------


def mlp_forward(layers, x):
    """layers: [{'weight': (out,in), 'bias': (out,1) or None, 'act': name}, ...]"""
    cache = []
    for L in layers:
        z = L["weight"] @ x
        if L.get("bias") is not None:
            z = z + L["bias"].reshape(-1, 1)
        cache.append((x, z))
        x = act(L.get("act", "identity"), z)
    return x, cache


def mlp_backward(layers, cache, dy):
    grads = [None] * len(layers)
    for k in range(len(layers) - 1, -1, -1):
        L = layers[k]
        x, z = cache[k]
        dz = dy * dact(L.get("act", "identity"), z)
        g = {"weight": dz @ x.T}
        if L.get("bias") is not None:
            g["bias"] = dz.sum(axis=1, keepdims=True)
        grads[k] = g
        dy = L["weight"].T @ dz
    return dy, grads


# --------------------------------------------------------------------------------------------
# GCNConv   (src/layers.jl:200-239)
# --------------------------------------------------------------------------------------------


def gcn_conv(x, weight, bias, g, activation="identity", add_self_loops_=True,
             use_edge_weight=False, edge_weight=None):
    """Returns (y, cache).  Follows the op order of src/layers.jl:210-238 line by line."""
    T = x.dtype
    if edge_weight is not None:
        edge_weight = np.asarray(edge_weight, T)
        assert edge_weight.size == g.num_edges, \
            f"Wrong number of edge weights (expected {g.num_edges} but given {edge_weight.size})"  # :207
    if add_self_loops_:                                                       # :210-218
        g = add_self_loops(g)
        if edge_weight is not None:
            edge_weight = np.concatenate([edge_weight, np.ones(g.num_nodes, T)])
    Dout, Din = weight.shape
    x_in = x
    if Dout < Din:                                                            # :220-223
        x = weight @ x
    d = degree_in(g, T, edge_weight)                                          # :224
    with np.errstate(divide="ignore"):
        c = 1.0 / np.sqrt(d)                                                  # :225
    x = x * c[None, :]                                                        # :226
    if edge_weight is not None:                                               # :227-233
        w = edge_weight
    elif use_edge_weight:
        w = g.edge_weight.astype(T)
    else:
        w = None
    msg = (lambda xi, xj, e: xj) if w is None else (lambda xi, xj, e: xj * w[None, :])
    x = propagate(msg, g, "+", xj=x)
    x = x * c[None, :]                                                        # :234
    x3 = x
    if Dout >= Din:                                                           # :235-237
        x = weight @ x
    z = x + (0.0 if bias is None else bias.reshape(-1, 1))                    # :238
    y = act(activation, z)
    cache = dict(g=g, c=c, w=w, x_in=x_in, x3=x3, z=z, weight=weight, bias=bias,
                 activation=activation, ew_arg=edge_weight is not None, self_loops=add_self_loops_)
    return y, cache


def gcn_conv_backward(cache, dy):
    """VJP of gcn_conv w.r.t. x, weight, bias (c and edge weights carry no gradient)."""
    g, c, w, W = cache["g"], cache["c"], cache["w"], cache["weight"]
    Dout, Din = W.shape
    dz = dy * dact(cache["activation"], cache["z"])
    grads = {}
    if cache["bias"] is not None:
        grads["bias"] = dz.sum(axis=1, keepdims=True)

    def agg_T(d):  # pullback of  x -> ((x .* c') A) .* c'
        d = d * c[None, :]
        d = gather(d, g.t)               # pullback of scatter(+) at t
        if w is not None:
            d = d * w[None, :]
        d = scatter("+", d, g.s, g.num_nodes)   # pullback of gather at s
        return d * c[None, :]

    if Dout >= Din:
        grads["weight"] = dz @ cache["x3"].T
        grads["x"] = agg_T(W.T @ dz)
    else:
        dxw = agg_T(dz)
        grads["weight"] = dxw @ cache["x_in"].T
        grads["x"] = W.T @ dxw
    if cache.get("ew_arg"):
        # The `edge_weight` ARGUMENT (src/layers.jl:200-233) is differentiable in the reference: through e_mul_xj (:228) and through
        # the weighted degree (:224).  With xp = the array entering the propagation (x, or W x when Dout < Din), x1 = xp .* c',
        # x2 = sum_e w_e x1[:, s_e], x3 = x2 .* c', g3 = dL/dx3:
        #   direct      dL/dw_e  = (g3 .* c')[:, t_e] . x1[:, s_e]
        #   via degree  d_i = sum_{e: t_e = i} w_e,  c = d^(-1/2):  dL/dd_i = -1/2 c_i^3 dL/dc_i,
        #               dL/dc_i = g3[:, i] . x2[:, i]  (x3 = c x2)  +  g1[:, i] . xp[:, i]  (x1 = xp c),  g1 = dL/dx1
        # The ones appended for the self loops are constants: only the first num_edges entries are returned.
        xp = cache["x_in"] if Dout >= Din else W @ cache["x_in"]
        g3 = W.T @ dz if Dout >= Din else dz
        g2 = g3 * c[None, :]
        x1 = xp * c[None, :]
        dw = (gather(g2, g.t) * gather(x1, g.s)).sum(axis=0)
        with np.errstate(invalid="ignore", divide="ignore"):
            x2 = cache["x3"] / c[None, :]
        g1 = scatter("+", gather(g2, g.t) * w[None, :], g.s, g.num_nodes)
        dc = (g3 * x2).sum(axis=0) + (g1 * xp).sum(axis=0)
        dd = -0.5 * c ** 3 * dc
        dw = dw + dd[g.t]
        n_orig = g.num_edges - (g.num_nodes if cache["self_loops"] else 0)
        grads["edge_weight"] = dw[:n_orig]
    return grads


# --------------------------------------------------------------------------------------------
# edge-MLP layers: ExplicitEdgeConv (:94-112), VMHConv (:308-332), MPPDEConv (:390-422)
# --------------------------------------------------------------------------------------------


def _as_named(x):
    return x if isinstance(x, dict) else {"preservedname": x}   # :94-96, :308-310


def _merge(a, b):  # Base.merge(::NamedTuple, ::NamedTuple): order of a, then new keys of b
    out = dict(a)
    out.update(b)
    return out


def _vcat(arrs, ncols, dtype):
    arrs = list(arrs)
    return np.concatenate(arrs, axis=0) if arrs else np.zeros((0, ncols), dtype)


def _split_rows(d, sizes):
    out, o = [], 0
    for n in sizes:
        out.append(d[o:o + n])
        o += n
    return out


def explicit_edge_conv(x, phi, g, aggr="mean"):
    """h'_i = aggr_j phi([h_i...; h_j...; x_j - x_i])   (src/layers.jl:98-112)"""
    xn = _as_named(x)
    xs = _merge(xn, g.ndata)                                                   # :110
    hk = [k for k in xs if k != "x"]                                           # drop(xi, :x)
    dt = next(iter(xn.values())).dtype
    E = g.num_edges
    xi = {k: gather(v, g.t) for k, v in xs.items()}
    xj = {k: gather(v, g.s) for k, v in xs.items()}
    inp = _vcat([xi[k] for k in hk] + [xj[k] for k in hk] + [xj["x"] - xi["x"]], E, dt)  # :106
    m, mc = mlp_forward(phi, inp)
    y = scatter(aggr, m, g.t, g.num_nodes)
    return y, dict(g=g, xn=xn, xs=xs, hk=hk, inp=inp, m=m, mc=mc, y=y, aggr=aggr, phi=phi)


def explicit_edge_conv_backward(c, dy):
    g, hk, xs = c["g"], c["hk"], c["xs"]
    dm = scatter_pullback(c["aggr"], c["m"], g.t, g.num_nodes, c["y"], dy)
    dinp, gphi = mlp_backward(c["phi"], c["mc"], dm)
    sizes = [xs[k].shape[0] for k in hk]
    blocks = _split_rows(dinp, sizes + sizes)
    dx = {}
    for n, k in enumerate(hk):
        if k in c["xn"]:
            dx[k] = scatter("+", blocks[n], g.t, g.num_nodes) + \
                scatter("+", blocks[len(hk) + n], g.s, g.num_nodes)
    return {"x": dx["preservedname"] if list(c["xn"]) == ["preservedname"] else dx, "phi": gphi}


def vmh_conv(x, phi, gamma, g, aggr="mean"):
    """m_i = aggr_j phi([h_i...; (h_j - h_i)...; x_j - x_i]);  h' = gamma([h_i...; m_i])
    (src/layers.jl:312-332)"""
    xn = _as_named(x)
    xs = _merge(xn, g.ndata)                                                   # :324
    hk = [k for k in xs if k != "x"]                                           # :315
    dt = next(iter(xn.values())).dtype
    E = g.num_edges
    xi = {k: gather(v, g.t) for k, v in xs.items()}
    xj = {k: gather(v, g.s) for k, v in xs.items()}
    inp = _vcat([xi[k] for k in hk] + [xj[k] - xi[k] for k in hk] + [xj["x"] - xi["x"]], E, dt)  # :316
    m, mc = mlp_forward(phi, inp)
    agg = scatter(aggr, m, g.t, g.num_nodes)
    ginp = _vcat(list(xn.values()) + [agg], g.num_nodes, dt)                   # :328
    y, gc = mlp_forward(gamma, ginp)
    return y, dict(g=g, xn=xn, xs=xs, hk=hk, m=m, mc=mc, agg=agg, gc=gc, aggr=aggr,
                   phi=phi, gamma=gamma)


def vmh_conv_backward(c, dy):
    g, hk, xs, xn = c["g"], c["hk"], c["xs"], c["xn"]
    dginp, ggamma = mlp_backward(c["gamma"], c["gc"], dy)
    sizes_x = [v.shape[0] for v in xn.values()]
    parts = _split_rows(dginp, sizes_x + [c["agg"].shape[0]])
    dx = {k: parts[n].copy() for n, k in enumerate(xn)}
    dagg = parts[-1]
    dm = scatter_pullback(c["aggr"], c["m"], g.t, g.num_nodes, c["agg"], dagg)
    dinp, gphi = mlp_backward(c["phi"], c["mc"], dm)
    sizes = [xs[k].shape[0] for k in hk]
    blocks = _split_rows(dinp, sizes + sizes)
    for n, k in enumerate(hk):
        if k in xn:
            di, dd = blocks[n], blocks[len(hk) + n]
            dx[k] = dx[k] + scatter("+", di - dd, g.t, g.num_nodes) + scatter("+", dd, g.s, g.num_nodes)
    return {"x": dx["preservedname"] if list(xn) == ["preservedname"] else dx,
            "phi": gphi, "gamma": ggamma}


def mppde_conv(x, phi, psi, g, aggr="mean"):
    """m_i = aggr_j phi([h_i; h_j; d_i - d_j; e_ij; theta]); h' = psi([h_i; m_i; theta])
    (src/layers.jl:390-422)"""
    dt = x.dtype
    N, E, G = g.num_nodes, g.num_edges, g.num_graphs
    theta = _vcat(g.gdata.values(), G, dt)                                      # :397
    d = _vcat(g.ndata.values(), N, dt)                                          # :403-405
    e = _vcat(g.edata.values(), E, dt)                                          # :407
    hi, hj = gather(x, g.t), gather(x, g.s)                                     # :408
    di, dj = gather(d, g.t), gather(d, g.s)
    th_e = np.repeat(theta, E // G, axis=1) if G else theta                     # :410 repeat(inner=(1,E÷G))
    inp = np.concatenate([hi, hj, di - dj, e, th_e], axis=0).astype(
        np.result_type(dt, d.dtype, e.dtype, theta.dtype))                      # :409
    m, mc = mlp_forward(phi, inp)
    agg = scatter(aggr, m, g.t, N)                                              # :416
    th_n = np.repeat(theta, N // G, axis=1) if G else theta
    pinp = np.concatenate([x, agg, th_n], axis=0)                               # :418
    y, pc = mlp_forward(psi, pinp)
    return y, dict(g=g, h=x.shape[0], m=m, mc=mc, agg=agg, pc=pc, aggr=aggr, phi=phi, psi=psi)


def mppde_conv_backward(c, dy):
    g, h = c["g"], c["h"]
    dpinp, gpsi = mlp_backward(c["psi"], c["pc"], dy)
    dx = dpinp[:h].copy()
    dagg = dpinp[h:h + c["agg"].shape[0]]
    dm = scatter_pullback(c["aggr"], c["m"], g.t, g.num_nodes, c["agg"], dagg)
    dinp, gphi = mlp_backward(c["phi"], c["mc"], dm)
    dx += scatter("+", dinp[:h], g.t, g.num_nodes) + scatter("+", dinp[h:2 * h], g.s, g.num_nodes)
    return {"x": dx, "phi": gphi, "psi": gpsi}     # theta / ndata / edata: no gradient (:397,:418)


# --------------------------------------------------------------------------------------------
# GNOConv (src/layers.jl:509-547)
# --------------------------------------------------------------------------------------------


def gno_conv(x, phi, lin_weight, lin_bias, g, in_chs, out_chs, activation="identity", aggr="mean"):
    dt = x.dtype
    N, E = g.num_nodes, g.num_edges
    sfeat = _vcat(g.ndata.values(), N, dt)
    si, sj = gather(sfeat, g.t), gather(sfeat, g.s)                             # :517-519
    e = _vcat(g.edata.values(), E, dt)                                          # :521
    kin = np.concatenate([si, sj, e], axis=0)
    Wk, kc = mlp_forward(phi, kin)                                              # :523
    hj = gather(x, g.s)                                                         # :526
    # reshape(W, :, in, E) column-major: K[o, i, e] = Wk[o + out*i, e]          # :527
    K = Wk.reshape(in_chs, out_chs, E).transpose(1, 0, 2)
    m = np.einsum("oie,ie->oe", K, hj)                                          # :529 batched_mul
    agg = scatter(aggr, m, g.t, N)                                              # :534
    z = lin_weight @ x + agg                                                    # :541-547
    if lin_bias is not None:
        z = z + lin_bias.reshape(-1, 1)
    y = act(activation, z)                                                      # :536
    return y, dict(g=g, x=x, K=K, hj=hj, m=m, agg=agg, z=z, kc=kc, phi=phi, W=lin_weight,
                   b=lin_bias, activation=activation, aggr=aggr, in_chs=in_chs, out_chs=out_chs)


def gno_conv_backward(c, dy):
    g = c["g"]
    dz = dy * dact(c["activation"], c["z"])
    grads = {"weight": dz @ c["x"].T}
    if c["b"] is not None:
        grads["bias"] = dz.sum(axis=1, keepdims=True)
    dx = c["W"].T @ dz
    dm = scatter_pullback(c["aggr"], c["m"], g.t, g.num_nodes, c["agg"], dz)
    dK = np.einsum("oe,ie->oie", dm, c["hj"])
    dhj = np.einsum("oie,oe->ie", c["K"], dm)
    dx = dx + scatter("+", dhj, g.s, g.num_nodes)
    dWk = dK.transpose(1, 0, 2).reshape(c["in_chs"] * c["out_chs"], -1)
    _, gphi = mlp_backward(c["phi"], c["kc"], dWk)
    grads["x"] = dx
    grads["phi"] = gphi
    return grads


# --------------------------------------------------------------------------------------------
# SpectralConv (src/layers.jl:639-662) -- the only layer with a reference known-answer test
# --------------------------------------------------------------------------------------------


def spectral_graph(n, dtype=np.float64):
    """initialstates(::SpectralConv) :639-648 -- complete digraph, edata e = x[t] - x[s]."""
    x = np.linspace(0.0, 2.0 * np.pi, n + 1, dtype=np.float64)[1:]
    s, t = np.nonzero(~np.eye(n, dtype=bool))       # lexicographic (src, dst), as Graphs.edges
    diff = (x[t] - x[s]).astype(dtype)
    return Graph(s, t, num_nodes=n, edata=diff.reshape(1, -1), index_base=0)


def spectral_conv(u, g, n):
    """y_i = sum_j 1/2 cos(n e/2) cot(e/2) u_j  with e = x_t - x_s   (:652-657)"""
    vec = (u.ndim == 1)
    x = u.reshape(1, -1) if vec else u
    e = g.edata["e"]
    msg = lambda xi, xj, ee: np.cos(e * n / 2) * (1.0 / np.tan(e / 2)) / 2 * xj
    y = propagate(msg, g, "+", xj=x, e=e)
    return y.reshape(-1) if vec else y


# --------------------------------------------------------------------------------------------
# GAT-style softmax aggregation [UPSTREAM GraphNeuralNetworks.jl GATConv; the reference only
# re-exports the primitive softmax_edge_neighbors, src/NeuralGraphPDE.jl:7]
# --------------------------------------------------------------------------------------------


def gat_conv(x, weight, a, bias, g, heads, out_chs, activation="identity", negative_slope=0.2,
             add_self_loops_=True, concat=True):
    if add_self_loops_:
        g = add_self_loops(g)
    N, E, H, C = g.num_nodes, g.num_edges, heads, out_chs
    Wx = (weight @ x)                                    # (C*H, N), row r = c + C*h (column-major reshape)
    Wx3 = Wx.reshape(H, C, N).transpose(1, 0, 2)          # (C, H, N)
    Wxi, Wxj = Wx3[:, :, g.t], Wx3[:, :, g.s]
    a3 = a.reshape(2 * C, H) if a.ndim == 2 else a
    aWW = np.einsum("ch,che->he", a3[:C], Wxi) + np.einsum("ch,che->he", a3[C:], Wxj)
    logit = np.where(aWW > 0, aWW, negative_slope * aWW)  # leakyrelu
    mx = scatter("max", logit, g.t, N)
    ex = np.exp(logit - mx[:, g.t])
    den = scatter("+", ex, g.t, N)
    alpha = ex / den[:, g.t]                              # softmax_edge_neighbors
    beta = alpha[None, :, :] * Wxj
    out3 = scatter("+", beta.reshape(C * H, E), g.t, N).reshape(C, H, N)
    if concat:
        pre = out3.transpose(1, 0, 2).reshape(H * C, N)   # reshape(x, :, N): row = c + C*h
    else:
        pre = out3.mean(axis=1)
    z = pre + (0.0 if bias is None else bias.reshape(-1, 1))
    y = act(activation, z)
    return y, dict(g=g, x=x, W=weight, a3=a3, Wx3=Wx3, aWW=aWW, alpha=alpha, z=z, H=H, C=C,
                   activation=activation, slope=negative_slope, concat=concat, bias=bias)


def gat_conv_backward(c, dy):
    g, H, C, a3 = c["g"], c["H"], c["C"], c["a3"]
    N = g.num_nodes
    dz = dy * dact(c["activation"], c["z"])
    grads = {}
    if c["bias"] is not None:
        grads["bias"] = dz.sum(axis=1)
    if c["concat"]:
        dout3 = dz.reshape(H, C, N).transpose(1, 0, 2)
    else:
        dout3 = np.repeat(dz[:, None, :], H, axis=1) / H
    Wxj = c["Wx3"][:, :, g.s]
    Wxi = c["Wx3"][:, :, g.t]
    dbeta = dout3[:, :, g.t]                                   # (C,H,E)
    dalpha = np.einsum("che,che->he", dbeta, Wxj)
    dWxj = c["alpha"][None] * dbeta
    # softmax pullback per target segment
    tmp = c["alpha"] * dalpha
    dlogit = tmp - c["alpha"] * scatter("+", tmp, g.t, N)[:, g.t]
    daWW = dlogit * np.where(c["aWW"] > 0, 1.0, c["slope"])
    da = np.concatenate([np.einsum("he,che->ch", daWW, Wxi), np.einsum("he,che->ch", daWW, Wxj)], axis=0)
    dWxi = a3[:C][:, :, None] * daWW[None]
    dWxj = dWxj + a3[C:][:, :, None] * daWW[None]
    E = g.num_edges
    dWx3 = scatter("+", dWxi.reshape(C * H, E), g.t, N) + scatter("+", dWxj.reshape(C * H, E), g.s, N)
    dWx = dWx3.reshape(C, H, N).transpose(1, 0, 2).reshape(H * C, N)
    grads["weight"] = dWx @ c["x"].T
    grads["a"] = da
    grads["x"] = c["W"].T @ dWx
    return grads


# --------------------------------------------------------------------------------------------
# fixed-step explicit Runge-Kutta + discrete adjoint (caller of the hot path: the tutorial
# NeuralODE, docs/src/tutorials/graph_node.md:44-66; BASELINE configs use fixed step counts)
# --------------------------------------------------------------------------------------------

EULER = dict(name="euler", c=[0.0], a=[[]], b=[1.0])

# Tsitouras 2011 5(4) pair as used by OrdinaryDiffEq.Tsit5; with a fixed step only the 5th-order
# weights are needed and they equal row a7 (FSAL), so the scheme is a 6-stage explicit RK.
_TS_A = [
    [],
    [0.161],
    [-0.008480655492356989, 0.335480655492357],
    [2.8971530571054935, -6.359448489975075, 4.3622954328695815],
    [5.325864828439257, -11.748883564062828, 7.4955393428898365, -0.09249506636175525],
    [5.86145544294642, -12.92096931784711, 8.159367898576159, -0.071584973281401,
     -0.028269050394068383],
]
_TS_B = [0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742, -3.290069515436081,
         2.324710524099774]
TSIT5 = dict(name="tsit5", c=[0.0, 0.161, 0.327, 0.9, 0.9800255409045097, 1.0], a=_TS_A, b=_TS_B)
TABLEAUS = {"euler": EULER, "tsit5": TSIT5}


def rk_solve(rhs, u0, tableau, dt, nsteps):
    """rhs(u) -> (k, cache).  Returns u(T) and the tape needed by rk_adjoint."""
    a, b = tableau["a"], tableau["b"]
    S = len(b)
    u = u0
    tape = []
    for _ in range(nsteps):
        ks, caches = [], []
        for i in range(S):
            U = u
            for j in range(i):
                if a[i][j] != 0.0:
                    U = U + (dt * a[i][j]) * ks[j]
            k, cch = rhs(U)
            ks.append(k)
            caches.append(cch)
        for i in range(S):
            u = u + (dt * b[i]) * ks[i]
        tape.append(caches)
    return u, tape


def rk_adjoint(rhs_vjp, tape, lam, tableau, dt, accumulate):
    """Discrete adjoint of rk_solve.  rhs_vjp(cache, kbar) -> (ubar, param_grads);
    accumulate(param_grads) sums parameter gradients.  Returns dL/du0."""
    a, b = tableau["a"], tableau["b"]
    S = len(b)
    for caches in reversed(tape):
        ubars = [None] * S
        for i in range(S - 1, -1, -1):
            kbar = (dt * b[i]) * lam
            for j in range(i + 1, S):
                if a[j][i] != 0.0:
                    kbar = kbar + (dt * a[j][i]) * ubars[j]
            ubars[i], pg = rhs_vjp(caches[i], kbar)
            accumulate(pg)
        for i in range(S):
            lam = lam + ubars[i]
    return lam


def gcn2_rhs(params, g, activation="relu"):
    """RHS of the graph neural ODE of graph_node.md:78: Chain(GCNConv(D=>D, act), GCNConv(D=>D, act))."""
    def rhs(u):
        y1, c1 = gcn_conv(u, params[0]["weight"], params[0]["bias"], g, activation)
        y2, c2 = gcn_conv(y1, params[1]["weight"], params[1]["bias"], g, activation)
        return y2, (c1, c2)

    def vjp(cache, kbar):
        c1, c2 = cache
        g2 = gcn_conv_backward(c2, kbar)
        g1 = gcn_conv_backward(c1, g2["x"])
        return g1["x"], [dict(weight=g1["weight"], bias=g1.get("bias")),
                         dict(weight=g2["weight"], bias=g2.get("bias"))]
    return rhs, vjp


def gcn2_node_loss_and_grads(params, g, u0, tableau, dt, nsteps, activation="relu"):
    """loss = sum(u(T)); returns (uT, du0, [grads layer1, grads layer2])."""
    rhs, vjp = gcn2_rhs(params, g, activation)
    uT, tape = rk_solve(rhs, u0, tableau, dt, nsteps)
    acc = [dict(weight=np.zeros_like(p["weight"]), bias=np.zeros_like(p["bias"])) for p in params]

    def accumulate(pg):
        for A, G in zip(acc, pg):
            A["weight"] += G["weight"]
            if G["bias"] is not None:
                A["bias"] += G["bias"].reshape(A["bias"].shape)
    du0 = rk_adjoint(vjp, tape, np.ones_like(uT), tableau, dt, accumulate)
    return uT, du0, acc


def gcn2_node_prescaled(params, g, u0, tableau, dt, nsteps, activation="relu"):
    """The same solve and discrete adjoint written in the variables the HIP solver plan keeps (DESIGN.md section 5,
    "pre-scaled pipeline"): every feature array multiplied by c[row] = 1/sqrt(degree),  u~ = u .* c',  y~ = y .* c',
    adjoint products G~ = G .* c'.  With A~ = A + I (0/1 adjacency with self loops, A~[i, j] = 1 for an edge j -> i):
        forward   a = (x~ A~') .* c'        (plain sums of stored rows, one factor c_i per row)
                  z = W a + b ,  y~ = act(z) .* c'
        pullback  dz = (dy~ .* c') .* act'(z) ,  dW = dz a' ,  db = sum dz ,  G~ = (W' dz) .* c' ,  dx~ = G~ A~
    Entry u~0 = u0 .* c', exit u(T) = u~(T) ./ c', seed lambda~ = dL/du(T) ./ c', du0 = lambda~0 .* c'.
    Checker of the algebra only (dense adjacency: small graphs); returns (uT, du0, [grads layer1, grads layer2])."""
    ga = add_self_loops(g)
    n = g.num_nodes
    At = np.zeros((n, n))
    np.add.at(At, (ga.t, ga.s), 1.0)                      # At[i, j] = number of edges j -> i (self loops included)
    c = (1.0 / np.sqrt(At.sum(axis=1)))[None, :]          # c' as a row: scales columns (nodes)

    def layer(xs, p):
        a = (xs @ At.T) * c
        z = p["weight"] @ a + p["bias"]
        return act(activation, z) * c, (a, z, p)

    def layer_vjp(cache, dys):
        a, z, p = cache
        dz = (dys * c) * dact(activation, z)
        return ((p["weight"].T @ dz) * c) @ At, dict(weight=dz @ a.T, bias=dz.sum(axis=1, keepdims=True))

    def rhs(us):
        y1, c1 = layer(us, params[0])
        y2, c2 = layer(y1, params[1])
        return y2, (c1, c2)

    def vjp(cache, kbar):
        d1, g2 = layer_vjp(cache[1], kbar)
        d0, g1 = layer_vjp(cache[0], d1)
        return d0, [g1, g2]

    uTs, tape = rk_solve(rhs, u0 * c, tableau, dt, nsteps)
    acc = [dict(weight=np.zeros_like(p["weight"]), bias=np.zeros_like(p["bias"])) for p in params]

    def accumulate(pg):
        for A, G in zip(acc, pg):
            A["weight"] += G["weight"]
            A["bias"] += G["bias"].reshape(A["bias"].shape)
    lam0 = rk_adjoint(vjp, tape, np.ones_like(uTs) / c, tableau, dt, accumulate)
    return uTs / c, lam0 * c, acc


# --------------------------------------------------------------------------------------------
# Derived-graph handle (integer / byte work; the HIP library's ngpde_graph_t, include/ngpde.h).
# Restates what the kernels are specified to read: CSR lists by target / by source in COO order
# inside a row (NNlib's scatter order), the cross positions, the GCN coefficients of
# src/layers.jl:210-226 in float32, the 32-row tile schedule and the per-tile halo lists.
# Pure-Python loops: small cases only.  Parity bar: bit-exact.
# --------------------------------------------------------------------------------------------

TILE_ROWS, HALO_CAP, SLOT_WIDTH, ELL_WIDTH = 32, 96, 32, 16


def csr_stable(key, other, n):
    """rowptr [n+1], col [m], eid [m]: entries of row r are the COO positions e with key[e] == r, in increasing e"""
    key = np.asarray(key, dtype=np.int64)
    eid = np.argsort(key, kind="stable").astype(np.int32)
    rowptr = np.zeros(n + 1, dtype=np.int32)
    np.add.at(rowptr, key + 1, 1)
    return np.cumsum(rowptr, dtype=np.int32), np.asarray(other, dtype=np.int32)[eid], eid


def locality_order(n, rp_in, col_in, rp_out, col_out, tile=TILE_ROWS):
    """clusters of `tile` nodes grown breadth-first (in- then out-neighbours, list order), emitted in a breadth-first
    sweep over clusters: seed = oldest untaken frontier node, else the lowest untaken index"""
    from collections import deque
    order, taken, frontier, next_free = [], np.zeros(n, dtype=bool), deque(), 0

    def neighbours(v):
        yield from col_in[rp_in[v]:rp_in[v + 1]]
        yield from col_out[rp_out[v]:rp_out[v + 1]]

    while len(order) < n:
        local, head = [], 0
        while len(local) < tile and len(order) + len(local) < n:
            if head == len(local):
                seed = -1
                while frontier:
                    v = frontier.popleft()
                    if not taken[v]:
                        seed = v
                        break
                if seed < 0:
                    while taken[next_free]:
                        next_free += 1
                    seed = next_free
                taken[seed] = True
                local.append(int(seed))
            v = local[head]
            head += 1
            for w in neighbours(v):
                if taken[w]:
                    continue
                if len(local) < tile:
                    taken[w] = True
                    local.append(int(w))
                else:
                    frontier.append(int(w))
        for v in local[head:]:
            for w in neighbours(v):
                if not taken[w]:
                    frontier.append(int(w))
        order.extend(local)
    return np.asarray(order, dtype=np.int32)


def derived_graph(s, t, n, order=None, add_self_loops_=True, edge_weight=None, weighted_degree=False):
    """dict of every array of the handle, per direction 't' (lists by target) / 's' (by source)"""
    s, t = np.asarray(s, dtype=np.int64), np.asarray(t, dtype=np.int64)
    m = s.size
    out = {}
    rp_t, col_t, eid_t = csr_stable(t, s, n)
    rp_s, col_s, eid_s = csr_stable(s, t, n)
    pos_t, pos_s = np.empty(m, np.int32), np.empty(m, np.int32)
    pos_t[eid_t] = np.arange(m, dtype=np.int32)
    pos_s[eid_s] = np.arange(m, dtype=np.int32)
    if order is None:
        order = locality_order(n, rp_t, col_t, rp_s, col_s)
    order = np.asarray(order, dtype=np.int32)
    out["order"] = order
    w = None if edge_weight is None else np.asarray(edge_weight, dtype=np.float32)
    deg = np.zeros(n, dtype=np.float32)
    for i in range(n):                                  # float32 running sum in COO order, then the self loop's 1
        acc = np.float32(0.0)
        for p in range(rp_t[i], rp_t[i + 1]):
            acc = np.float32(acc + (w[eid_t[p]] if weighted_degree else np.float32(1.0)))
        deg[i] = np.float32(acc + np.float32(1.0 if add_self_loops_ else 0.0))
    with np.errstate(divide="ignore"):
        c = (np.float32(1.0) / np.sqrt(deg, dtype=np.float32)).astype(np.float32)
    out["c"] = c
    n_tiles = (n + TILE_ROWS - 1) // TILE_ROWS
    n_sched = n_tiles * TILE_ROWS
    for tag, rp, col, eid, xp in (("t", rp_t, col_t, eid_t, pos_s[eid_t]), ("s", rp_s, col_s, eid_s, pos_t[eid_s])):
        coef = ((w[eid] if w is not None else np.float32(1.0)) * c[col]).astype(np.float32)
        ent = np.stack([col.astype(np.int32), coef.view(np.int32)], axis=1) if m else np.zeros((0, 2), np.int32)
        sched = np.zeros((n_sched, 4), dtype=np.int32)
        sched[:, 0] = -1
        ell = np.zeros((n_sched, ELL_WIDTH, 2), dtype=np.int32)
        for k in range(n):
            v = order[k]
            d = rp[v + 1] - rp[v]
            sched[k] = (v, rp[v], d, c[v:v + 1].view(np.int32)[0])
            ell[k, :min(d, ELL_WIDTH)] = ent[rp[v]:rp[v] + min(d, ELL_WIDTH)]
        halo = np.zeros((n_tiles, HALO_CAP, 2), dtype=np.int32)
        info = np.zeros((n_tiles, 2), dtype=np.int32)
        slots = np.full((n_sched, SLOT_WIDTH), HALO_CAP, dtype=np.uint8)
        slot_w = np.zeros((n_sched, SLOT_WIDTH), dtype=np.float32) if w is not None else None
        for tl in range(n_tiles):
            slot_of, count, fits = {}, TILE_ROWS, True
            for k in range(TILE_ROWS):                   # own rows: slot k = k-th row of the tile
                pos = tl * TILE_ROWS + k
                if pos < n:
                    v = int(order[pos])
                    slot_of[v] = k
                    halo[tl, k] = (v, c[v:v + 1].view(np.int32)[0])
            rows = [int(order[tl * TILE_ROWS + k]) for k in range(TILE_ROWS) if tl * TILE_ROWS + k < n]
            if any(rp[v + 1] - rp[v] > SLOT_WIDTH for v in rows):
                fits = False
            else:
                tile_slots = []
                for v in rows:                           # other columns: order of first appearance
                    for p in range(rp[v], rp[v + 1]):
                        u = int(col[p])
                        if u not in slot_of:
                            slot_of[u] = count
                            if count < HALO_CAP:
                                halo[tl, count] = (u, c[u:u + 1].view(np.int32)[0])
                            count += 1
                        tile_slots.append(slot_of[u])
                fits = count <= HALO_CAP
                if fits:
                    i = 0
                    for k, v in enumerate(rows):
                        for j in range(rp[v + 1] - rp[v]):
                            slots[tl * TILE_ROWS + k, j] = tile_slots[i]
                            if slot_w is not None:
                                slot_w[tl * TILE_ROWS + k, j] = w[eid[rp[v] + j]]
                            i += 1
            info[tl, 0] = count if fits else 0
        out[tag] = dict(rowptr=rp, col=col, eid=eid, xpos=xp.astype(np.int32), ent=ent, sched=sched, ell=ell, halo=halo,
                        tile_info=info, slots=slots, slot_w=slot_w, halo_ok=bool(n_tiles > 0 and (info[:, 0] > 0).all()))
    return out


# --------------------------------------------------------------------------------------------
# Neighbour search: GNNGraphs.radius_graph / knn_graph [UPSTREAM GraphNeuralNetworks.jl, re-exported at
# src/NeuralGraphPDE.jl:4; NearestNeighbors.jl inrange / knn underneath], restated as brute force over all pairs.
# float32 arithmetic spelled out so that the device search can be compared bit for bit: d2 = sum over coordinates,
# in order, of (p_i - p_j)^2, each operation rounded to float32; neighbours iff d2 <= r*r (float32 product).
# Canonical edge order (the reference's is the tree's traversal order): by point, neighbours ascending by index
# (radius) or by (d2, index) (k-NN).  dir="in": neighbours are sources.  parity unpinned (no Julia here): checked
# against scipy's cKDTree in float64 away from the threshold (tests/test_oracle.py).
# --------------------------------------------------------------------------------------------


def _pair_d2_f32(P, Q):
    d2 = np.zeros((P.shape[0], Q.shape[0]), dtype=np.float32)
    for d in range(P.shape[1]):
        df = P[:, None, d] - Q[None, :, d]
        d2 = d2 + df * df
    return d2


def _points_rows(points):
    P = np.asarray(points, dtype=np.float32)
    if P.ndim == 1:
        P = P.reshape(1, -1)
    return np.ascontiguousarray(P.T)          # [n][dim]


def radius_graph(points, r, graph_indicator=None, self_loops=False, dir="in", chunk=1024):
    """(s, t) 0-based.  points (dim x n)."""
    P = _points_rows(points)
    n = P.shape[0]
    gi = None if graph_indicator is None else np.asarray(graph_indicator).reshape(-1)
    r2 = np.float32(r) * np.float32(r)
    me, nb = [], []
    for a in range(0, n, chunk):
        d2 = _pair_d2_f32(P[a:a + chunk], P)
        hit = d2 <= r2
        if gi is not None:
            hit &= gi[a:a + chunk, None] == gi[None, :]
        if not self_loops:
            rows = np.arange(a, min(a + chunk, n))
            hit[rows - a, rows] = False
        i, j = np.nonzero(hit)                 # row-major: by point, neighbours ascending
        me.append(i + a)
        nb.append(j)
    me = np.concatenate(me) if me else np.zeros(0, np.int64)
    nb = np.concatenate(nb) if nb else np.zeros(0, np.int64)
    return (nb, me) if dir == "in" else (me, nb)


def knn_graph(points, k, graph_indicator=None, self_loops=False, dir="in", chunk=1024):
    """(s, t) 0-based, exactly n*k edges; the k smallest (d2, index) pairs per point."""
    P = _points_rows(points)
    n = P.shape[0]
    gi = None if graph_indicator is None else np.asarray(graph_indicator).reshape(-1)
    me, nb = [], []
    for a in range(0, n, chunk):
        d2 = _pair_d2_f32(P[a:a + chunk], P).astype(np.float64)
        if gi is not None:
            d2[gi[a:a + chunk, None] != gi[None, :]] = np.inf
        if not self_loops:
            rows = np.arange(a, min(a + chunk, n))
            d2[rows - a, rows] = np.inf
        idx = np.argsort(d2, axis=1, kind="stable")[:, :k]     # stable: ties by index
        if np.isinf(np.take_along_axis(d2, idx, axis=1)).any():
            raise ValueError("knn_graph: a graph has fewer than k (+1) points")
        me.append(np.repeat(np.arange(a, a + d2.shape[0]), k))
        nb.append(idx.reshape(-1))
    me = np.concatenate(me) if me else np.zeros(0, np.int64)
    nb = np.concatenate(nb) if nb else np.zeros(0, np.int64)
    return (nb, me) if dir == "in" else (me, nb)


def _hilbert2(x, y, bits):
    x, y = x.astype(np.uint64), y.astype(np.uint64)
    d = np.zeros(x.shape, dtype=np.uint64)
    side = np.uint64(1 << bits)
    s = 1 << (bits - 1)
    while s > 0:
        su = np.uint64(s)
        rx = ((x & su) > 0).astype(np.uint64)
        ry = ((y & su) > 0).astype(np.uint64)
        d += su * su * ((np.uint64(3) * rx) ^ ry)
        flip = (ry == 0) & (rx == 1)
        x = np.where(flip, side - np.uint64(1) - x, x)
        y = np.where(flip, side - np.uint64(1) - y, y)
        swap = ry == 0
        x, y = np.where(swap, y, x), np.where(swap, x, y)
        s >>= 1
    return d


def _spread3(v):
    v = v.astype(np.uint64) & np.uint64(0x3ff)
    v = (v | (v << np.uint64(16))) & np.uint64(0x030000ff)
    v = (v | (v << np.uint64(8))) & np.uint64(0x0300f00f)
    v = (v | (v << np.uint64(4))) & np.uint64(0x030c30c3)
    v = (v | (v << np.uint64(2))) & np.uint64(0x09249249)
    return v


def spatial_order(points, graph_indicator=None, id_base=0):
    """node permutation along a space-filling curve (Hilbert 2-D, Morton 3-D, the coordinate 1-D), graph by graph;
    coordinates quantised in float32 exactly as the device does; ties by index (stable sort)."""
    P = _points_rows(points)
    n, dim = P.shape
    if n == 0:
        return np.zeros(0, np.int32)
    bits = {1: 30, 2: 16, 3: 10}[dim]
    lo = P.min(axis=0)
    ext = (P.max(axis=0) - lo).astype(np.float32)
    with np.errstate(divide="ignore"):
        scale = np.where(ext > 0, np.float32(1 << bits) / ext, np.float32(0)).astype(np.float32)
    q = ((P - lo).astype(np.float32) * scale).astype(np.float32)
    q = np.clip(q.astype(np.int64), 0, (1 << bits) - 1)
    if dim == 1:
        code = q[:, 0].astype(np.uint64)
    elif dim == 2:
        code = _hilbert2(q[:, 0], q[:, 1], bits)
    else:
        code = (_spread3(q[:, 0]) << np.uint64(2)) | (_spread3(q[:, 1]) << np.uint64(1)) | _spread3(q[:, 2])
    g = np.zeros(n, np.uint64) if graph_indicator is None else (np.asarray(graph_indicator).reshape(-1) - id_base).astype(np.uint64)
    key = (g << np.uint64(32)) | code
    return np.argsort(key, kind="stable").astype(np.int32)


# --------------------------------------------------------------------------------------------
# Optimiser rules on the flat parameter vector [UPSTREAM Optimisers.jl Adam / Rprop, published update rules;
# reference call sites docs/src/tutorials/graph_node.md:122-129, VMH.md:97].  float32 arithmetic, as the
# reference's Float32 ComponentArray.  parity unpinned (no Julia here); checked against closed forms in tests.
# --------------------------------------------------------------------------------------------


def adam_init(x):
    return dict(m=np.zeros_like(x, dtype=np.float32), v=np.zeros_like(x, dtype=np.float32), t=0)


def adam_step(x, g, state, eta=0.001, beta=(0.9, 0.999), eps=1e-8):
    f = np.float32
    g = g.astype(np.float32)
    state["t"] += 1
    state["m"] = f(beta[0]) * state["m"] + (f(1) - f(beta[0])) * g
    state["v"] = f(beta[1]) * state["v"] + (f(1) - f(beta[1])) * g * g
    c1 = f(1) - f(np.power(f(beta[0]), f(state["t"])))
    c2 = f(1) - f(np.power(f(beta[1]), f(state["t"])))
    return (x - state["m"] / c1 / (np.sqrt(state["v"] / c2) + f(eps)) * f(eta)).astype(np.float32), state


def rprop_init(x, eta=1e-3):
    return dict(g=np.zeros_like(x, dtype=np.float32), step=np.full_like(x, eta, dtype=np.float32))


def rprop_step(x, g, state, ell=(0.5, 1.2), gamma=(1e-6, 50.0)):
    f = np.float32
    g = g.astype(np.float32)
    p = state["g"] * g
    s = state["step"]
    s = np.where(p > 0, np.minimum(s * f(ell[1]), f(gamma[1])), np.where(p < 0, np.maximum(s * f(ell[0]), f(gamma[0])), s)).astype(np.float32)
    keep = np.where(p < 0, f(0), g).astype(np.float32)
    state["step"], state["g"] = s, keep
    return (x - s * np.sign(keep)).astype(np.float32), state
