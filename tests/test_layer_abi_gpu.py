"""The layer-level C entries (include/ngpde.h: ngpde_edge_layer_*, ngpde_gno_layer_*, api_layers.hip) against the same layers composed
from the primitives' autograd nodes (tests/composed.py: rounds 1 - 4's host path; the layers themselves are checked against the
float64 oracle in test_mp_gpu.py / test_configs_gpu.py).  One call per layer and one per pullback must give the composed path's bits: the entry
only sequences the primitives' launches (/root/reference/src/layers.jl:94-112, :308-332, :390-422)."""
import ctypes as C

import numpy as np
import pytest
import torch

import ngpde_amd as ng
from ngpde_amd import _lib, synth as S
import composed
from test_mp_gpu import grad_leaves, prep, rgraph

pytestmark = pytest.mark.gpu
DEV = "cuda"


def spatial(N, seed, ndata_extra=0, edata=0, gdata=0, n_graphs=1):
    pts, s, t = S.closest_pairs_graph(N, 3 * N, seed=seed)
    rng = np.random.default_rng(seed)
    nd = {"x": pts.T.astype(np.float32).copy()}
    for k in range(ndata_extra):
        nd[f"f{k}"] = rng.normal(size=(1 + k, N)).astype(np.float32)
    kw = dict(ndata=nd)
    if edata:
        kw["edata"] = {"e": rng.normal(size=(edata, s.size)).astype(np.float32)}
    if gdata:
        kw["gdata"] = {"θ": rng.normal(size=(gdata, n_graphs)).astype(np.float32)}
    return ng.GNNGraph(s, t, num_nodes=N, index_base=0, num_graphs=n_graphs, **kw)


def mesh_batch(n, traj, seed):
    s, t = S.periodic_mesh_batch(n, traj)
    N = n * traj
    rng = np.random.default_rng(seed)
    return ng.GNNGraph(s, t, num_nodes=N, index_base=0, num_graphs=traj,
                       ndata={"u": rng.random((1, N)).astype(np.float32), "x": np.tile(np.arange(n) / n, traj)[None, :].astype(np.float32)},
                       gdata={"θ": rng.random((2, traj)).astype(np.float32)})


def both_ways(layer, x, seed, monkeypatch, training=True):
    """(y, dx or [dx...], parameter gradients) through the layer-level entry and through the composed path"""
    ps0, st = ng.setup(seed, layer)
    outs = []
    for call in (lambda xs, ps: layer(xs, ps, st), lambda xs, ps: composed.apply(layer, xs, ps, st)):
        ps = prep(ps0, seed)
        xs = {k: v.detach().clone().requires_grad_(training) for k, v in x.items()} if isinstance(x, dict) else x.detach().clone().requires_grad_(training)
        if training:
            y, _ = call(xs, ps)
            R = torch.as_tensor(np.random.default_rng(seed + 1).normal(size=tuple(y.shape)).astype(np.float32), device=DEV)
            (y * R).sum().backward()
            gx = [v.grad for v in xs.values()] if isinstance(xs, dict) else [xs.grad]
            outs.append([y.detach()] + gx + [p.grad for p in grad_leaves(ps)])
        else:
            with torch.no_grad():
                outs.append([call(xs, ps)[0]])
    return outs


def assert_same(outs):
    a, b = outs
    assert len(a) == len(b)
    for k, (u, v) in enumerate(zip(a, b)):
        assert u is not None and v is not None, k
        # bit for bit (max / min of an empty neighbourhood is -inf / +inf in both, as NNlib's scatter leaves it)
        assert u.shape == v.shape and torch.equal(u.contiguous().view(torch.int32), v.contiguous().view(torch.int32)), \
            f"output {k}: max diff {float((u - v).abs().nan_to_num(0.0).max()):.3e}"


@pytest.mark.parametrize("aggr,widths,extra", [("mean", (16, 8), 0), ("+", (24,), 1), ("max", (12, 20, 8), 0), ("*", (8,), 0), ("mean", (7, 5), 2)])
def test_edgeconv_entry_equals_the_composed_layer(aggr, widths, extra, monkeypatch):
    g = spatial(700, 3, ndata_extra=extra)
    dh = 6 + sum(1 + k for k in range(extra))
    dims = (2 * dh + 2,) + widths
    acts = ["tanh", "swish", "relu"]
    phi = ng.Chain(*[ng.Dense(dims[l], dims[l + 1], acts[l % 3] if l + 1 < len(widths) else "identity") for l in range(len(widths))])
    layer = ng.ExplicitEdgeConv(phi, initialgraph=g, aggr=aggr)
    x = torch.randn(6, 700, device=DEV)
    assert_same(both_ways(layer, x, 11, monkeypatch))
    assert_same(both_ways(layer, x, 11, monkeypatch, training=False))


@pytest.mark.parametrize("aggr,msg,upd,named", [("mean", (60, 60, 40), (60, 60, 1), False), ("+", (16, 8), (12,), False),
                                                ("mean", (32, 32), (20, 20, 20, 2), True), ("min", (8, 4), (8, 3), False)])
def test_vmh_entry_equals_the_composed_layer(aggr, msg, upd, named, monkeypatch):
    g = spatial(900, 5)
    if named:
        x = {"u": torch.randn(2, 900, device=DEV), "v": torch.randn(3, 900, device=DEV)}
        dh = 5
    else:
        x = torch.randn(1, 900, device=DEV)
        dh = 1
    pd = (2 * dh + 2,) + msg
    gd = (dh + msg[-1],) + upd
    phi = ng.Chain(*[ng.Dense(pd[l], pd[l + 1], "tanh" if l + 1 < len(msg) else "identity") for l in range(len(msg))])
    gam = ng.Chain(*[ng.Dense(gd[l], gd[l + 1], "swish" if l + 1 < len(upd) else "identity") for l in range(len(upd))])
    layer = ng.VMHConv(phi, gam, initialgraph=g, aggr=aggr)
    assert_same(both_ways(layer, x, 21, monkeypatch))
    assert_same(both_ways(layer, x, 21, monkeypatch, training=False))


@pytest.mark.parametrize("h,edata,theta,traj,msg,upd", [(64, 0, True, 4, (64, 64), (64, 64)),      # BASELINE config 4's shape: every fused launch
                                                         (64, 0, True, 160, (64, 64), (64, 64)),    # ... at a size where the pair's pullback is ONE launch
                                                         (64, 3, True, 2, (64, 64), (64, 64)),      # ... with edge features (dE is needed)
                                                         (16, 2, False, 1, (24, 12), (20,)),
                                                         (10, 0, True, 3, (30,), (14, 6)),          # widths outside the fused kernels
                                                         (32, 0, False, 1, (48, 48, 48, 32), (40, 40, 32))])
def test_mppde_entry_equals_the_composed_layer(h, edata, theta, traj, msg, upd, monkeypatch):
    n = 256
    g = mesh_batch(n, traj, 7)
    N = n * traj
    if edata:
        g = ng.GNNGraph(*g.edge_index(0), num_nodes=N, index_base=0, num_graphs=traj, ndata=dict(g.ndata), gdata=dict(g.gdata),
                        edata={"e": np.random.default_rng(3).normal(size=(edata, g.num_edges)).astype(np.float32)})
    if not theta:
        g = ng.GNNGraph(*g.edge_index(0), num_nodes=N, index_base=0, num_graphs=traj, ndata=dict(g.ndata), edata=dict(g.edata))
    dth = 2 if theta else 0
    pd = (2 * h + 2 + edata + dth,) + msg
    ud = (h + msg[-1] + dth,) + upd
    phi = ng.Chain(*[ng.Dense(pd[l], pd[l + 1], "swish") for l in range(len(msg))])
    psi = ng.Chain(*[ng.Dense(ud[l], ud[l + 1], "swish" if l + 1 < len(upd) else "identity") for l in range(len(upd))])
    layer = ng.MPPDEConv(phi, psi, initialgraph=g)
    x = torch.randn(N, h, device=DEV).T
    assert_same(both_ways(layer, x, 31, monkeypatch))
    assert_same(both_ways(layer, x, 31, monkeypatch, training=False))


@pytest.mark.parametrize("switch", ["NGPDE_NO_FUSED_EDGE", "NGPDE_NO_FUSED_EDGE_BWD", "NGPDE_DEEP_EDGE_BWD"])
def test_entry_follows_the_librarys_path_switches(switch, monkeypatch):
    # the fused forward with the primitives' pullback (saved pre-activations), no fused launch at all, the deep one-launch pullback
    monkeypatch.setenv(switch, "1")
    g = spatial(640, 9)
    phi = ng.Chain(ng.Dense(4, 32, "tanh"), ng.Dense(32, 32, "tanh"), ng.Dense(32, 16))
    gam = ng.Chain(ng.Dense(17, 24, "tanh"), ng.Dense(24, 1))
    layer = ng.VMHConv(phi, gam, initialgraph=g)
    assert_same(both_ways(layer, torch.randn(1, 640, device=DEV), 41, monkeypatch))


@pytest.mark.parametrize("variant,aggr", [("two-layer", "mean"), ("two-layer", "+"), ("two-layer", "max"), ("three-layer", "mean"), ("materialized", "mean"),
                                          ("single-dense", "+"), ("no-bias", "mean"), ("tanh-first", "mean"), ("edge-only", "mean")])
def test_gno_entry_equals_the_composed_layer(variant, aggr, monkeypatch):
    # GNOConv (src/layers.jl:509-547) through ngpde_gno_layer_*: the reassociated message with its per-edge input formed in the launch
    # (two-layer phi, relu / identity first), on the primitives (deeper phi, other first activations), the literal batched_mul
    cin, cout, k = 16, 32, 16
    g = spatial(600, 13, ndata_extra=1, edata=2)          # ndata = (x (2), f0 (1)) -> s has 3 rows; edata e (2)
    ds, de = 3, 2
    if variant == "edge-only":
        g = ng.GNNGraph(*g.edge_index(0), num_nodes=600, index_base=0, edata=dict(g.edata))
        ds = 0
    if variant == "materialized":
        monkeypatch.setenv("NGPDE_GNO_MATERIALIZE", "1")
    first = "tanh" if variant == "tanh-first" else "relu"
    if variant == "three-layer":
        phi = ng.Chain(ng.Dense(2 * ds + de, 24, first), ng.Dense(24, k, "swish"), ng.Dense(k, cin * cout))
    elif variant == "single-dense":
        phi = ng.Dense(2 * ds + de, cin * cout)
    else:
        phi = ng.Chain(ng.Dense(2 * ds + de, k, first), ng.Dense(k, cin * cout, bias=(variant != "no-bias")))
    layer = ng.GNOConv((cin, cout), phi, "swish" if variant in ("three-layer", "no-bias") else "relu", initialgraph=g, aggr=aggr,
                       bias=(variant != "no-bias"))
    x = torch.randn(cin, 600, device=DEV)
    assert_same(both_ways(layer, x, 51, monkeypatch))
    assert_same(both_ways(layer, x, 51, monkeypatch, training=False))


def dense_spatial(N, deg, seed, edata=0):
    """a random graph of `deg` incoming edges per node with 2-d positions and one more node feature: the density at which the layer
    entry takes the aggregate-then-transform form in training too (ngpde_gno_gform_preferred: >= 64 edges per node)"""
    rng = np.random.default_rng(seed)
    t = np.repeat(np.arange(N), deg)
    s = rng.integers(0, N, N * deg)
    perm = rng.permutation(N * deg)
    s, t = s[perm], t[perm]
    kw = dict(ndata={"x": rng.random((2, N)).astype(np.float32), "f0": rng.normal(size=(1, N)).astype(np.float32)})
    if edata:
        kw["edata"] = {"e": rng.normal(size=(edata, s.size)).astype(np.float32)}
    return ng.GNNGraph(s, t, num_nodes=N, index_base=0, **kw)


@pytest.mark.parametrize("cin,cout,aggr,edata,bias,first,deg", [(128, 128, "mean", 0, True, "relu", 70),    # BASELINE config 5's shape and density
                                                                 (32, 48, "+", 2, True, "identity", 64), (64, 20, "mean", 0, False, "relu", 65),
                                                                 (128, 64, "+", 1, True, "relu", 6)])        # sparse: by-source when training
def test_gno_gform_entry_equals_the_composed_layer(cin, cout, aggr, edata, bias, first, deg, monkeypatch):
    # the aggregate-then-transform form (csrc/gno_gform.hip: k = 64, in in {32, 64, 128}, sum / mean): the entry and the composed path
    # run the same launches; with NGPDE_NO_GNO_GFORM=1 both fall back to the by-source form
    monkeypatch.delenv("NGPDE_NO_GNO_GFORM", raising=False)      # (the suite may run under the switch: tools/switch_matrix.sh)
    k, N = 64, 300
    g = dense_spatial(N, deg, 17, edata=edata)
    ds = 3
    phi = ng.Chain(ng.Dense(2 * ds + edata, k, first), ng.Dense(k, cin * cout, bias=bias))
    layer = ng.GNOConv((cin, cout), phi, "relu", initialgraph=g, aggr=aggr, bias=bias)
    x = torch.randn(cin, N, device=DEV)
    act1 = 1 if first == "relu" else 0
    assert composed.gno_gform_preferred(N, N * deg, cin, cout, k, act1, aggr, False)
    assert composed.gno_gform_preferred(N, N * deg, cin, cout, k, act1, aggr, True) == (deg >= 64)
    a = both_ways(layer, x, 61, monkeypatch)
    assert_same(a)
    ai = both_ways(layer, x, 61, monkeypatch, training=False)
    assert_same(ai)
    monkeypatch.setenv("NGPDE_NO_GNO_GFORM", "1")
    assert not composed.gno_gform_preferred(N, N * deg, cin, cout, k, act1, aggr, False)
    b = both_ways(layer, x, 61, monkeypatch)
    assert_same(b)
    bi = both_ways(layer, x, 61, monkeypatch, training=False)
    # the two forms are two summation orders of the same layer: outputs and every gradient agree to rounding
    for u, v in list(zip(a[0], b[0])) + list(zip(ai[0], bi[0])):
        scale = float(v.abs().max()) + 1e-30
        assert float((u - v).abs().max()) <= 2e-5 * scale + 1e-6, (float((u - v).abs().max()), scale)


def test_gno_gform_aggregate_against_float64(monkeypatch):
    # G_i[k][i'] = sum_{e -> i} z_e[k] h_{s_e}[i'], hsum_i, z_out: rows of 0, 1, 33 and 70 edges, sum and mean, all three feature blocks
    monkeypatch.delenv("NGPDE_NO_GNO_GFORM", raising=False)
    lib = _lib.load()
    rng = np.random.default_rng(5)
    N, k = 90, 64
    s, t = [], []
    for node, deg in ((1, 1), (2, 33), (3, 70), (5, 4), (7, 32)):
        src = rng.integers(0, N, deg)
        s += list(src); t += [node] * deg
    perm = rng.permutation(len(s))
    s, t = np.asarray(s)[perm], np.asarray(t)[perm]
    g = ng.GNNGraph(s, t, num_nodes=N, index_base=0)
    handle = g.handle()
    E = len(s)
    order = np.argsort(t, kind="stable")                    # p order: by target, COO order inside a row
    for cin in (32, 64, 128):
        for mean in (0, 1):
            P, Q, Et = (rng.normal(size=(N, k)).astype(np.float32), rng.normal(size=(N, k)).astype(np.float32),
                        rng.normal(size=(E, k)).astype(np.float32))
            h = rng.normal(size=(N, cin)).astype(np.float32)
            dv = lambda a: torch.as_tensor(a, device=DEV)
            G = torch.full((N, k * cin), float("nan"), device=DEV)
            hs = torch.full((N, cin), float("nan"), device=DEV)
            zo = torch.full((E, k), float("nan"), device=DEV)
            tP, tQ, tE, th = dv(P), dv(Q), dv(Et), dv(h)
            _lib.check(lib.ngpde_gno_gform_aggregate(handle.ptr, cin, k, 1, mean, _lib.ptr(tP), _lib.ptr(tQ), _lib.ptr(tE), _lib.ptr(th), _lib.ptr(G),
                                                     _lib.ptr(hs), _lib.ptr(zo), _lib.current_stream()))
            z = np.maximum(P[t[order]].astype(np.float64) + Q[s[order]] + Et, 0.0)          # [E][k], p order
            Gr, hr = np.zeros((N, k, cin)), np.zeros((N, cin))
            for p_, e in enumerate(order):
                Gr[t[e]] += np.outer(z[p_], h[s[e]].astype(np.float64))
                hr[t[e]] += h[s[e]]
            if mean:
                deg = np.maximum(np.bincount(t, minlength=N), 1)[:, None]
                Gr, hr = Gr / deg[:, :, None], hr / deg
            assert np.abs(zo.cpu().numpy() - z).max() <= 1e-6
            assert np.abs(G.cpu().numpy().reshape(N, k, cin) - Gr).max() <= 1e-4 * np.abs(Gr).max() + 1e-6
            assert np.abs(hs.cpu().numpy() - hr).max() <= 1e-5 * np.abs(hr).max() + 1e-6
    assert lib.ngpde_gno_gform_supported(16, 64) == 0 and lib.ngpde_gno_gform_supported(128, 32) == 0
    assert lib.ngpde_gno_gform_aggregate(handle.ptr, 16, 64, 1, 0, None, None, None, None, None, None, None, None) == _lib.ERR_UNSUPPORTED


def test_second_backward_through_a_retained_graph(monkeypatch):
    # the layer entries keep their workspace (every saved activation) with the autograd node: a second pullback through a retained
    # graph reads the same saved state and gives the same bits as the first
    g = spatial(640, 9)
    phi = ng.Chain(ng.Dense(4, 32, "tanh"), ng.Dense(32, 16))
    gam = ng.Chain(ng.Dense(17, 24, "tanh"), ng.Dense(24, 1))
    vmh = ng.VMHConv(phi, gam, initialgraph=g)
    g2 = spatial(600, 13, ndata_extra=1)
    gno = ng.GNOConv((32, 32), ng.Chain(ng.Dense(6, 64, "relu"), ng.Dense(64, 32 * 32)), "relu", initialgraph=g2, aggr="mean")
    for layer, x in ((vmh, torch.randn(1, 640, device=DEV)), (gno, torch.randn(32, 600, device=DEV))):
        ps0, st = ng.setup(71, layer)
        ps = prep(ps0, 71)
        xs = x.clone().requires_grad_(True)
        y, _ = layer(xs, ps, st)
        leaves = [xs] + list(grad_leaves(ps))
        loss = (y * y).sum()
        first = torch.autograd.grad(loss, leaves, retain_graph=True)
        second = torch.autograd.grad(loss, leaves)
        for a, b in zip(first, second):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32))


def test_pullback_refuses_a_workspace_filled_under_another_plan(monkeypatch):
    # the plan (kernel sequence + workspace layout) is re-derived by every call; a path switch flipped between the forward and its pullback
    # used to give wrong gradients silently -- now the pullback compares the plan with the one the forward recorded for this workspace
    for var in ("NGPDE_NO_GNO_GFORM", "NGPDE_NO_FUSED_EDGE_BWD"):
        monkeypatch.delenv(var, raising=False)
    g = spatial(640, 9)
    phi = ng.Chain(ng.Dense(4, 32, "tanh"), ng.Dense(32, 16))
    gam = ng.Chain(ng.Dense(17, 24, "tanh"), ng.Dense(24, 1))
    layer = ng.VMHConv(phi, gam, initialgraph=g)
    ps0, st = ng.setup(81, layer)
    ps = prep(ps0, 81)
    x = torch.randn(1, 640, device=DEV, requires_grad=True)
    y, _ = layer(x, ps, st)
    monkeypatch.setenv("NGPDE_NO_FUSED_EDGE_BWD", "1")
    with pytest.raises(_lib.NgpdeError) as err:
        y.sum().backward()
    assert err.value.code == _lib.ERR_STATE and "another kernel sequence" in str(err.value)
    monkeypatch.delenv("NGPDE_NO_FUSED_EDGE_BWD")
    y2, _ = layer(x, ps, st)
    y2.sum().backward()                      # the same switches on both sides: fine
    g2 = dense_spatial(300, 70, 23)
    gno = ng.GNOConv((32, 32), ng.Chain(ng.Dense(6, 64, "relu"), ng.Dense(64, 32 * 32)), "relu", initialgraph=g2, aggr="mean")
    psg0, stg = ng.setup(82, gno)
    psg = prep(psg0, 82)
    xg = torch.randn(32, 300, device=DEV, requires_grad=True)
    yg, _ = gno(xg, psg, stg)                # (training, 70 edges per node: the aggregate-then-transform form)
    monkeypatch.setenv("NGPDE_NO_GNO_GFORM", "1")
    with pytest.raises(_lib.NgpdeError) as err:
        yg.sum().backward()
    assert err.value.code == _lib.ERR_STATE


def test_entry_rejects_what_the_reference_rejects():
    lib = _lib.load()
    g = spatial(64, 2)
    h = g.handle()
    d = _lib.EdgeLayer()
    assert lib.ngpde_edge_layer_workspace_bytes(None, C.byref(d), 1) == 0
    assert lib.ngpde_edge_layer_forward(h.ptr, None, 0, None, None, 0, None) == _lib.ERR_INVALID_ARGUMENT
    x = torch.zeros(64, 3, device=DEV)
    w = torch.zeros(8, 8, device=DEV)
    d.kind, d.aggr, d.n_state = _lib.LAYER_EDGECONV, 1, 1
    d.state[0], d.state_width[0] = x.data_ptr(), 3
    d.phi.n_layers, d.phi.dims[0], d.phi.dims[1], d.phi.weight[0] = 1, 7, 8, w.data_ptr()      # the message has 2 * 3 + 0 rows, not 7
    y = torch.zeros(64, 8, device=DEV)
    assert lib.ngpde_edge_layer_forward(h.ptr, C.byref(d), 0, y.data_ptr(), y.data_ptr(), 1 << 20, None) == _lib.ERR_DIMENSION_MISMATCH
    assert b"first layer expects 7 input features, the message has 6" in lib.ngpde_last_error()
    d.phi.dims[0] = 6
    assert lib.ngpde_edge_layer_forward(h.ptr, C.byref(d), 0, y.data_ptr(), y.data_ptr(), 16, None) == _lib.ERR_WORKSPACE
    d.kind = 7
    assert lib.ngpde_edge_layer_forward(h.ptr, C.byref(d), 0, y.data_ptr(), y.data_ptr(), 1 << 20, None) == _lib.ERR_INVALID_ARGUMENT
    # MPPDEConv on a batch whose graphs do not share one structure (src/layers.jl:359-361)
    gb = ng.GNNGraph(np.array([0, 1, 2, 3, 4]), np.array([1, 0, 3, 2, 2]), num_nodes=5, index_base=0, num_graphs=2,
                     gdata={"θ": np.zeros((1, 2), np.float32)})
    layer = ng.MPPDEConv(ng.Dense(2 * 4 + 1, 8), ng.Dense(4 + 8 + 1, 4), initialgraph=gb)
    ps, st = ng.setup(0, layer)
    with pytest.raises(_lib.DimensionMismatch):
        layer(torch.zeros(4, 5, device=DEV), ng.to_device(ps, DEV), st)
