"""GPU parity tests of the GCNConv path and the fixed-step neural graph ODE: HIP kernels (through
the C ABI) against the float64 numpy oracle on the same seeded inputs.

Tolerances: forward  max|y - y_oracle| <= 1e-4 * max|y_oracle| + 1e-5,
gradients 2e-4 relative (segmented sums of <= ~150 fp32 terms; fp32 MFMA is an exact fmaf chain).
"""
import os

import numpy as np
import pytest
import torch

import ngpde_amd as ng
from oracle import ngpde_oracle as O
from ngpde_amd import synth as S

pytestmark = pytest.mark.gpu

DEV = "cuda"


def close(a, ref, rtol=1e-4, atol=1e-5, what=""):
    a = a.detach().cpu().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    ref = np.asarray(ref, dtype=np.float64)
    assert a.shape == ref.shape, (what, a.shape, ref.shape)
    err = np.abs(a - ref).max() if ref.size else 0.0
    bound = rtol * (np.abs(ref).max() if ref.size else 0.0) + atol
    assert err <= bound, f"{what}: max err {err:.3e} > {bound:.3e}"


def make_graph(N, E, seed, symmetric=False):
    rng = np.random.default_rng(seed)
    s = rng.integers(0, N, E)
    t = rng.integers(0, N, E)
    if symmetric:
        s, t = np.concatenate([s, t]), np.concatenate([t, s])
    return s, t


def run_layer(N, E, din, dout, act, seed, add_self_loops=True, bias=True, weights=None, use_edge_weight=False,
              grads=True):
    rng = np.random.default_rng(seed)
    s, t = make_graph(N, E, seed)
    ew = None
    if weights == "arg" or use_edge_weight:
        ew = (rng.random(s.size) + 0.5).astype(np.float32)
    g = ng.GNNGraph(s, t, num_nodes=N, index_base=0, edge_weight=ew if use_edge_weight else None)
    og = O.Graph(s, t, num_nodes=N, index_base=0, edge_weight=ew if use_edge_weight else None)
    l = ng.GCNConv((din, dout), act, initialgraph=g, bias=bias, add_self_loops=add_self_loops,
                   use_edge_weight=use_edge_weight)
    ps, st = ng.setup(seed, l)
    ps = ng.to_device(ps, DEV)
    if bias:
        ps["bias"] = torch.as_tensor(rng.normal(size=(dout, 1)).astype(np.float32), device=DEV)
    x = torch.as_tensor(rng.normal(size=(din, N)).astype(np.float32), device=DEV)
    for p in ps.values():
        p.requires_grad_(grads)
    x.requires_grad_(grads)
    ew_arg = torch.as_tensor(ew) if weights == "arg" else None
    y, st2 = l(x, ps, st, ew_arg) if ew_arg is not None else l(x, ps, st)
    assert y.shape == (dout, N) and st2["graph"] is st["graph"]
    W = ps["weight"].detach().cpu().double().numpy()
    b = ps["bias"].detach().cpu().double().numpy() if bias else None
    X = x.detach().cpu().double().numpy()
    yo, cache = O.gcn_conv(X, W, b, og, act, add_self_loops, use_edge_weight,
                           edge_weight=ew.astype(np.float64) if weights == "arg" else None)
    close(y, yo, what=f"gcn fwd {din}->{dout} {act}")
    if grads:
        R = rng.normal(size=(dout, N))
        (y * torch.as_tensor(R, dtype=torch.float32, device=DEV)).sum().backward()
        go = O.gcn_conv_backward(cache, R)
        close(x.grad, go["x"], rtol=2e-4, what="dx")
        close(ps["weight"].grad, go["weight"], rtol=2e-4, atol=1e-4, what="dW")
        if bias:
            close(ps["bias"].grad, go["bias"], rtol=2e-4, atol=1e-4, what="db")


def test_reference_fixture_gcn_3_to_5():
    # /root/reference/test/runtests.jl:9-25
    g = ng.GNNGraph([1, 1, 2, 3], [2, 3, 1, 1])
    l = ng.GCNConv((3, 5), initialgraph=g)
    ps, st = ng.setup(0, l)
    assert st == {"graph": g}
    x = torch.randn(3, g.num_nodes, device=DEV)
    y, st = l(x, ng.to_device(ps, DEV), st)
    assert tuple(y.shape) == (5, g.num_nodes)
    assert st == {"graph": g}
    og = O.Graph([1, 1, 2, 3], [2, 3, 1, 1])
    yo, _ = O.gcn_conv(x.cpu().double().numpy(), ps["weight"].double().numpy(), ps["bias"].double().numpy(), og)
    close(y, yo)


@pytest.mark.parametrize("din,dout", [(3, 5), (6, 3), (20, 33), (48, 16), (16, 64)])
def test_gcn_generic_dims(din, dout):
    run_layer(61, 300, din, dout, "tanh", seed=din * 100 + dout)


@pytest.mark.parametrize("din,dout,N", [(6, 3, 600), (4, 3, 600), (3, 2, 1000), (4, 3, 1999), (5, 9, 1000)])
def test_gcn_narrow_layers_beyond_512_nodes(din, dout, N):
    # dout < din with bias above 512 nodes: the bias gradient takes the two-stage column sum, whose partial sums share the
    # workspace region of the weight pullback's slabs (once sized for the slabs alone: 1536 bytes written into 1024)
    run_layer(N, 4 * N, din, dout, "tanh", seed=din * 1000 + dout)


@pytest.mark.parametrize("din,dout,N", [(6, 3, 600), (4, 3, 600), (3, 2, 1000), (4, 3, 1999), (2, 1, 513)])
def test_gcn_backward_stays_inside_the_queried_workspace(din, dout, N):
    # the C ABI with a workspace of EXACTLY the queried size and a canary behind it
    import ctypes as C
    from ngpde_amd import _lib
    lib = _lib.load()
    s, t = make_graph(N, 4 * N, seed=3)
    g = ng.GNNGraph(s, t, num_nodes=N, index_base=0)
    h = g.handle((True, None, False))
    x, wt = torch.randn(N, din, device=DEV), torch.randn(din, dout, device=DEV)
    z, dy = torch.randn(N, dout, device=DEV), torch.randn(N, dout, device=DEV)
    dx, dwt, db = torch.empty_like(x), torch.empty_like(wt), torch.empty(dout, device=DEV)
    need = int(lib.ngpde_gcn_workspace_bytes(h.ptr, din, dout, 1))
    guard = 1 << 16
    buf = torch.full((need + guard,), 0x5A, dtype=torch.uint8, device=DEV)
    _lib.check(lib.ngpde_gcn_backward(h.ptr, din, dout, 1, _lib.ptr(x), _lib.ptr(wt), _lib.ptr(z), None, _lib.ptr(dy), _lib.ptr(dx),
                                      _lib.ptr(dwt), _lib.ptr(db), _lib.ptr(buf), C.c_size_t(need), _lib.current_stream()))
    torch.cuda.synchronize()
    assert bool((buf[need:] == 0x5A).all()), "ngpde_gcn_backward wrote behind its workspace"
    assert torch.isfinite(db).all() and torch.isfinite(dwt).all()


@pytest.mark.parametrize("d", [16, 32, 64, 128])
@pytest.mark.parametrize("act", ["relu", "swish"])
def test_gcn_fused_dims(d, act):
    run_layer(1000 + d, 9000, d, d, act, seed=d)   # N not a multiple of the 32-row tile


def test_gcn_no_self_loops_no_bias_identity():
    # every node needs an incoming edge (an isolated node gives NaN in the reference too: 0 * Inf)
    N = 200
    s = np.concatenate([np.arange(N), np.random.default_rng(0).integers(0, N, 600)])
    t = np.concatenate([(np.arange(N) + 1) % N, np.random.default_rng(1).integers(0, N, 600)])
    g = ng.GNNGraph(s, t, num_nodes=N, index_base=0)
    og = O.Graph(s, t, num_nodes=N, index_base=0)
    for d in (64, 7):
        l = ng.GCNConv((d, d), initialgraph=g, bias=False, add_self_loops=False)
        ps, st = ng.setup(1, l)
        assert list(ps) == ["weight"]
        x = torch.randn(d, N, device=DEV)
        y, _ = l(x, ng.to_device(ps, DEV), st)
        yo, _ = O.gcn_conv(x.cpu().double().numpy(), ps["weight"].double().numpy(), None, og, "identity", False)
        close(y, yo)


def test_gcn_isolated_node_without_self_loops_is_nan_like_reference():
    g = ng.GNNGraph([1, 2], [2, 1], num_nodes=3)
    og = O.Graph([1, 2], [2, 1], num_nodes=3)
    l = ng.GCNConv((16, 16), initialgraph=g, add_self_loops=False)
    ps, st = ng.setup(1, l)
    x = torch.randn(16, 3, device=DEV)
    y, _ = l(x, ng.to_device(ps, DEV), st)
    with np.errstate(invalid="ignore"):
        yo, _ = O.gcn_conv(x.cpu().double().numpy(), ps["weight"].double().numpy(), ps["bias"].double().numpy(), og,
                           "identity", False)
    assert np.isnan(yo[:, 2]).all() and torch.isnan(y[:, 2]).all()
    close(y[:, :2], yo[:, :2])


@pytest.mark.parametrize("d", [64, 10])
def test_gcn_edge_weight_argument(d):
    run_layer(300, 2500, d, d, "relu", seed=5, weights="arg")


@pytest.mark.parametrize("din,dout,loops", [(64, 64, True), (10, 10, True), (12, 5, True), (5, 12, False), (32, 32, False)])
def test_gcn_gradient_wrt_the_edge_weight_argument(din, dout, loops):
    # (l::GCNConv)(x, ps, st, edge_weight): the reference differentiates through e_mul_xj (:228) and the weighted degree (:224);
    # fused widths (64, 32), the any-width path in both orders of W, with and without self loops; 2e-4 against the float64 oracle
    N, E = 700, 5000
    rng = np.random.default_rng(din * 100 + dout)
    s, t = make_graph(N, E, seed=77)
    if not loops:       # every node needs an incoming edge without self loops
        s, t = np.concatenate([s, np.arange(N)]), np.concatenate([t, (np.arange(N) + 1) % N])
    ew0 = (rng.random(s.size) + 0.5)
    g = ng.GNNGraph(s, t, num_nodes=N, index_base=0)
    og = O.Graph(s, t, num_nodes=N, index_base=0)
    l = ng.GCNConv((din, dout), "tanh", initialgraph=g, add_self_loops=loops)
    ps, st = ng.setup(1, l)
    ps = ng.to_device(ps, DEV)
    ps["bias"] = torch.as_tensor(rng.normal(size=(dout, 1)).astype(np.float32), device=DEV)
    for p in ps.values():
        p.requires_grad_(True)
    x = torch.as_tensor(rng.normal(size=(din, N)).astype(np.float32), device=DEV).requires_grad_(True)
    ew = torch.as_tensor(ew0.astype(np.float32), device=DEV).requires_grad_(True)
    y, _ = l(x, ps, st, ew)
    R = rng.normal(size=(dout, N))
    (y * torch.as_tensor(R, dtype=torch.float32, device=DEV)).sum().backward()
    W, b = ps["weight"].detach().cpu().double().numpy(), ps["bias"].detach().cpu().double().numpy()
    yo, cache = O.gcn_conv(x.detach().cpu().double().numpy(), W, b, og, "tanh", loops, edge_weight=ew0.astype(np.float32).astype(np.float64))
    go = O.gcn_conv_backward(cache, R)
    close(y, yo, what="y")
    close(x.grad, go["x"], rtol=2e-4, what="dx")
    close(ps["weight"].grad, go["weight"], rtol=2e-4, atol=1e-4, what="dW")
    close(ew.grad, go["edge_weight"], rtol=2e-4, atol=1e-5, what="d edge_weight")
    # without a gradient request the plain backward runs and x / W gradients are the same bits
    x2 = x.detach().clone().requires_grad_(True)
    y2, _ = l(x2, ps, st, ew.detach())
    (y2 * torch.as_tensor(R, dtype=torch.float32, device=DEV)).sum().backward()
    assert torch.equal(x2.grad, x.grad)


@pytest.mark.parametrize("d", [32, 9])
def test_gcn_use_edge_weight_unweighted_degree_quirk(d):
    run_layer(300, 2500, d, d, "relu", seed=6, use_edge_weight=True)


def test_gcn_wrong_number_of_edge_weights():
    g = ng.GNNGraph([1, 1, 2, 3], [2, 3, 1, 1])
    l = ng.GCNConv((16, 16), initialgraph=g)
    ps, st = ng.setup(0, l)
    with pytest.raises(ng.ArgumentError, match="Wrong number of edge weights \\(expected 4 but given 3\\)"):
        l(torch.randn(16, 3, device=DEV), ng.to_device(ps, DEV), st, torch.ones(3))


def test_gcn_empty_and_edgeless_graphs():
    g = ng.GNNGraph([], [], num_nodes=5)
    l = ng.GCNConv((16, 16), "relu", initialgraph=g)
    ps, st = ng.setup(0, l)
    x = torch.randn(16, 5, device=DEV)
    y, _ = l(x, ng.to_device(ps, DEV), st)
    yo, _ = O.gcn_conv(x.cpu().double().numpy(), ps["weight"].double().numpy(), ps["bias"].double().numpy(),
                       O.Graph([], [], num_nodes=5), "relu")
    close(y, yo)
    # default EMPTYGRAPH state (src/layers.jl:14,21): zero nodes in, zero nodes out
    l0 = ng.GCNConv((16, 16))
    ps0, st0 = ng.setup(0, l0)
    y0, _ = l0(torch.zeros(16, 0, device=DEV), ng.to_device(ps0, DEV), st0)
    assert tuple(y0.shape) == (16, 0)


def test_gcn_high_degree_rows():
    # a hub with 700 incoming edges (> one chunk of 16 entries many times over) and Cora-like skew
    N = 800
    s = np.concatenate([np.arange(1, 701), np.random.default_rng(0).integers(0, N, 3000)])
    t = np.concatenate([np.zeros(700, np.int64), np.random.default_rng(1).integers(0, N, 3000)])
    g = ng.GNNGraph(s, t, num_nodes=N, index_base=0)
    og = O.Graph(s, t, num_nodes=N, index_base=0)
    for d in (64, 32, 128):
        l = ng.GCNConv((d, d), "relu", initialgraph=g)
        ps, st = ng.setup(d, l)
        x = torch.randn(d, N, device=DEV)
        y, _ = l(x, ng.to_device(ps, DEV), st)
        yo, _ = O.gcn_conv(x.cpu().double().numpy(), ps["weight"].double().numpy(), ps["bias"].double().numpy(), og, "relu")
        close(y, yo)


def test_gcn_c2_full_size_parity_and_determinism():
    # BASELINE config 2 graph: 16384 nodes, 65536 closest pairs -> 131072 directed edges, 64-d
    pts, s, t = S.closest_pairs_graph(16384, 65536, seed=2)
    g = ng.GNNGraph(s, t, num_nodes=16384, index_base=0)
    og = O.Graph(s, t, num_nodes=16384, index_base=0)
    l = ng.GCNConv((64, 64), "relu", initialgraph=g)
    ps, st = ng.setup(2, l)
    ps = ng.to_device(ps, DEV)
    x = torch.as_tensor(S.normal(22, 64 * 16384).reshape(64, 16384).astype(np.float32), device=DEV)
    y, _ = l(x, ps, st)
    yo, _ = O.gcn_conv(x.cpu().double().numpy(), ps["weight"].cpu().double().numpy(),
                       ps["bias"].cpu().double().numpy(), og, "relu")
    close(y, yo)
    y2, _ = l(x, ps, st)
    assert torch.equal(y, y2), "segmented reduction must be bitwise reproducible"


def test_chain_of_gcn_and_updategraph_on_gpu():
    # test/runtests.jl:181-185: one updategraph call re-targets every layer of a Chain
    g = ng.rand_graph(50, 200, seed=0)
    g2 = ng.rand_graph(50, 300, seed=1)
    model = ng.Chain(ng.GCNConv((16, 16), "relu", initialgraph=g), ng.GCNConv((16, 16), initialgraph=g))
    ps, st = ng.setup(0, model)
    st = ng.updategraph(st, g2)
    assert st["layer_1"]["graph"] is g2 and st["layer_2"]["graph"] is g2
    ps = ng.to_device(ps, DEV)
    x = torch.randn(16, 50, device=DEV)
    y, _ = model(x, ps, st)
    s, t = g2.edge_index(0)
    og = O.Graph(s, t, num_nodes=50, index_base=0)
    p = lambda k, n: ps[k][n].cpu().double().numpy()
    h, _ = O.gcn_conv(x.cpu().double().numpy(), p("layer_1", "weight"), p("layer_1", "bias"), og, "relu")
    yo, _ = O.gcn_conv(h, p("layer_2", "weight"), p("layer_2", "bias"), og)
    close(y, yo)


# ---- fixed-step neural graph ODE -----------------------------------------------------------------

def node_case(N, E, d, tab, nsteps, dt, act, seed):
    rng = np.random.default_rng(seed)
    s, t = make_graph(N, E, seed, symmetric=False)
    g = ng.GNNGraph(s, t, num_nodes=N, index_base=0)
    og = O.Graph(s, t, num_nodes=N, index_base=0)
    params = [dict(weight=S.glorot_uniform(seed + 10 + k, d, d), bias=rng.normal(size=(d, 1)) * 0.1) for k in range(2)]
    u0 = rng.normal(size=(d, N))
    return g, og, params, u0


@pytest.mark.parametrize("tab,d,act", [("euler", 32, "relu"), ("tsit5", 64, "relu"), ("tsit5", 16, "tanh"),
                                        ("tsit5", 128, "swish")])
def test_node_forward_backward_parity(tab, d, act):
    N, E, nsteps, dt = 333, 2400, 3, 0.1
    g, og, params, u0 = node_case(N, E, d, tab, nsteps, dt, act, seed=d)
    rhs = ng.Chain(ng.GCNConv((d, d), act, initialgraph=g), ng.GCNConv((d, d), act, initialgraph=g))
    node = ng.NeuralODE(rhs, solver=tab, n_steps=nsteps, dt=dt)
    ps, st = ng.setup(0, node)
    for k, name in enumerate(["layer_1", "layer_2"]):
        ps[name]["weight"] = torch.as_tensor(params[k]["weight"].astype(np.float32))
        ps[name]["bias"] = torch.as_tensor(params[k]["bias"].astype(np.float32))
    ps = ng.to_device(ps, DEV)
    for lp in ps.values():
        for v in lp.values():
            v.requires_grad_(True)
    u = torch.as_tensor(u0.astype(np.float32), device=DEV).requires_grad_(True)
    uT, _ = node(u, ps, st)
    uTo, du0o, acc = O.gcn2_node_loss_and_grads(params, og, u0, O.TABLEAUS[tab], dt, nsteps, act)
    close(uT, uTo, rtol=2e-4, what="u(T)")
    uT.sum().backward()
    close(u.grad, du0o, rtol=5e-4, atol=1e-4, what="du0")
    for k, name in enumerate(["layer_1", "layer_2"]):
        close(ps[name]["weight"].grad, acc[k]["weight"], rtol=5e-4, atol=1e-3, what=f"dW{k + 1}")
        close(ps[name]["bias"].grad, acc[k]["bias"], rtol=5e-4, atol=1e-3, what=f"db{k + 1}")


@pytest.mark.parametrize("d,act,graph", [(16, "tanh", "knn"), (32, "swish", "radius"), (64, "tanh", "knn"), (64, "relu", "radius"),
                                         (32, "relu", "knn")])
def test_node_prescaled_pipeline_parity(d, act, graph, monkeypatch):
    """The solver plan on graphs whose tiles fit the LDS halo holds its arrays as c .* x and stages halo rows by LDS-DMA
    (ngpde_node_flags: prescaled).  Same parity bar as the register-staged form, on a symmetric radius graph and on a
    directed k-NN graph (in- and out-lists differ: forward and pullback walk different halos), and bit-for-bit agreement
    is NOT required between the two forms -- they round differently -- only agreement within the tolerance."""
    N, nsteps, dt = 2048, 3, 0.1
    rng = np.random.default_rng(d)
    P = rng.random((2, N)).astype(np.float32)
    g = ng.radius_graph(P, 0.035) if graph == "radius" else ng.knn_graph(P, 6)
    s, t = g.edge_index(0)
    og = O.Graph(s, t, num_nodes=N, index_base=0)
    params = [dict(weight=S.glorot_uniform(d + 10 + k, d, d), bias=rng.normal(size=(d, 1)) * 0.1) for k in range(2)]
    u0 = rng.normal(size=(d, N))
    uTo, du0o, acc = O.gcn2_node_loss_and_grads(params, og, u0, O.TABLEAUS["tsit5"], dt, nsteps, act)
    results = {}
    switched_off = bool(os.environ.get("NGPDE_NO_PRESCALE") or os.environ.get("NGPDE_NO_HALO"))   # suite run under a switch
    for form in ("prescaled", "plain"):
        if form == "plain":
            monkeypatch.setenv("NGPDE_NO_PRESCALE", "1")
        rhs = ng.Chain(ng.GCNConv((d, d), act, initialgraph=g), ng.GCNConv((d, d), act, initialgraph=g))
        node = ng.NeuralODE(rhs, solver="tsit5", n_steps=nsteps, dt=dt)
        ps, st = ng.setup(0, node)
        for k, name in enumerate(["layer_1", "layer_2"]):
            ps[name]["weight"] = torch.as_tensor(params[k]["weight"].astype(np.float32))
            ps[name]["bias"] = torch.as_tensor(params[k]["bias"].astype(np.float32))
        ps = ng.to_device(ps, DEV)
        for lp in ps.values():
            for v in lp.values():
                v.requires_grad_(True)
        u = torch.as_tensor(u0.astype(np.float32), device=DEV).requires_grad_(True)
        uT, _ = node(u, ps, st)
        plan = node.plan_for(ps, st, True)
        assert ("prescaled" in plan.flags()) == (form == "prescaled" and not switched_off)
        close(uT, uTo, rtol=2e-4, what=f"{form} u(T)")
        uT.sum().backward()
        # relu: a pre-activation within an ulp of zero may fall on either side of the kink (see test_configs_gpu.py)
        loose = 20.0 if act == "relu" else 1.0
        close(u.grad, du0o, rtol=5e-4 * loose, atol=1e-4 * loose, what=f"{form} du0")
        for k, name in enumerate(["layer_1", "layer_2"]):
            close(ps[name]["weight"].grad, acc[k]["weight"], rtol=5e-4 * loose, atol=1e-3, what=f"{form} dW{k + 1}")
            close(ps[name]["bias"].grad, acc[k]["bias"], rtol=5e-4 * loose, atol=1e-3, what=f"{form} db{k + 1}")
        results[form] = uT.detach().cpu().numpy()
    assert np.allclose(results["prescaled"], results["plain"], rtol=1e-4, atol=1e-5)


def test_node_matches_layerwise_euler_step():
    # one Euler step through the plan == u + dt * Chain(GCNConv, GCNConv)(u) through the layer API
    N, d = 200, 32
    g = ng.rand_graph(N, 1200, seed=3)
    rhs = ng.Chain(ng.GCNConv((d, d), "relu", initialgraph=g), ng.GCNConv((d, d), "relu", initialgraph=g))
    node = ng.NeuralODE(rhs, solver="euler", n_steps=1, dt=0.25)
    ps, st = ng.setup(4, node)
    ps = ng.to_device(ps, DEV)
    u = torch.randn(d, N, device=DEV)
    uT, _ = node(u, ps, st)
    k, _ = rhs(u, ps, st)
    assert torch.allclose(uT, u + 0.25 * k, rtol=1e-5, atol=1e-6)


def test_node_replay_is_deterministic_and_reusable():
    N, d = 500, 64
    g = ng.rand_graph(N, 4000, seed=5)
    rhs = ng.Chain(ng.GCNConv((d, d), "relu", initialgraph=g), ng.GCNConv((d, d), "relu", initialgraph=g))
    node = ng.NeuralODE(rhs, solver="tsit5", n_steps=4, dt=0.05)
    ps, st = ng.setup(4, node)
    ps = ng.to_device(ps, DEV)
    u = torch.randn(d, N, device=DEV)
    a, _ = node(u, ps, st)
    b, _ = node(u, ps, st)
    assert torch.equal(a, b)
    c, _ = node(2 * u, ps, st)
    assert not torch.equal(a, c)


def test_node_two_forwards_then_one_backward_matches_the_oracle():
    # y1 = node(u1); y2 = node(u2); (y1.sum() + y2.sum()).backward(): each solve keeps its own tape (one plan per outstanding
    # solve), so du1 / du2 are each solve's own adjoint and the parameter gradients are the sum of both
    N, E, d, nsteps, dt = 257, 1900, 32, 3, 0.1
    g, og, params, u0 = node_case(N, E, d, "tsit5", nsteps, dt, "relu", seed=77)
    u0b = np.random.default_rng(78).normal(size=(d, N))
    rhs = ng.Chain(ng.GCNConv((d, d), "relu", initialgraph=g), ng.GCNConv((d, d), "relu", initialgraph=g))
    node = ng.NeuralODE(rhs, solver="tsit5", n_steps=nsteps, dt=dt)
    ps, st = ng.setup(0, node)
    for k, name in enumerate(["layer_1", "layer_2"]):
        ps[name]["weight"] = torch.as_tensor(params[k]["weight"].astype(np.float32))
        ps[name]["bias"] = torch.as_tensor(params[k]["bias"].astype(np.float32))
    ps = ng.to_device(ps, DEV)
    for lp in ps.values():
        for v in lp.values():
            v.requires_grad_(True)
    u1 = torch.as_tensor(u0.astype(np.float32), device=DEV).requires_grad_(True)
    u2 = torch.as_tensor(u0b.astype(np.float32), device=DEV).requires_grad_(True)
    y1, _ = node(u1, ps, st)
    y2, _ = node(u2, ps, st)
    assert sum(len(pool) for pool in node._plans.values()) == 2          # the second solve did not take the first one's tape
    (y1.sum() + y2.sum()).backward()
    o1 = O.gcn2_node_loss_and_grads(params, og, u0, O.TABLEAUS["tsit5"], dt, nsteps, "relu")
    o2 = O.gcn2_node_loss_and_grads(params, og, u0b, O.TABLEAUS["tsit5"], dt, nsteps, "relu")
    close(y1, o1[0], rtol=2e-4, what="u1(T)")
    close(y2, o2[0], rtol=2e-4, what="u2(T)")
    close(u1.grad, o1[1], rtol=5e-4, atol=1e-4, what="du1")
    close(u2.grad, o2[1], rtol=5e-4, atol=1e-4, what="du2")
    for k, name in enumerate(["layer_1", "layer_2"]):
        close(ps[name]["weight"].grad, o1[2][k]["weight"] + o2[2][k]["weight"], rtol=5e-4, atol=1e-3, what=f"dW{k + 1}")
        close(ps[name]["bias"].grad, o1[2][k]["bias"] + o2[2][k]["bias"], rtol=5e-4, atol=1e-3, what=f"db{k + 1}")
    # both plans are free again: the next solve re-uses one of them
    y3, _ = node(u1, ps, st)
    assert sum(len(pool) for pool in node._plans.values()) == 2
    del y3


def test_node_plan_rejects_a_backward_whose_tape_was_overwritten():
    # straight through the C ABI: forward(u1), forward(u2), then the backward of the FIRST solve -> NGPDE_ERR_STATE
    import ctypes as C
    from ngpde_amd import _lib
    from ngpde_amd.node import _Plan
    N, d = 128, 16
    g = ng.rand_graph(N, 800, seed=9)
    plan = _Plan(g.handle((True, None, False)), d, _lib.ACT["relu"], "euler", 2, 0.1, True)
    lib, p = _lib.load(), _lib.ptr
    u1, u2 = torch.randn(N, d, device=DEV), torch.randn(N, d, device=DEV)
    w, b, out = torch.randn(d, d, device=DEV) * 0.1, torch.zeros(d, device=DEV), torch.empty(N, d, device=DEV)
    stream = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.ngpde_node_gcn2_forward(plan.ptr, p(u1), p(w), p(b), p(w), p(b), p(out), stream))
    gen1, pending = plan.generation()
    assert pending and gen1 == 1
    _lib.check(lib.ngpde_node_expect_generation(plan.ptr, gen1))
    _lib.check(lib.ngpde_node_gcn2_forward(plan.ptr, p(u2), p(w), p(b), p(w), p(b), p(out), stream))
    with pytest.raises(ng.NgpdeError, match="another forward ran on this plan"):
        _lib.check(lib.ngpde_node_expect_generation(plan.ptr, gen1))
    gen2, _ = plan.generation()
    _lib.check(lib.ngpde_node_expect_generation(plan.ptr, gen2))
    _lib.check(lib.ngpde_node_gcn2_backward(plan.ptr, p(torch.ones(N, d, device=DEV)), p(out), None, None, None, None, stream))
    assert plan.generation() == (gen2, False)
    torch.cuda.synchronize()


def test_gcn_edge_weight_argument_reuses_one_handle_and_stays_bounded():
    # GCNConv(x, ps, st, edge_weight) as an ODE right-hand side passes the same weights on every call: one derived-graph
    # handle, no download of the weights, and a bounded cache when the weights do change
    N, E, d = 300, 2500, 16
    s, t = make_graph(N, E, 3)
    g = ng.GNNGraph(s, t, num_nodes=N, index_base=0)
    l = ng.GCNConv((d, d), "relu", initialgraph=g)
    ps, st = ng.setup(0, l)
    ps = ng.to_device(ps, DEV)
    x = torch.randn(d, N, device=DEV)
    ew = torch.rand(E, device=DEV) + 0.5
    y0, _ = l(x, ps, st, ew)
    lru = st["graph"]._shared["weighted_handles"]
    assert len(lru) == 1
    h0 = next(iter(lru.values()))[0]
    for _ in range(20):
        y, _ = l(x, ps, st, ew)
    assert len(lru) == 1 and next(iter(lru.values()))[0] is h0 and torch.equal(y, y0)
    ew.mul_(2.0)                                   # in-place change: new content version -> a new normalisation
    y2, _ = l(x, ps, st, ew)
    assert len(lru) == 2 and not torch.equal(y2, y0)
    og = O.Graph(s, t, num_nodes=N, index_base=0)
    yo, _ = O.gcn_conv(x.cpu().double().numpy(), ps["weight"].cpu().double().numpy(), ps["bias"].cpu().double().numpy(), og, "relu",
                       True, False, edge_weight=ew.cpu().double().numpy())
    close(y2, yo, what="weights changed in place")
    for k in range(10):                            # fresh weight tensors every call: the cache stays at its cap
        l(x, ps, st, torch.rand(E, device=DEV) + 0.5)
    assert len(lru) <= ng.GNNGraph.max_weighted_handles


def spatial_case(N, pairs, d, seed):
    """closest-pairs graph on N uniform points (the C2 construction at test size; N need not be a multiple of the tile height)"""
    _, s, t = S.closest_pairs_graph(N, pairs, seed=seed)
    rng = np.random.default_rng(seed)
    params = [dict(weight=S.glorot_uniform(seed + 10 + k, d, d), bias=rng.normal(size=(d, 1)) * 0.1) for k in range(2)]
    return ng.GNNGraph(s, t, num_nodes=N, index_base=0), O.Graph(s, t, num_nodes=N, index_base=0), params


def _oracle_node_with_seed(params, og, u0, seed, tableau, dt, nsteps, act="relu"):
    """solve + discrete adjoint of loss = sum(seed .* u(T)) with the oracle's own pieces (gcn2_rhs, rk_solve, rk_adjoint)"""
    rhs, vjp = O.gcn2_rhs(params, og, act)
    uT, tape = O.rk_solve(rhs, u0, tableau, dt, nsteps)
    acc = [dict(weight=np.zeros_like(p["weight"]), bias=np.zeros_like(p["bias"])) for p in params]

    def accumulate(pg):
        for A, G in zip(acc, pg):
            A["weight"] += G["weight"]
            A["bias"] += G["bias"].reshape(A["bias"].shape)
    du0 = O.rk_adjoint(vjp, tape, seed.copy(), tableau, dt, accumulate)
    return uT, du0, acc


PLAN_SWITCHES = ("NGPDE_NO_PERSISTENT", "NGPDE_PERSISTENT", "NGPDE_NO_WIDEN", "NGPDE_NO_TILE_PAIRS", "NGPDE_TILE_ROUNDS", "NGPDE_NO_INTERLEAVE",
                 "NGPDE_WEIGHTED_TILE_ROUNDS", "NGPDE_NO_TILE_PIPE", "NGPDE_NO_PRESCALE", "NGPDE_NO_MASK", "NGPDE_NO_OWN_FIRST", "NGPDE_OWN_FIRST_ADJOINT")


def needs_persistent_plan(monkeypatch=None):
    """The persistent solver plan is what these tests are about: they lift a suite-wide NGPDE_NO_PERSISTENT (read at every plan
    creation) and skip under NGPDE_NO_HALO, which the library reads once and which leaves no plan it could run on."""
    import os
    if os.environ.get("NGPDE_NO_HALO") == "1":
        pytest.skip("NGPDE_NO_HALO=1: no LDS-staged tiles, so no persistent plan")
    if monkeypatch is not None:      # (the suite may run under any of the plan-selecting switches: tools/switch_matrix.sh)
        for var in PLAN_SWITCHES:
            monkeypatch.delenv(var, raising=False)


@pytest.mark.parametrize("tab,persistent", [("tsit5", True), ("euler", True), ("tsit5", False)])
def test_node_batch_of_identical_graphs_member_by_member(tab, persistent, monkeypatch):
    # batch([g, g, g]) (test/runtests.jl:89-102; "all graphs need to have the same structure", src/layers.jl:359-361): the
    # persistent plan solves the members one after the other on the member's handle.  Every member against the float64 oracle of
    # that member alone, parameter gradients against the sum over the members; the same through the one-big-handle path.
    if persistent:
        needs_persistent_plan(monkeypatch)
    else:
        monkeypatch.setenv("NGPDE_NO_PERSISTENT", "1")
    N, d, K, nsteps, dt = 1000, 64, 3, 2, 0.1
    g, og, params = spatial_case(N, 4 * N, d, seed=91)      # a radius-style graph: its tiles fit the LDS halo (persistent plan)
    gb = ng.batch([g, g.copy(), g])
    rhs = ng.Chain(ng.GCNConv((d, d), "relu", initialgraph=gb), ng.GCNConv((d, d), "relu", initialgraph=gb))
    node = ng.NeuralODE(rhs, solver=tab, n_steps=nsteps, dt=dt)
    ps, st = ng.setup(0, node)
    for k, name in enumerate(["layer_1", "layer_2"]):
        ps[name]["weight"] = torch.as_tensor(params[k]["weight"].astype(np.float32))
        ps[name]["bias"] = torch.as_tensor(params[k]["bias"].astype(np.float32))
    ps = ng.to_device(ps, DEV)
    for lp in ps.values():
        for v in lp.values():
            v.requires_grad_(True)
    rng = np.random.default_rng(92)
    u0 = rng.normal(size=(d, K * N))
    R = rng.normal(size=(d, K * N))
    u = torch.as_tensor(u0.astype(np.float32), device=DEV).requires_grad_(True)
    uT, _ = node(u, ps, st)
    plan = next(iter(node._plans.values()))[0]
    assert plan.members == (K if persistent else 1)
    assert ("persistent_fwd" in plan.flags()) == persistent
    (uT * torch.as_tensor(R.astype(np.float32), device=DEV)).sum().backward()
    assert not plan.fault()
    accW = [np.zeros((d, d)), np.zeros((d, d))]
    accb = [np.zeros((d, 1)), np.zeros((d, 1))]
    for m in range(K):
        sl = slice(m * N, (m + 1) * N)
        uTo, du0o, acc = _oracle_node_with_seed(params, og, u0[:, sl], R[:, sl], O.TABLEAUS[tab], dt, nsteps)
        close(uT[:, sl], uTo, rtol=2e-4, what=f"u(T) member {m}")
        close(u.grad[:, sl], du0o, rtol=5e-4, atol=1e-4, what=f"du0 member {m}")
        for k in range(2):
            accW[k] += acc[k]["weight"]
            accb[k] += acc[k]["bias"]
    for k, name in enumerate(["layer_1", "layer_2"]):
        close(ps[name]["weight"].grad, accW[k], rtol=5e-4, atol=1e-3, what=f"dW{k + 1}")
        close(ps[name]["bias"].grad, accb[k], rtol=5e-4, atol=1e-3, what=f"db{k + 1}")


@pytest.mark.parametrize("tab,N,nsteps,act", [("tsit5", 1000, 3, "relu"), ("euler", 2048, 5, "relu"), ("tsit5", 77, 2, "relu"),
                                              ("tsit5", 1000, 3, "tanh"), ("euler", 2048, 4, "swish"), ("tsit5", 16384, 2, "tanh"),
                                              ("tsit5", 500, 2, "leakyrelu"), ("tsit5", 500, 2, "identity")])
def test_node_persistent_plan_against_oracle(tab, N, nsteps, act, monkeypatch):
    needs_persistent_plan(monkeypatch)
    # the persistent plan (one launch per direction, tiles synchronised by phase flags) on graphs whose tiles fit the LDS halo:
    # u(T), du0 and the parameter gradients against the float64 oracle; a last tile with padding rows (N = 1000, 77); activations
    # other than relu (the tutorial's chain with tanh, graph_node.md:78 takes any NNlib activation): the adjoint launch forms
    # act'(z) from the pre-activations the forward launch keeps instead of the sign bits
    d, dt = 64, 0.1
    g, og, params = spatial_case(N, 4 * N, d, seed=N)
    rhs = ng.Chain(ng.GCNConv((d, d), act, initialgraph=g), ng.GCNConv((d, d), act, initialgraph=g))
    node = ng.NeuralODE(rhs, solver=tab, n_steps=nsteps, dt=dt)
    ps, st = ng.setup(0, node)
    for k, name in enumerate(["layer_1", "layer_2"]):
        ps[name]["weight"] = torch.as_tensor(params[k]["weight"].astype(np.float32))
        ps[name]["bias"] = torch.as_tensor(params[k]["bias"].astype(np.float32))
    ps = ng.to_device(ps, DEV)
    for lp in ps.values():
        for v in lp.values():
            v.requires_grad_(True)
    rng = np.random.default_rng(N + 1)
    u0, R = rng.normal(size=(d, N)), rng.normal(size=(d, N))
    u = torch.as_tensor(u0.astype(np.float32), device=DEV).requires_grad_(True)
    uT, _ = node(u, ps, st)
    plan = next(iter(node._plans.values()))[0]
    assert {"persistent_fwd", "persistent_bwd"} <= plan.flags()
    assert plan.launch_count() == (2, 3)      # flag reset + solve; flag reset + adjoint + one reduction (the fault latches ride on the exit kernels)
    (uT * torch.as_tensor(R.astype(np.float32), device=DEV)).sum().backward()
    assert not plan.fault()
    uTo, du0o, acc = _oracle_node_with_seed(params, og, u0, R, O.TABLEAUS[tab], dt, nsteps, act)
    close(uT, uTo, rtol=2e-4, what="u(T)")
    close(u.grad, du0o, rtol=5e-4, atol=1e-4, what="du0")
    for k, name in enumerate(["layer_1", "layer_2"]):
        close(ps[name]["weight"].grad, acc[k]["weight"], rtol=5e-4, atol=1e-3, what=f"dW{k + 1}")
        close(ps[name]["bias"].grad, acc[k]["bias"], rtol=5e-4, atol=1e-3, what=f"db{k + 1}")


@pytest.mark.parametrize("d,tab,N,nsteps,act,members", [(32, "tsit5", 1000, 3, "relu", 1), (16, "tsit5", 2048, 2, "tanh", 1),
                                                      (32, "euler", 16384, 3, "relu", 1), (16, "tsit5", 33000, 2, "relu", 1),
                                                      (32, "tsit5", 40000, 2, "tanh", 1), (16, "tsit5", 700, 2, "sigmoid", 1),
                                                      (32, "tsit5", 900, 2, "relu", 3)])
def test_node_persistent_plan_narrow_widths_against_oracle(d, tab, N, nsteps, act, members, monkeypatch):
    needs_persistent_plan(monkeypatch)
    # d = 16 / 32 (graph_node.md:44-66 takes any width): the state and the parameters run zero-padded on the 64-wide persistent
    # kernels -- one tile per workgroup, tile pairs (33 000) and tile rounds (40 000 with tanh) alike; sigmoid makes the padded
    # columns of the state non-zero (sigmoid(0) = 0.5), which must not leak into the real ones; a block-diagonal batch of members
    dt = 0.1
    g, og, params = spatial_case(N, 4 * N, d, seed=N + d)
    gg = ng.batch([g] + [g.copy() for _ in range(members - 1)]) if members > 1 else g
    rhs = ng.Chain(ng.GCNConv((d, d), act, initialgraph=gg), ng.GCNConv((d, d), act, initialgraph=gg))
    node = ng.NeuralODE(rhs, solver=tab, n_steps=nsteps, dt=dt)
    ps, st = ng.setup(0, node)
    for k, name in enumerate(["layer_1", "layer_2"]):
        ps[name]["weight"] = torch.as_tensor(params[k]["weight"].astype(np.float32))
        ps[name]["bias"] = torch.as_tensor(params[k]["bias"].astype(np.float32))
    ps = ng.to_device(ps, DEV)
    for lp in ps.values():
        for v in lp.values():
            v.requires_grad_(True)
    rng = np.random.default_rng(N + 1)
    u0, R = rng.normal(size=(d, N * members)), rng.normal(size=(d, N * members))
    u = torch.as_tensor(u0.astype(np.float32), device=DEV).requires_grad_(True)
    uT, _ = node(u, ps, st)
    plan = next(iter(node._plans.values()))[0]
    assert {"persistent_fwd", "persistent_bwd", "widened"} <= plan.flags(), plan.flags()
    (uT * torch.as_tensor(R.astype(np.float32), device=DEV)).sum().backward()
    assert not plan.fault()
    accW = [0, 0]; accb = [0, 0]
    for m in range(members):
        sl = slice(m * N, (m + 1) * N)
        uTo, du0o, acc = _oracle_node_with_seed(params, og, u0[:, sl], R[:, sl], O.TABLEAUS[tab], dt, nsteps, act)
        close(uT[:, sl], uTo, rtol=2e-4, what=f"u(T) member {m}")
        if act == "relu":
            bad = (torch.abs(u.grad[:, sl].double().cpu() - torch.as_tensor(du0o)) > 1e-4 + 5e-4 * torch.abs(torch.as_tensor(du0o))).any(0)
            assert bad.double().mean() <= 5e-3, f"du0: {int(bad.sum())} of {N} nodes off (relu kinks allow a few)"
        else:
            close(u.grad[:, sl], du0o, rtol=5e-4, atol=1e-4, what=f"du0 member {m}")
        for k in range(2):
            accW[k] = accW[k] + acc[k]["weight"]; accb[k] = accb[k] + acc[k]["bias"]
    for k, name in enumerate(["layer_1", "layer_2"]):
        close(ps[name]["weight"].grad, accW[k], rtol=5e-4, atol=1e-3, what=f"dW{k + 1}")
        close(ps[name]["bias"].grad, accb[k], rtol=5e-4, atol=1e-3, what=f"db{k + 1}")


def _oracle_weighted_node(params, og, u0, seed, tableau, dt, nsteps, act):
    """the chain of two GCNConv(use_edge_weight=true) as ODE right-hand side: messages weighted by the graph's stored weights
    (src/layers.jl:230), unweighted degree in the normalisation (:224), from the oracle's layer and its pullback"""
    def rhs(u):
        y1, c1 = O.gcn_conv(u, params[0]["weight"], params[0]["bias"], og, act, True, True)
        y2, c2 = O.gcn_conv(y1, params[1]["weight"], params[1]["bias"], og, act, True, True)
        return y2, (c1, c2)

    def vjp(cache, kbar):
        g2 = O.gcn_conv_backward(cache[1], kbar)
        g1 = O.gcn_conv_backward(cache[0], g2["x"])
        return g1["x"], [dict(weight=g1["weight"], bias=g1.get("bias")), dict(weight=g2["weight"], bias=g2.get("bias"))]
    uT, tape = O.rk_solve(rhs, u0, tableau, dt, nsteps)
    acc = [dict(weight=np.zeros_like(p["weight"]), bias=np.zeros_like(p["bias"])) for p in params]

    def accumulate(pg):
        for A, G in zip(acc, pg):
            A["weight"] += G["weight"]
            A["bias"] += G["bias"].reshape(A["bias"].shape)
    du0 = O.rk_adjoint(vjp, tape, seed.copy(), tableau, dt, accumulate)
    return uT, du0, acc


@pytest.mark.parametrize("d,tab,N,nsteps,act,rounds", [(64, "tsit5", 1000, 3, "relu", False), (64, "euler", 2048, 4, "tanh", False), (32, "tsit5", 1500, 2, "relu", False),
                                                       (64, "tsit5", 16384, 2, "relu", False), (64, "tsit5", 16384, 2, "relu", True), (64, "tsit5", 40000, 2, "tanh", False),
                                                       (64, "tsit5", 77, 2, "swish", False), (64, "euler", 1000, 3, "relu", True)])
def test_node_persistent_plan_weighted_graph_against_oracle_and_replayed_plan(d, tab, N, nsteps, act, rounds, monkeypatch):
    needs_persistent_plan(monkeypatch)
    # GCNConv(use_edge_weight=true) on a graph with stored edge weights (src/layers.jl:206-231) as the chain of graph_node.md:78.
    # Graphs of one wave of workgroups run on the one-tile kernels' WGT forms (slot weights in the LDS of one W: layer 1's W as
    # register fragments in the forward, unpadded and swizzled in the adjoint), larger ones -- and `rounds` -- on the tile-round
    # kernels.  u(T), du0 and the parameter gradients against the float64 oracle; u(T) and du0 bitwise equal to the replayed plan
    # (same order of operations per row).
    dt = 0.1
    if rounds:
        monkeypatch.setenv("NGPDE_WEIGHTED_TILE_ROUNDS", "1")
    else:
        monkeypatch.delenv("NGPDE_WEIGHTED_TILE_ROUNDS", raising=False)
    _, s, t = S.closest_pairs_graph(N, 4 * N, seed=N + 3)
    rng = np.random.default_rng(N + d)
    ew = (0.25 + rng.random(s.size)).astype(np.float32)
    g = ng.GNNGraph(s, t, num_nodes=N, index_base=0, edge_weight=ew)
    og = O.Graph(s, t, num_nodes=N, index_base=0, edge_weight=ew)
    params = [dict(weight=S.glorot_uniform(N + 10 + k, d, d), bias=rng.normal(size=(d, 1)) * 0.1) for k in range(2)]
    u0, R = rng.normal(size=(d, N)), rng.normal(size=(d, N))

    def solve():
        rhs = ng.Chain(ng.GCNConv((d, d), act, initialgraph=g, use_edge_weight=True), ng.GCNConv((d, d), act, initialgraph=g, use_edge_weight=True))
        node = ng.NeuralODE(rhs, solver=tab, n_steps=nsteps, dt=dt)
        ps, st = ng.setup(0, node)
        for k, name in enumerate(["layer_1", "layer_2"]):
            ps[name]["weight"] = torch.as_tensor(params[k]["weight"].astype(np.float32))
            ps[name]["bias"] = torch.as_tensor(params[k]["bias"].astype(np.float32))
        ps = ng.to_device(ps, DEV)
        for lp in ps.values():
            for v in lp.values():
                v.requires_grad_(True)
        u = torch.as_tensor(u0.astype(np.float32), device=DEV).requires_grad_(True)
        uT, _ = node(u, ps, st)
        plan = next(iter(node._plans.values()))[0]
        (uT * torch.as_tensor(R.astype(np.float32), device=DEV)).sum().backward()
        assert not plan.fault()
        return uT.detach(), u.grad, ps, plan.flags()

    uT, du0, ps, flags = solve()
    assert {"persistent_fwd", "persistent_bwd"} <= flags, flags
    assert ("tile_rounds" in flags) == (rounds or N > 16384), flags
    assert ("widened" in flags) == (d != 64)
    uTo, du0o, acc = _oracle_weighted_node(params, og, u0, R, O.TABLEAUS[tab], dt, nsteps, act)
    close(uT, uTo, rtol=2e-4, what="u(T)")
    if act == "relu":
        bad = (torch.abs(du0.double().cpu() - torch.as_tensor(du0o)) > 1e-4 + 5e-4 * torch.abs(torch.as_tensor(du0o))).any(0)
        assert bad.double().mean() <= 5e-3, f"du0: {int(bad.sum())} of {N} nodes off (relu kinks allow a few)"
    else:
        close(du0, du0o, rtol=5e-4, atol=1e-4, what="du0")
    for k, name in enumerate(["layer_1", "layer_2"]):
        close(ps[name]["weight"].grad, acc[k]["weight"], rtol=5e-4, atol=1e-3, what=f"dW{k + 1}")
        close(ps[name]["bias"].grad, acc[k]["bias"], rtol=5e-4, atol=1e-3, what=f"db{k + 1}")
    if d == 64:
        monkeypatch.setenv("NGPDE_NO_PERSISTENT", "1")
        uT2, du02, _, flags2 = solve()
        assert "persistent_fwd" not in flags2
        assert torch.equal(uT, uT2) and torch.equal(du0, du02)


@pytest.mark.parametrize("act", ["tanh", "relu"])
def test_node_persistent_forward_only_plan_any_activation(act, monkeypatch):
    needs_persistent_plan(monkeypatch)
    # without gradients the persistent forward launch takes any activation (the adjoint launch is relu-only: sign-bit tape)
    N, d, nsteps, dt = 1500, 64, 4, 0.05
    g, og, params = spatial_case(N, 4 * N, d, seed=5)
    rhs = ng.Chain(ng.GCNConv((d, d), act, initialgraph=g), ng.GCNConv((d, d), act, initialgraph=g))
    node = ng.NeuralODE(rhs, solver="tsit5", n_steps=nsteps, dt=dt)
    ps, st = ng.setup(0, node)
    for k, name in enumerate(["layer_1", "layer_2"]):
        ps[name]["weight"] = torch.as_tensor(params[k]["weight"].astype(np.float32))
        ps[name]["bias"] = torch.as_tensor(params[k]["bias"].astype(np.float32))
    ps = ng.to_device(ps, DEV)
    u0 = np.random.default_rng(6).normal(size=(d, N))
    with torch.no_grad():
        uT, _ = node(torch.as_tensor(u0.astype(np.float32), device=DEV), ps, st)
    plan = next(iter(node._plans.values()))[0]
    assert "persistent_fwd" in plan.flags() and "persistent_bwd" not in plan.flags()
    rhs_o, _ = O.gcn2_rhs(params, og, act)
    uTo, _ = O.rk_solve(rhs_o, u0, O.TABLEAUS["tsit5"], dt, nsteps)
    close(uT, uTo, rtol=2e-4, what=f"u(T) {act}")


def test_node_persistent_and_replayed_plans_agree_bitwise(monkeypatch):
    needs_persistent_plan(monkeypatch)
    # same arithmetic, operation for operation: u(T) and du0 of the persistent plan equal the replayed plan's bit for bit;
    # the parameter gradients differ only in the order the per-tile partial sums are added
    N, d, nsteps, dt = 4096, 64, 3, 0.02
    g, og, params = spatial_case(N, 4 * N, d, seed=8)
    rng = np.random.default_rng(9)
    u0 = torch.as_tensor(rng.normal(size=(d, N)).astype(np.float32), device=DEV)
    outs = {}
    for mode in ("persistent", "replayed"):
        if mode == "replayed":
            monkeypatch.setenv("NGPDE_NO_PERSISTENT", "1")
        rhs = ng.Chain(ng.GCNConv((d, d), "relu", initialgraph=g), ng.GCNConv((d, d), "relu", initialgraph=g))
        node = ng.NeuralODE(rhs, solver="tsit5", n_steps=nsteps, dt=dt)
        ps, st = ng.setup(0, node)
        for k, name in enumerate(["layer_1", "layer_2"]):
            ps[name]["weight"] = torch.as_tensor(params[k]["weight"].astype(np.float32))
            ps[name]["bias"] = torch.as_tensor(params[k]["bias"].astype(np.float32))
        ps = ng.to_device(ps, DEV)
        for lp in ps.values():
            for v in lp.values():
                v.requires_grad_(True)
        u = u0.clone().requires_grad_(True)
        uT, _ = node(u, ps, st)
        assert ("persistent_fwd" in next(iter(node._plans.values()))[0].flags()) == (mode == "persistent")
        uT.sum().backward()
        outs[mode] = (uT.detach().clone(), u.grad.clone(), ps["layer_1"]["weight"].grad.clone(), ps["layer_2"]["bias"].grad.clone())
    a, b = outs["persistent"], outs["replayed"]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert torch.allclose(a[2], b[2], rtol=1e-5, atol=1e-5) and torch.allclose(a[3], b[3], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("N,weighted", [(4096, False), (1000, False), (2048, True)])
def test_node_own_first_tables_change_the_order_of_a_sum_only(N, weighted, monkeypatch):
    needs_persistent_plan(monkeypatch)
    # a GCN solver plan reads its own slot tables (NGPDE_NODE_OWN_FIRST: a row's own-tile neighbours first, padded with the all-zero
    # row; csrc/node_persistent.hip own_first_tables_kernel).  Against the plan on the handle's order (NGPDE_NO_OWN_FIRST=1): the
    # same neighbours and weights in another order, so u(T), du0 and the parameter gradients agree to rounding, and both agree with
    # the float64 oracle; the own-first persistent plan and the own-first replayed plan stay bitwise equal (one table for both)
    d, nsteps, dt = 64, 3, 0.05
    g, og, params = spatial_case(N, 4 * N, d, seed=N + 3)
    if weighted:
        _, s, t = S.closest_pairs_graph(N, 4 * N, seed=N + 3)
        ew = (0.5 + np.random.default_rng(N).random(s.size)).astype(np.float32)
        g = ng.GNNGraph(s, t, num_nodes=N, index_base=0, edge_weight=ew)
        og = O.Graph(s, t, num_nodes=N, index_base=0, edge_weight=ew)
    rng = np.random.default_rng(N + 4)
    u0n, Rn = rng.normal(size=(d, N)), rng.normal(size=(d, N))
    u0 = torch.as_tensor(u0n.astype(np.float32), device=DEV)
    R = torch.as_tensor(Rn.astype(np.float32), device=DEV)
    outs = {}
    for mode in ("own_first", "handle_order", "own_first_replayed"):
        monkeypatch.delenv("NGPDE_NO_OWN_FIRST", raising=False)
        monkeypatch.delenv("NGPDE_NO_PERSISTENT", raising=False)
        if mode == "handle_order":
            monkeypatch.setenv("NGPDE_NO_OWN_FIRST", "1")
        if mode == "own_first_replayed":
            monkeypatch.setenv("NGPDE_NO_PERSISTENT", "1")
        kw = dict(initialgraph=g, use_edge_weight=True) if weighted else dict(initialgraph=g)
        rhs = ng.Chain(ng.GCNConv((d, d), "tanh", **kw), ng.GCNConv((d, d), "tanh", **kw))
        node = ng.NeuralODE(rhs, solver="tsit5", n_steps=nsteps, dt=dt)
        ps, st = ng.setup(0, node)
        for k, name in enumerate(["layer_1", "layer_2"]):
            ps[name]["weight"] = torch.as_tensor(params[k]["weight"].astype(np.float32))
            ps[name]["bias"] = torch.as_tensor(params[k]["bias"].astype(np.float32))
        ps = ng.to_device(ps, DEV)
        for lp in ps.values():
            for v in lp.values():
                v.requires_grad_(True)
        u = u0.clone().requires_grad_(True)
        uT, _ = node(u, ps, st)
        flags = next(iter(node._plans.values()))[0].flags()
        assert ("own_first" in flags) == (mode != "handle_order"), flags
        assert ("persistent_fwd" in flags) == (mode != "own_first_replayed"), flags
        (uT * R).sum().backward()
        outs[mode] = (uT.detach().clone(), u.grad.clone(), ps["layer_1"]["weight"].grad.clone(), ps["layer_2"]["bias"].grad.clone())
    a, b, c = outs["own_first"], outs["handle_order"], outs["own_first_replayed"]
    assert torch.equal(a[0], c[0]) and torch.equal(a[1], c[1])
    assert not torch.equal(a[0], b[0]), "the two orders give the same bits: the own-first tables were not read"
    for x, y, what in zip(a, b, ("u(T)", "du0", "dW1", "db2")):
        scale = float(y.abs().max())
        assert float((x - y).abs().max()) <= 2e-5 * scale + 1e-6, (what, float((x - y).abs().max()), scale)
    oracle = _oracle_weighted_node if weighted else _oracle_node_with_seed
    uTo, du0o, acc = oracle(params, og, u0n, Rn, O.TABLEAUS["tsit5"], dt, nsteps, "tanh")
    close(a[0], uTo, rtol=2e-4, what="u(T)")
    close(a[1], du0o, rtol=5e-4, atol=1e-4, what="du0")


@pytest.mark.parametrize("K,N,tab,nsteps", [(2, 1000, "tsit5", 3), (3, 1000, "euler", 4), (8, 2048, "tsit5", 2), (2, 16384, "tsit5", 4)])
def test_node_batch_interleaved_members_equal_member_by_member_bitwise(K, N, tab, nsteps, monkeypatch):
    # a batch of identical structures runs two members at a time per workgroup (one computes while the other's rows travel);
    # per member the arithmetic is that of the member-by-member launch (NGPDE_NO_INTERLEAVE=1) operation for operation: u(T)
    # and du0 bit for bit, the parameter gradients to rounding (summed over the members in another order).  K = 3: an odd last
    # member alone in slot 0; N = 16 384: the 512-tile / two-workgroups-per-CU regime of the bench's `batched` leg
    needs_persistent_plan(monkeypatch)
    d, dt = 64, 0.05
    g, og, params = spatial_case(N, 4 * N, d, seed=300 + K)
    rng = np.random.default_rng(301 + K)
    u0 = torch.as_tensor(rng.normal(size=(d, K * N)).astype(np.float32), device=DEV)
    R = torch.as_tensor(rng.normal(size=(d, K * N)).astype(np.float32), device=DEV)
    outs = {}
    for mode in ("interleaved", "member_by_member"):
        if mode == "member_by_member":
            monkeypatch.setenv("NGPDE_NO_INTERLEAVE", "1")
        else:
            monkeypatch.delenv("NGPDE_NO_INTERLEAVE", raising=False)
        gb = ng.batch([g] + [g.copy() for _ in range(K - 1)])
        rhs = ng.Chain(ng.GCNConv((d, d), "relu", initialgraph=gb), ng.GCNConv((d, d), "relu", initialgraph=gb))
        node = ng.NeuralODE(rhs, solver=tab, n_steps=nsteps, dt=dt)
        ps, st = ng.setup(0, node)
        for k, name in enumerate(["layer_1", "layer_2"]):
            ps[name]["weight"] = torch.as_tensor(params[k]["weight"].astype(np.float32))
            ps[name]["bias"] = torch.as_tensor(params[k]["bias"].astype(np.float32))
        ps = ng.to_device(ps, DEV)
        for lp in ps.values():
            for v in lp.values():
                v.requires_grad_(True)
        u = u0.clone().requires_grad_(True)
        uT, _ = node(u, ps, st)
        plan = next(iter(node._plans.values()))[0]
        assert plan.members == K and {"persistent_fwd", "persistent_bwd"} <= plan.flags()
        (uT * R).sum().backward()
        assert not plan.fault()
        outs[mode] = [uT.detach().clone(), u.grad.clone()] + [ps[l][k].grad.clone() for l in ("layer_1", "layer_2") for k in ("weight", "bias")]
        with torch.no_grad():                       # the forward-only plan (no tape) of the same batch
            uT2, _ = node(u0, ps, st)
        assert torch.equal(uT2, outs[mode][0])
    a, b = outs["interleaved"], outs["member_by_member"]
    assert torch.equal(a[0], b[0]), "u(T)"
    assert torch.equal(a[1], b[1]), "du0"
    for x, y in zip(a[2:], b[2:]):
        assert torch.allclose(x, y, rtol=2e-5, atol=2e-5 * float(y.abs().max()))
    # and member 0 / the last member against the float64 oracle of that member alone
    for m in (0, K - 1):
        sl = slice(m * N, (m + 1) * N)
        uTo, du0o, _ = _oracle_node_with_seed(params, og, u0[:, sl].cpu().double().numpy(), R[:, sl].cpu().double().numpy(),
                                              O.TABLEAUS[tab], dt, nsteps)
        close(a[0][:, sl], uTo, rtol=2e-4, what=f"u(T) member {m}")
        close(a[1][:, sl], du0o, rtol=5e-4, atol=1e-4, what=f"du0 member {m}")


@pytest.mark.parametrize("N,tab,nsteps", [(32768, "tsit5", 2), (20000, "euler", 3), (16416, "tsit5", 2)])
def test_node_persistent_tile_pairs_beyond_512_tiles(N, tab, nsteps, monkeypatch):
    # graphs of more 32-row tiles than co-resident workgroups (512 on the MI355X: 16 384 nodes) and at most twice as many: the
    # two-slot kernels with the slots = two TILES of the one trajectory (workgroup b holds tiles t and t + W).  u(T) and du0 bit
    # for bit equal to the replayed plan, everything against the float64 oracle; 20 000 nodes = 625 tiles (the last workgroup has
    # one tile), 16 416 = 513 tiles (one workgroup has two)
    needs_persistent_plan(monkeypatch)
    d, dt = 64, 0.05
    g, og, params = spatial_case(N, 4 * N, d, seed=N % 1000)
    rng = np.random.default_rng(N)
    u0n, Rn = rng.normal(size=(d, N)), rng.normal(size=(d, N))
    outs = {}
    for mode in ("pairs", "replayed"):
        if mode == "replayed":
            monkeypatch.setenv("NGPDE_NO_PERSISTENT", "1")
        rhs = ng.Chain(ng.GCNConv((d, d), "relu", initialgraph=g), ng.GCNConv((d, d), "relu", initialgraph=g))
        node = ng.NeuralODE(rhs, solver=tab, n_steps=nsteps, dt=dt)
        ps, st = ng.setup(0, node)
        for k, name in enumerate(["layer_1", "layer_2"]):
            ps[name]["weight"] = torch.as_tensor(params[k]["weight"].astype(np.float32))
            ps[name]["bias"] = torch.as_tensor(params[k]["bias"].astype(np.float32))
        ps = ng.to_device(ps, DEV)
        for lp in ps.values():
            for v in lp.values():
                v.requires_grad_(True)
        u = torch.as_tensor(u0n.astype(np.float32), device=DEV).requires_grad_(True)
        uT, _ = node(u, ps, st)
        plan = next(iter(node._plans.values()))[0]
        if mode == "pairs":
            assert {"persistent_fwd", "persistent_bwd", "tile_pairs"} <= plan.flags(), plan.flags()
        else:
            assert "persistent_fwd" not in plan.flags()
        (uT * torch.as_tensor(Rn.astype(np.float32), device=DEV)).sum().backward()
        assert not plan.fault()
        outs[mode] = [uT.detach().clone(), u.grad.clone()] + [ps[l][k].grad.clone() for l in ("layer_1", "layer_2") for k in ("weight", "bias")]
        with torch.no_grad():                       # the forward-only plan
            uT2, _ = node(u.detach(), ps, st)
        assert torch.equal(uT2, outs[mode][0])
    a, b = outs["pairs"], outs["replayed"]
    assert torch.equal(a[0], b[0]), "u(T)"
    assert torch.equal(a[1], b[1]), "du0"
    for x, y in zip(a[2:], b[2:]):
        assert torch.allclose(x, y, rtol=2e-5, atol=2e-5 * float(y.abs().max()))
    uTo, du0o, acc = _oracle_node_with_seed(params, og, u0n, Rn, O.TABLEAUS[tab], dt, nsteps)
    close(a[0], uTo, rtol=2e-4, what="u(T)")
    close(a[1], du0o, rtol=5e-4, atol=1e-4, what="du0")
    # (relu'(z) is decided by rounding where |z| is within an ulp of zero -- one or two of the 10^8 pre-activations of these solves;
    # each such kink moves a row of dW by |a_n| |K-bar| ~ 5e-2, in the oracle's float64 as much as here: the tight statement about
    # the parameter gradients is the comparison with the replayed plan above)
    for k in range(2):
        close(a[2 + 2 * k], acc[k]["weight"], rtol=5e-3, atol=1e-3, what=f"dW{k + 1}")


@pytest.mark.parametrize("N,tab,nsteps,act", [(40000, "tsit5", 2, "relu"), (33000, "euler", 3, "tanh"), (70000, "tsit5", 1, "relu")])
def test_node_persistent_tile_rounds_beyond_1024_tiles(N, tab, nsteps, act, monkeypatch):
    # graphs of more than two 32-row tiles per co-resident workgroup (> 32 768 nodes on the MI355X): k tiles per workgroup taking
    # turns, their state in memory (node_*_persistentK_kernel).  40 000 nodes = 1 250 tiles = 3 per workgroup with the last
    # workgroups short of one, 33 000 = 1 032 (just over the tile pairs' reach; tanh: the pre-activation tape), 70 000 = 2 188 = 5
    # per workgroup.  u(T) and du0 bit for bit equal to the replayed plan; 40 000 also against the float64 oracle.
    needs_persistent_plan(monkeypatch)
    d, dt = 64, 0.05
    g, og, params = spatial_case(N, 4 * N, d, seed=N % 1000)
    rng = np.random.default_rng(N)
    u0n, Rn = rng.normal(size=(d, N)), rng.normal(size=(d, N))
    outs = {}
    for mode in ("rounds", "replayed"):
        if mode == "replayed":
            monkeypatch.setenv("NGPDE_NO_PERSISTENT", "1")
        rhs = ng.Chain(ng.GCNConv((d, d), act, initialgraph=g), ng.GCNConv((d, d), act, initialgraph=g))
        node = ng.NeuralODE(rhs, solver=tab, n_steps=nsteps, dt=dt)
        ps, st = ng.setup(0, node)
        for k, name in enumerate(["layer_1", "layer_2"]):
            ps[name]["weight"] = torch.as_tensor(params[k]["weight"].astype(np.float32))
            ps[name]["bias"] = torch.as_tensor(params[k]["bias"].astype(np.float32))
        ps = ng.to_device(ps, DEV)
        for lp in ps.values():
            for v in lp.values():
                v.requires_grad_(True)
        u = torch.as_tensor(u0n.astype(np.float32), device=DEV).requires_grad_(True)
        uT, _ = node(u, ps, st)
        plan = next(iter(node._plans.values()))[0]
        if mode == "rounds":
            assert {"persistent_fwd", "persistent_bwd", "tile_rounds"} <= plan.flags(), plan.flags()
        else:
            assert "persistent_fwd" not in plan.flags()
        (uT * torch.as_tensor(Rn.astype(np.float32), device=DEV)).sum().backward()
        assert not plan.fault()
        outs[mode] = [uT.detach().clone(), u.grad.clone()] + [ps[l][k].grad.clone() for l in ("layer_1", "layer_2") for k in ("weight", "bias")]
        with torch.no_grad():                       # the forward-only plan
            uT2, _ = node(u.detach(), ps, st)
        assert torch.equal(uT2, outs[mode][0])
    a, b = outs["rounds"], outs["replayed"]
    assert torch.equal(a[0], b[0]), "u(T)"
    assert torch.equal(a[1], b[1]), "du0"
    for x, y in zip(a[2:], b[2:]):
        assert torch.allclose(x, y, rtol=2e-5, atol=2e-5 * float(y.abs().max()))
    if N == 40000:
        uTo, du0o, acc = _oracle_node_with_seed(params, og, u0n, Rn, O.TABLEAUS[tab], dt, nsteps)
        close(a[0], uTo, rtol=2e-4, what="u(T)")
        # du0: relu' is decided by rounding where a pre-activation is within an ulp of zero (3e7 of them here: a few flip), and each
        # kink moves du0 in the rows around it -- every node within 5e-4 of the largest entry except at most 0.5 % of them, those within 20x
        dcol = np.abs(a[1].cpu().double().numpy() - du0o).max(axis=0)
        bound = 5e-4 * np.abs(du0o).max() + 1e-4
        assert (dcol > bound).sum() <= 0.005 * dcol.size and dcol.max() <= 20 * bound, f"du0: {(dcol > bound).sum()} nodes beyond {bound:.2e}, max {dcol.max():.2e}"


def test_node_persistent_abort_poisons_outputs_and_the_plan_refuses_further_work(monkeypatch):
    # a persistent launch whose waits give up (forced here: NGPDE_DEBUG_FORCE_ABORT=1 starts the launch with its abort word set;
    # in production: another kernel holding the compute units for ~2 s) writes NaN outputs and latches the plan's fault word,
    # which lives in pinned host memory: the NEXT entry of the plan fails with ERR_STATE instead of computing on garbage
    needs_persistent_plan(monkeypatch)
    from ngpde_amd import _lib
    from ngpde_amd.node import _Plan
    N, d = 4096, 64
    g, og, params = spatial_case(N, 4 * N, d, seed=12)
    lib, p = _lib.load(), _lib.ptr
    plan = _Plan(g.handle((True, None, False)), d, _lib.ACT["relu"], "tsit5", 3, 0.05, True)
    assert "persistent_fwd" in plan.flags()
    dv = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32), device=DEV)
    u0 = dv(np.random.default_rng(1).normal(size=(N, d)))
    w1, w2, b = dv(params[0]["weight"].T), dv(params[1]["weight"].T), torch.zeros(d, device=DEV)
    uT = torch.empty_like(u0)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.ngpde_node_gcn2_forward(plan.ptr, p(u0), p(w1), p(b), p(w2), p(b), p(uT), st))
    assert not plan.fault() and torch.isfinite(uT).all()
    monkeypatch.setenv("NGPDE_DEBUG_FORCE_ABORT", "1")
    _lib.check(lib.ngpde_node_gcn2_forward(plan.ptr, p(u0), p(w1), p(b), p(w2), p(b), p(uT), st))
    monkeypatch.delenv("NGPDE_DEBUG_FORCE_ABORT")
    assert plan.fault() and torch.isnan(uT).all()
    with pytest.raises(_lib.NgpdeError, match="gave up waiting"):
        _lib.check(lib.ngpde_node_gcn2_forward(plan.ptr, p(u0), p(w1), p(b), p(w2), p(b), p(uT), st))
    fresh = _Plan(g.handle((True, None, False)), d, _lib.ACT["relu"], "tsit5", 3, 0.05, True)      # a new plan works
    _lib.check(lib.ngpde_node_gcn2_forward(fresh.ptr, p(u0), p(w1), p(b), p(w2), p(b), p(uT), st))
    assert not fresh.fault() and torch.isfinite(uT).all()


@pytest.mark.parametrize("tab,capture,save_start", [("tsit5", False, True), ("euler", False, False), ("tsit5", True, True)])
def test_node_saveat_returns_the_solution_at_the_saved_times(tab, capture, save_start):
    # NeuralODE(model, tspan, Tsit5(); saveat = dt_train) of docs/src/tutorials/VMH.md:85: the output is the (D x N x T) array of the
    # solution at t0, t0 + saveat, ..., and the loss reads every time point (VMH.md:104-108).  Against the float64 oracle solved segment
    # by segment; the discrete adjoint carries the cotangent of every saved state into lambda at its time.
    N, d, nsteps, k, dt = 300, 16, 6, 2, 0.1
    g, og, params = spatial_case(N, 4 * N, d, seed=77)
    rhs = ng.Chain(ng.GCNConv((d, d), "tanh", initialgraph=g), ng.GCNConv((d, d), "tanh", initialgraph=g))
    node = ng.NeuralODE(rhs, solver=tab, n_steps=nsteps, dt=dt, saveat=k * dt, save_start=save_start, capture=capture)
    ps, st = ng.setup(0, node)
    for j, name in enumerate(["layer_1", "layer_2"]):
        ps[name]["weight"] = torch.as_tensor(params[j]["weight"].astype(np.float32))
        ps[name]["bias"] = torch.as_tensor(params[j]["bias"].astype(np.float32))
    ps = ng.to_device(ps, DEV)
    for lp in ps.values():
        for v in lp.values():
            v.requires_grad_(True)
    rng = np.random.default_rng(78)
    u0 = rng.normal(size=(d, N))
    T = nsteps // k + (1 if save_start else 0)
    R = rng.normal(size=(d, N, T))
    for rep in range(2 if capture else 1):          # (captured: the second call replays the graphs)
        for lp in ps.values():
            for v in lp.values():
                v.grad = None
        u = torch.as_tensor(u0.astype(np.float32), device=DEV).requires_grad_(True)
        Y, _ = node(u, ps, st)
        assert tuple(Y.shape) == (d, N, T)
        (Y * torch.as_tensor(R.astype(np.float32), device=DEV)).sum().backward()
    rhs_o, vjp_o = O.gcn2_rhs(params, og, "tanh")
    states, tapes, cur = [u0], [], u0
    for _ in range(nsteps // k):
        cur, tape = O.rk_solve(rhs_o, cur, O.TABLEAUS[tab], dt, k)
        states.append(cur); tapes.append(tape)
    saved = states if save_start else states[1:]
    for j in range(T):
        close(Y[:, :, j], saved[j], rtol=2e-4, what=f"u(t_{j})")
    acc = [dict(weight=np.zeros_like(p["weight"]), bias=np.zeros_like(p["bias"])) for p in params]

    def accumulate(pg):
        for A, G in zip(acc, pg):
            A["weight"] += G["weight"]
            A["bias"] += G["bias"].reshape(A["bias"].shape)
    lam = R[:, :, T - 1].copy()
    for seg in range(nsteps // k - 1, -1, -1):
        lam = O.rk_adjoint(vjp_o, tapes[seg], lam, O.TABLEAUS[tab], dt, accumulate)
        j = seg if save_start else seg - 1
        if j >= 0:
            lam = lam + R[:, :, j]
    close(u.grad, lam, rtol=5e-4, atol=1e-4, what="du0")
    for j, name in enumerate(["layer_1", "layer_2"]):
        close(ps[name]["weight"].grad, acc[j]["weight"], rtol=5e-4, atol=1e-3, what=f"dW{j + 1}")
        close(ps[name]["bias"].grad, acc[j]["bias"], rtol=5e-4, atol=1e-3, what=f"db{j + 1}")


def test_node_saveat_must_be_a_whole_number_of_steps():
    rhs = ng.Chain(ng.GCNConv((4, 4), "tanh"), ng.GCNConv((4, 4), "tanh"))
    with pytest.raises(ng.ArgumentError, match="whole number of steps"):
        ng.NeuralODE(rhs, n_steps=10, dt=0.1, saveat=0.25)
    with pytest.raises(ng.ArgumentError, match="whole number of steps"):
        ng.NeuralODE(rhs, n_steps=10, dt=0.1, saveat=0.3)
    assert ng.NeuralODE(rhs, n_steps=10, dt=0.1, saveat=0.5).save_every == 5


def test_node_device_resident_plans_check_their_inputs_against_the_graph():
    # the plans' C entries take pointers only and walk the handle's rows: a state with another node count or width, or a weight of
    # another shape, must raise the reference's DimensionMismatch BEFORE a kernel reads or writes out of bounds (GCNConv.__call__
    # would have raised it; the plan path skips the layer's own checks)
    from ngpde_amd import _lib
    N, d = 1024, 64
    g, og, params = spatial_case(N, 4 * N, d, seed=21)
    rhs = ng.Chain(ng.GCNConv((d, d), "relu", initialgraph=g), ng.GCNConv((d, d), "relu", initialgraph=g))
    node = ng.NeuralODE(rhs, solver="euler", n_steps=2, dt=0.1)
    ps, st = ng.setup(0, node)
    ps = ng.to_device(ps, DEV)
    good = torch.zeros(d, N, device=DEV)
    assert node(good, ps, st)[0].shape == (d, N)
    for bad in (torch.zeros(d, N - 32, device=DEV), torch.zeros(d, N + 32, device=DEV), torch.zeros(d // 2, N, device=DEV)):
        with pytest.raises(_lib.DimensionMismatch):
            node(bad, ps, st)
    ps_bad = {"layer_1": dict(ps["layer_1"]), "layer_2": dict(ps["layer_2"])}
    ps_bad["layer_2"]["weight"] = torch.zeros(d, d // 2, device=DEV)
    with pytest.raises(_lib.DimensionMismatch):
        node(good, ps_bad, st)
    # the GAT-style layer as right-hand side (ngpde_node_gat_*)
    gat = ng.NeuralODE(ng.GATConv((64, 16), "relu", heads=4, initialgraph=g), solver="euler", n_steps=2, dt=0.1)
    gps, gst = ng.setup(1, gat)
    gps = ng.to_device(gps, DEV)
    assert gat(good, gps, gst)[0].shape == (d, N)
    if gat._plans and any(k[0] == "gat" for k in gat._plans if isinstance(k, tuple)):     # (the device-resident plan took it)
        for bad in (torch.zeros(d, N - 32, device=DEV), torch.zeros(d, N + 64, device=DEV)):
            with pytest.raises(_lib.DimensionMismatch):
                gat(bad, gps, gst)
        gbad = dict(gps)
        gbad["a"] = torch.zeros(2 * 16, 3, device=DEV)
        with pytest.raises(_lib.DimensionMismatch):
            gat(good, gbad, gst)
