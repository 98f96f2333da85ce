"""GPU parity tests of the device-resident NeuralODE(VMHConv) plan (ngpde_node_vmh_*: /root/reference/docs/src/tutorials/VMH.md:75-89
-- phi and gamma Dense stacks on a scalar state, /root/reference/src/layers.jl:402-416 as the right-hand side of every Runge-Kutta
stage): the float64 oracle's rk_solve / rk_adjoint, the generic solver on the same inputs, shape checks, the abort protocol."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import ngpde_amd as ng
from ngpde_amd import _lib, synth as S
from oracle import ngpde_oracle as O
from test_mp_gpu import check_grads, close, mlp_grad_pairs, omlp, prep

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def spatial(N, seed, pd=2):
    pts, s, t = S.closest_pairs_graph(N, 3 * N, seed=seed)
    extra = np.random.default_rng(seed).random((1, N))           # (a third coordinate for pd = 3; the graph is the planar one)
    nd = {"x": np.concatenate([pts.T, extra], axis=0)[:pd].copy()}
    return ng.GNNGraph(s, t, num_nodes=N, index_base=0, ndata=nd), O.Graph(s, t, num_nodes=N, index_base=0, ndata=nd)


def tutorial_mlps(width=60, msg=40, depth=4, act="tanh", pd=2):
    hid = [ng.Dense(width, width, act) for _ in range(depth - 2)]
    phi = ng.Chain(ng.Dense(2 + pd, width, act), *hid, ng.Dense(width, msg))
    gam = ng.Chain(ng.Dense(1 + msg, width, act), *[ng.Dense(width, width, act) for _ in range(depth - 2)], ng.Dense(width, 1))
    return phi, gam


def plan_flags(node):
    return sorted({f for pool in node._plans.values() for p in pool for f in p.flags()})


def oracle_solve(phi, gam, ps, og, u0, tab_name, dt, n_steps, R, aggr="mean"):
    ophi, ogam = omlp(phi, ps["ϕ"]), omlp(gam, ps["γ"])
    tab = O.TABLEAUS[tab_name]
    gphi = [dict(weight=np.zeros_like(L["weight"]), bias=np.zeros_like(L["bias"])) for L in ophi]
    ggam = [dict(weight=np.zeros_like(L["weight"]), bias=np.zeros_like(L["bias"])) for L in ogam]

    def vjp(cache, kbar):
        gr = O.vmh_conv_backward(cache, kbar)
        return gr["x"], gr

    def accumulate(gr):
        for dst, src in ((gphi, gr["phi"]), (ggam, gr["gamma"])):
            for d_, s_ in zip(dst, src):
                d_["weight"] += s_["weight"]
                d_["bias"] += np.asarray(s_["bias"]).reshape(d_["bias"].shape)
    uT, tape = O.rk_solve(lambda u: O.vmh_conv(u, ophi, ogam, og, aggr=aggr), u0.astype(np.float64), tab, dt, n_steps)
    du0 = O.rk_adjoint(vjp, tape, R, tab, dt, accumulate)
    return uT, du0, gphi, ggam


@pytest.mark.parametrize("solver,n_steps,act,aggr,depth,pd,N", [
    ("tsit5", 2, "tanh", "mean", 4, 2, 700),      # the tutorial's model
    ("euler", 3, "tanh", "+", 3, 2, 700),
    ("tsit5", 1, "relu", "mean", 2, 1, 700),
    ("euler", 2, "sigmoid", "mean", 3, 3, 700),
    ("tsit5", 2, "tanh", "mean", 4, 2, 4700),     # more half tiles than compute units: two per workgroup, taking turns (tile rounds)
    ("euler", 3, "relu", "+", 3, 2, 9100),        # ... and three
])
def test_vmh_resident_solve_and_adjoint_against_the_oracle(solver, n_steps, act, aggr, depth, pd, N, monkeypatch):
    monkeypatch.delenv("NGPDE_NO_VMH_NODE", raising=False)
    monkeypatch.delenv("NGPDE_NO_VMH_ROUNDS", raising=False)
    dt = 0.05
    g, og = spatial(N, 31 + depth, pd=pd)
    phi, gam = tutorial_mlps(depth=depth, act=act, pd=pd)
    node = ng.NeuralODE(ng.VMHConv(phi, gam, aggr=aggr, initialgraph=g), solver=solver, n_steps=n_steps, dt=dt)
    ps0, st = ng.setup(3, node)
    ps = prep(ps0, 3)
    rng = np.random.default_rng(9)
    u0 = rng.normal(size=(1, N)).astype(np.float32)
    R = rng.normal(size=(1, N))
    u = torch.as_tensor(u0, device=DEV).requires_grad_(True)
    uT, _ = node(u, ps, st)
    assert "vmh" in plan_flags(node), plan_flags(node)       # the device-resident plan ran, not the generic solver
    uTo, du0, gphi, ggam = oracle_solve(phi, gam, ps, og, u0, solver, dt, n_steps, R, aggr=aggr)
    close(uT, uTo, rtol=2e-4)
    (uT * torch.as_tensor(R, dtype=torch.float32, device=DEV)).sum().backward()
    n1, o1 = mlp_grad_pairs(ps["ϕ"], gphi, phi)
    n2, o2 = mlp_grad_pairs(ps["γ"], ggam, gam)
    check_grads(ps, (n1 + n2, o1 + o2), u, du0)
    assert not any(p.fault() for pool in node._plans.values() for p in pool)


def test_vmh_resident_plan_equals_the_generic_solver_at_the_tutorial_shape(monkeypatch):
    # 3 000 points, 6 neighbours, phi = 4 => 60 => 60 => 60 => 40, gamma = 41 => 60 => 60 => 60 => 1 (docs/src/tutorials/VMH.md:75-83),
    # Tsit5 x 4: values, du0 and all 16 parameter gradients of both paths; repeated solves of the plan are bitwise equal
    nv, steps = 3000, 4
    pts = torch.as_tensor(S.uniform01(41, 2 * nv).reshape(2, nv).astype(np.float32), device=DEV)
    gv = ng.GNNGraph(ng.knn_graph(pts, 6), ndata={"x": pts})
    phi, gam = tutorial_mlps()
    u0 = torch.as_tensor(S.normal(42, nv).reshape(1, nv).astype(np.float32), device=DEV)
    R = torch.as_tensor(S.normal(43, nv).reshape(1, nv).astype(np.float32), device=DEV)
    res = {}
    for mode in ("resident", "generic", "resident2"):
        if mode == "generic":
            monkeypatch.setenv("NGPDE_NO_VMH_NODE", "1")
        else:
            monkeypatch.delenv("NGPDE_NO_VMH_NODE", raising=False)
        node = ng.NeuralODE(ng.VMHConv(phi, gam, initialgraph=gv), solver="tsit5", n_steps=steps, dt=0.05)
        ps0, st = ng.setup(4, node)
        ps = prep(ps0, 4)
        u = u0.clone().requires_grad_(True)
        uT, _ = node(u, ps, st)
        (uT * R).sum().backward()
        assert ("vmh" in plan_flags(node)) == (mode != "generic")
        n1, _ = mlp_grad_pairs(ps["ϕ"], [{"weight": 0, "bias": 0}] * 4, phi)
        n2, _ = mlp_grad_pairs(ps["γ"], [{"weight": 0, "bias": 0}] * 4, gam)
        res[mode] = [uT.detach().clone(), u.grad.clone()] + [p.grad.clone() for _, p in n1 + n2]
    for a, b in zip(res["resident"], res["generic"]):
        scale = float(b.abs().max())
        assert float((a - b).abs().max()) <= 2e-5 * scale + 1e-6
    for a, b in zip(res["resident"], res["resident2"]):
        assert torch.equal(a, b)     # no atomics, fixed summation orders, and every hand-off waited for


@pytest.mark.parametrize("solver,save_start,batched", [("tsit5", True, False), ("euler", False, False), ("tsit5", True, True), ("tsit5", True, "rounds")])
def test_vmh_resident_saveat_against_the_oracle_segment_by_segment(solver, save_start, batched, monkeypatch):
    # NeuralODE(gnn, tspan, Tsit5(); saveat = dt_train) (docs/src/tutorials/VMH.md:85): the (1 x N x T) array of the solution at the
    # saved times on the device-resident plan, the loss reads every one of them (:104-108); the oracle solves segment by segment and
    # its adjoint walks the segments backwards, adding each saved state's cotangent.  batched: a block-diagonal batch of three
    # point clouds (:132-134) is one graph to the plan
    monkeypatch.delenv("NGPDE_NO_VMH_NODE", raising=False)
    monkeypatch.delenv("NGPDE_NO_VMH_ROUNDS", raising=False)
    k, nseg, dt = 2, 3, 0.05
    if batched == "rounds":      # a batch of point clouds with more half tiles than compute units (VMH.md:120: 24 clouds of 3 000 points)
        gs, ogs = zip(*[spatial(1500 + 100 * j, 60 + j) for j in range(3)])
        g, og = ng.batch(list(gs)), O.batch(list(ogs))
    elif batched:
        gs, ogs = zip(*[spatial(220 + 16 * j, 50 + j) for j in range(3)])
        g, og = ng.batch(list(gs)), O.batch(list(ogs))
    else:
        g, og = spatial(600, 44)
    N = g.num_nodes
    phi, gam = tutorial_mlps(depth=3)
    node = ng.NeuralODE(ng.VMHConv(phi, gam, initialgraph=g), solver=solver, n_steps=k * nseg, dt=dt, saveat=k * dt, save_start=save_start)
    ps0, st = ng.setup(6, node)
    ps = prep(ps0, 6)
    rng = np.random.default_rng(10)
    u0 = rng.normal(size=(1, N)).astype(np.float32)
    T = nseg + int(save_start)
    R = rng.normal(size=(1, N, T))
    u = torch.as_tensor(u0, device=DEV).requires_grad_(True)
    us, _ = node(u, ps, st)
    assert "vmh" in plan_flags(node) and tuple(us.shape) == (1, N, T)
    ophi, ogam = omlp(phi, ps["ϕ"]), omlp(gam, ps["γ"])
    tab = O.TABLEAUS[solver]
    gphi = [dict(weight=np.zeros_like(L["weight"]), bias=np.zeros_like(L["bias"])) for L in ophi]
    ggam = [dict(weight=np.zeros_like(L["weight"]), bias=np.zeros_like(L["bias"])) for L in ogam]

    def vjp(cache, kbar):
        gr = O.vmh_conv_backward(cache, kbar)
        return gr["x"], gr

    def accumulate(gr):
        for dst, src in ((gphi, gr["phi"]), (ggam, gr["gamma"])):
            for d_, s_ in zip(dst, src):
                d_["weight"] += s_["weight"]
                d_["bias"] += np.asarray(s_["bias"]).reshape(d_["bias"].shape)
    states, tapes, cur = [u0.astype(np.float64)], [], u0.astype(np.float64)
    for _ in range(nseg):
        cur, tape = O.rk_solve(lambda x: O.vmh_conv(x, ophi, ogam, og), cur, tab, dt, k)
        states.append(cur); tapes.append(tape)
    want = np.stack(states[0 if save_start else 1:], axis=2)
    close(us, want, rtol=2e-4)
    (us * torch.as_tensor(R, dtype=torch.float32, device=DEV)).sum().backward()
    off = 1 if save_start else 0
    lam = np.zeros_like(cur)
    for j in range(nseg - 1, -1, -1):
        lam = O.rk_adjoint(vjp, tapes[j], lam + R[:, :, j + off], tab, dt, accumulate)
    if save_start:
        lam = lam + R[:, :, 0]
    n1, o1 = mlp_grad_pairs(ps["ϕ"], gphi, phi)
    n2, o2 = mlp_grad_pairs(ps["γ"], ggam, gam)
    check_grads(ps, (n1 + n2, o1 + o2), u, lam)


def test_vmh_resident_plan_leaves_unsupported_models_to_the_generic_solver(monkeypatch):
    monkeypatch.delenv("NGPDE_NO_VMH_NODE", raising=False)
    N = 400
    pts = torch.as_tensor(S.uniform01(5, 2 * N).reshape(2, N).astype(np.float32), device=DEV)
    g = ng.GNNGraph(ng.knn_graph(pts, 5), ndata={"x": pts})      # (every node has neighbours: a max over none is -Inf, as NNlib's)
    # a state of three features, a max aggregation, a message MLP wider than 64: none of them is the plan's; the solve runs on the generic solver
    phi3 = ng.Chain(ng.Dense(2 * 3 + 2, 16, "tanh"), ng.Dense(16, 8))
    gam3 = ng.Chain(ng.Dense(3 + 8, 16, "tanh"), ng.Dense(16, 3))
    cases = [(ng.NeuralODE(ng.VMHConv(phi3, gam3, initialgraph=g), solver="tsit5", n_steps=2, dt=0.05), 3)]
    phi, gam = tutorial_mlps(depth=3)
    cases.append((ng.NeuralODE(ng.VMHConv(phi, gam, aggr="max", initialgraph=g), solver="tsit5", n_steps=2, dt=0.05), 1))
    wide = ng.Chain(ng.Dense(4, 80, "tanh"), ng.Dense(80, 40))           # wider than the 64 columns the kernels stage
    cases.append((ng.NeuralODE(ng.VMHConv(wide, gam, initialgraph=g), solver="tsit5", n_steps=2, dt=0.05), 1))
    for node, h in cases:
        ps, st = ng.setup(1, node)
        ps = prep(ps, 1)
        u = torch.randn(h, N, device=DEV, requires_grad=True)
        out, _ = node(u, ps, st)
        out.sum().backward()
        assert "vmh" not in plan_flags(node)
        assert torch.isfinite(out).all() and torch.isfinite(u.grad).all()


def test_vmh_abi_rejects_null_and_mismatched_arguments(monkeypatch):
    monkeypatch.delenv("NGPDE_NO_VMH_NODE", raising=False)       # (the suite may run under that switch)
    lib = _lib.load()
    N = 300
    g, _ = spatial(N, 8)
    ia = lambda v: (C.c_int32 * len(v))(*v)
    acts = [_lib.ACT["tanh"], _lib.ACT["identity"]]
    dims_p, dims_g, mean = [4, 60, 40], [41, 60, 1], _lib.AGGR["mean"]
    h = g.handle()
    sup = lambda gp, hd, dp, dg, aggr: lib.ngpde_node_vmh_supported(gp, hd, 2, 2, ia(dp), ia(acts), 2, ia(dg), ia(acts), aggr)
    assert sup(h.ptr, 1, dims_p, dims_g, mean) == 1
    assert sup(None, 1, dims_p, dims_g, mean) == 0
    assert sup(h.ptr, 2, dims_p, dims_g, mean) == 0                   # a state of two features
    assert sup(h.ptr, 1, [4, 80, 40], dims_g, mean) == 0              # wider than 64
    assert sup(h.ptr, 1, dims_p, [40, 60, 1], mean) == 0              # gamma's input != 1 + the message width
    assert sup(h.ptr, 1, dims_p, dims_g, _lib.AGGR["max"]) == 0
    pos = torch.zeros(N, 2, device=DEV)
    out = C.c_void_p()
    mk = lambda gp, posp, outp: lib.ngpde_node_vmh_create(gp, 1, 2, posp, 2, ia(dims_p), ia(acts), 2, ia(dims_g), ia(acts), mean,
                                                          _lib.TABLEAU["tsit5"], 2, 0.1, 1, outp)
    assert mk(None, _lib.ptr(pos), C.byref(out)) == _lib.ERR_INVALID_ARGUMENT and not out.value
    assert mk(h.ptr, None, C.byref(out)) == _lib.ERR_INVALID_ARGUMENT and not out.value
    assert mk(h.ptr, _lib.ptr(pos), None) == _lib.ERR_INVALID_ARGUMENT
    assert lib.ngpde_node_vmh_destroy(None) == _lib.OK       # a no-op, as every destroy of the ABI
    assert lib.ngpde_node_vmh_tape_bytes(None) == 0
    f = C.c_int32()
    assert lib.ngpde_node_vmh_fault(None, None, C.byref(f)) == _lib.ERR_INVALID_ARGUMENT
    assert mk(h.ptr, _lib.ptr(pos), C.byref(out)) == _lib.OK and out.value
    u = torch.zeros(N, device=DEV)
    assert lib.ngpde_node_vmh_forward(out, None, None, None, None, None, _lib.ptr(u), None) == _lib.ERR_INVALID_ARGUMENT
    assert lib.ngpde_node_vmh_backward(out, None, None, _lib.ptr(u), _lib.ptr(u), None, None, None, None, None) in (_lib.ERR_INVALID_ARGUMENT, _lib.ERR_STATE)
    assert lib.ngpde_node_vmh_destroy(out) == _lib.OK


def test_vmh_plan_rejects_a_state_or_parameters_that_do_not_match_the_graph(monkeypatch):
    # the plan's C entries take pointers only (advisor, round 4): a state whose node count is not the graph's -- a forgotten updategraph in
    # the minibatch loop -- or a parameter tree that does not chain must raise the reference's DimensionMismatch (check_num_nodes / the
    # matrix product) BEFORE any kernel reads N floats through a buffer of N'
    monkeypatch.delenv("NGPDE_NO_VMH_NODE", raising=False)
    nv = 704
    g, _ = spatial(nv, 77)
    phi, gam = tutorial_mlps(width=24, msg=16, depth=3)
    node = ng.NeuralODE(ng.VMHConv(phi, gam, initialgraph=g), solver="euler", n_steps=2, dt=0.05)
    ps0, st = ng.setup(5, node)
    ps = prep(ps0, 5)
    u0 = torch.zeros(1, nv, device=DEV)
    with torch.no_grad():
        out, _ = node(u0, ps, st)
        assert any(k[0] == "vmh" for k in node._plans) and out.shape == (1, nv)
        for n_bad in (nv - 32, nv + 32):                                  # not the node count of the graph in st
            with pytest.raises(_lib.DimensionMismatch):
                node(torch.zeros(1, n_bad, device=DEV), ps, st)
        copy = lambda: {k: {l: dict(v) for l, v in sub.items()} for k, sub in ps.items()}
        bad = copy()
        bad["ϕ"]["layer_2"]["weight"] = torch.zeros(24, 23, device=DEV)   # layer_1 gives 24 rows, this one takes 23
        with pytest.raises(_lib.DimensionMismatch):
            node(u0, bad, st)
        bad = copy()
        bad["γ"]["layer_1"]["bias"] = torch.zeros(23, 1, device=DEV)
        with pytest.raises(_lib.DimensionMismatch):
            node(u0, bad, st)
        bad = copy()
        bad["γ"]["layer_1"]["weight"] = torch.zeros(24, 16, device=DEV)   # gamma takes [h_i; m_i] = 1 + 16 inputs
        with pytest.raises(_lib.DimensionMismatch):
            node(u0, bad, st)
        out2, _ = node(u0, ps, st)                                        # ... and the plan is still usable
    assert torch.equal(out, out2)


def test_vmh_tile_rounds_equal_the_generic_solver_and_can_be_switched_off(monkeypatch):
    # 9 000 points, 6 neighbours: 564 half tiles on 256 compute units -- three turns per workgroup and phase, the adjoint's second half
    # of a phase fused in front of the next phase's first.  Against the generic solver (which NGPDE_NO_VMH_ROUNDS=1 selects for such a
    # graph), repeated solves bitwise equal, and the forced abort poisons every half tile's rows
    nv, steps = 9000, 3
    pts = torch.as_tensor(S.uniform01(51, 2 * nv).reshape(2, nv).astype(np.float32), device=DEV)
    gv = ng.GNNGraph(ng.knn_graph(pts, 6), ndata={"x": pts})
    phi, gam = tutorial_mlps()
    u0 = torch.as_tensor(S.normal(52, nv).reshape(1, nv).astype(np.float32), device=DEV)
    R = torch.as_tensor(S.normal(53, nv).reshape(1, nv).astype(np.float32), device=DEV)
    res = {}
    monkeypatch.delenv("NGPDE_NO_VMH_NODE", raising=False)
    for mode in ("rounds", "generic", "rounds2"):
        if mode == "generic":
            monkeypatch.setenv("NGPDE_NO_VMH_ROUNDS", "1")
        else:
            monkeypatch.delenv("NGPDE_NO_VMH_ROUNDS", raising=False)
        node = ng.NeuralODE(ng.VMHConv(phi, gam, initialgraph=gv), solver="tsit5", n_steps=steps, dt=0.05)
        ps0, st = ng.setup(4, node)
        ps = prep(ps0, 4)
        u = u0.clone().requires_grad_(True)
        uT, _ = node(u, ps, st)
        (uT * R).sum().backward()
        assert ("vmh" in plan_flags(node)) == (mode != "generic")
        n1, _ = mlp_grad_pairs(ps["ϕ"], [{"weight": 0, "bias": 0}] * 4, phi)
        n2, _ = mlp_grad_pairs(ps["γ"], [{"weight": 0, "bias": 0}] * 4, gam)
        res[mode] = [uT.detach().clone(), u.grad.clone()] + [p.grad.clone() for _, p in n1 + n2]
    for a, b in zip(res["rounds"], res["generic"]):
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()) + 1e-6
    for a, b in zip(res["rounds"], res["rounds2"]):
        assert torch.equal(a, b)
    monkeypatch.setenv("NGPDE_DEBUG_FORCE_ABORT", "1")
    out, _ = node(u0, ng.to_device(ps0, DEV), st)
    monkeypatch.delenv("NGPDE_DEBUG_FORCE_ABORT")
    torch.cuda.synchronize()
    assert torch.isnan(out).all() and any(p.fault() for pool in node._plans.values() for p in pool)


def test_vmh_batch_of_clouds_that_share_tiles_runs_padded(monkeypatch):
    # four clouds of 3 000 points (VMH.md:120-134 batches 24 of them): 3 000 is not a multiple of the 32-row tile, a tile at a cloud
    # boundary stages two neighbourhoods and overflows its halo, and the union handle loses the persistent forms.  NeuralODE then runs the
    # plan on the same batch with every cloud padded to whole tiles by isolated nodes; values, du0 and parameter gradients against the
    # generic solver on the unpadded batch, with saveat
    monkeypatch.delenv("NGPDE_NO_VMH_NODE", raising=False)
    monkeypatch.delenv("NGPDE_NO_VMH_ROUNDS", raising=False)
    nv, nb, steps = 3000, 4, 2
    clouds = []
    for kb in range(nb):
        pk = torch.as_tensor(S.uniform01(200 + kb, 2 * nv).reshape(2, nv).astype(np.float32), device=DEV)
        clouds.append(ng.GNNGraph(ng.knn_graph(pk, 6), ndata={"x": pk}))
    gb = ng.batch(clouds)
    lib = _lib.load()
    ia = lambda v: (C.c_int32 * len(v))(*v)
    acts = [_lib.ACT["tanh"]] * 3 + [_lib.ACT["identity"]]
    if lib.ngpde_node_vmh_supported(gb.handle().ptr, 1, 2, 4, ia([4, 60, 60, 60, 40]), ia(acts), 4, ia([41, 60, 60, 60, 1]), ia(acts), _lib.AGGR["mean"]):
        pytest.skip("this batch's boundary tiles happen to fit their halos: nothing to pad")
    phi, gam = tutorial_mlps()
    N = nb * nv
    u0 = torch.as_tensor(S.normal(62, N).reshape(1, N).astype(np.float32), device=DEV)
    R = torch.as_tensor(S.normal(63, N * (steps + 1)).reshape(1, N, steps + 1).astype(np.float32), device=DEV)
    res = {}
    for mode in ("padded", "generic"):
        if mode == "generic":
            monkeypatch.setenv("NGPDE_NO_VMH_NODE", "1")
        node = ng.NeuralODE(ng.VMHConv(phi, gam, initialgraph=gb), solver="tsit5", n_steps=steps, dt=0.05, saveat=0.05)
        ps0, st = ng.setup(4, node)
        ps = prep(ps0, 4)
        u = u0.clone().requires_grad_(True)
        us, _ = node(u, ps, st)
        assert tuple(us.shape) == (1, N, steps + 1)
        (us * R).sum().backward()
        assert ("vmh" in plan_flags(node)) == (mode == "padded")
        n1, _ = mlp_grad_pairs(ps["ϕ"], [{"weight": 0, "bias": 0}] * 4, phi)
        n2, _ = mlp_grad_pairs(ps["γ"], [{"weight": 0, "bias": 0}] * 4, gam)
        res[mode] = [us.detach().clone(), u.grad.clone()] + [p.grad.clone() for _, p in n1 + n2]
    for a, b in zip(res["padded"], res["generic"]):
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()) + 1e-6


@pytest.mark.parametrize("sizes", [(700, 650, 900), (640, 960, 640)])
def test_vmh_reshuffled_batch_runs_on_the_first_batch_s_plan(sizes, monkeypatch):
    # DataLoader(shuffle = true) (VMH.md:120): the same clouds in a new order every epoch.  The second order is solved on the first
    # batch's graph and plan, the state's rows going through the permutation: ONE plan afterwards, the reshuffled solve equal bit for
    # bit to the first one's rows permuted (independent trajectories, the same tiles), and equal within rounding to a plan built for
    # the reshuffled batch itself (NGPDE_NO_BATCH_REUSE=1).  Sizes with and without padding to whole tiles; a cloud that occurs twice
    monkeypatch.delenv("NGPDE_NO_VMH_NODE", raising=False)
    monkeypatch.delenv("NGPDE_NO_BATCH_REUSE", raising=False)
    from ngpde_amd import batches as node_mod
    node_mod._CANON_BATCHES.clear()
    clouds = []
    for kb, nv in enumerate(sizes):
        pk = torch.as_tensor(S.uniform01(300 + kb, 2 * nv).reshape(2, nv).astype(np.float32), device=DEV)
        clouds.append(ng.GNNGraph(ng.knn_graph(pk, 6), ndata={"x": pk}))
    clouds.append(clouds[0])                                    # the first cloud again: four members, two of them one object
    phi, gam = tutorial_mlps(width=24, msg=12, depth=3)
    node = ng.NeuralODE(ng.VMHConv(phi, gam), solver="tsit5", n_steps=2, dt=0.05, saveat=0.05)
    ps0, st = ng.setup(4, node)
    member_u = [torch.as_tensor(S.normal(310 + k, c.num_nodes).astype(np.float32), device=DEV) for k, c in enumerate(clouds)]

    def solve(order, reuse=True):
        if reuse:
            monkeypatch.delenv("NGPDE_NO_BATCH_REUSE", raising=False)
        else:
            monkeypatch.setenv("NGPDE_NO_BATCH_REUSE", "1")
        gb = ng.batch([clouds[k] for k in order])
        st2 = ng.updategraph(st, gb)
        ps = prep(ps0, 4)
        u = torch.cat([member_u[k] for k in order]).reshape(1, -1).clone().requires_grad_(True)
        out, _ = node(u, ps, st2)
        (out * out).sum().backward()
        assert "vmh" in plan_flags(node)
        offs = np.concatenate([[0], np.cumsum([clouds[k].num_nodes for k in order])])
        n1, _ = mlp_grad_pairs(ps["ϕ"], [{"weight": 0, "bias": 0}] * 3, phi)
        n2, _ = mlp_grad_pairs(ps["γ"], [{"weight": 0, "bias": 0}] * 3, gam)
        per_member = {k: (out[:, offs[j]:offs[j + 1]].detach().clone(), u.grad[:, offs[j]:offs[j + 1]].clone()) for j, k in enumerate(order)}
        return per_member, [p_.grad.clone() for _, p_ in n1 + n2]

    first, gfirst = solve([0, 1, 2, 3])
    n_plans = sum(len(pool) for pool in node._plans.values())
    second, gsecond = solve([2, 3, 1, 0])
    assert sum(len(pool) for pool in node._plans.values()) == n_plans == 1      # no new plan for the new order
    for k in range(len(clouds)):
        assert torch.equal(first[k][0], second[k][0]) and torch.equal(first[k][1], second[k][1])
    for a, b in zip(gfirst, gsecond):
        assert torch.equal(a, b)
    own, gown = solve([2, 3, 1, 0], reuse=False)                               # a plan of the reshuffled batch itself
    for k in range(len(clouds)):
        for a, b in zip(second[k], own[k]):
            assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()) + 1e-6
    for a, b in zip(gsecond, gown):
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()) + 1e-6


def test_vmh_tapes_of_a_destroyed_plan_are_parked_and_can_be_released(monkeypatch):
    # a training loop that re-batches every epoch builds a plan per step (VMH.md:120-141): the tapes of a plan that went away are re-used
    # by the next one instead of going through hipFree / hipMalloc; ng.release_cached_memory() gives them back
    monkeypatch.delenv("NGPDE_NO_VMH_NODE", raising=False)
    import gc
    ng.release_cached_memory()
    nv = 3000
    phi, gam = tutorial_mlps()
    outs = []
    for rep in range(2):      # two graphs of one size: the second plan takes the first one's tapes
        pts = torch.as_tensor(S.uniform01(70 + rep, 2 * nv).reshape(2, nv).astype(np.float32), device=DEV)
        g = ng.GNNGraph(ng.knn_graph(pts, 6), ndata={"x": pts})
        node = ng.NeuralODE(ng.VMHConv(phi, gam, initialgraph=g), solver="tsit5", n_steps=4, dt=0.05)
        ps0, st = ng.setup(4, node)
        ps = prep(ps0, 4)
        u = torch.as_tensor(S.normal(71, nv).reshape(1, nv).astype(np.float32), device=DEV).requires_grad_(True)
        uT, _ = node(u, ps, st)
        uT.sum().backward()
        assert "vmh" in plan_flags(node) and torch.isfinite(u.grad).all()
        outs.append(uT.detach().clone())
        del node, uT
        gc.collect()
        torch.cuda.synchronize()
    released = ng.release_cached_memory()
    assert released >= 2 * 256 * 24 * (4 * 18000 + 4 * 3000)      # at least one plan's four tapes (24 evaluations) were parked
    assert ng.release_cached_memory() == 0


def test_vmh_solve_on_parked_tapes_of_another_graph_is_bitwise_the_solve_on_fresh_tapes(monkeypatch):
    # parked tapes are handed on unzeroed (node.hip: tape_pool_*): what the next plan reads of them must all have been written by its own
    # solve.  A wide, dense first model (k = 8: every edge row of a round is real; 60-wide layers) leaves its rows behind; the second
    # one -- variable degrees, 20 / 12-wide layers, fewer nodes -- runs on them, then again on freshly zeroed tapes: every output equal
    monkeypatch.delenv("NGPDE_NO_VMH_NODE", raising=False)
    import gc

    def first():
        nv = 3000
        pts = torch.as_tensor(S.uniform01(80, 2 * nv).reshape(2, nv).astype(np.float32), device=DEV)
        g = ng.GNNGraph(ng.knn_graph(pts, 8), ndata={"x": pts})
        phi, gam = tutorial_mlps()
        node = ng.NeuralODE(ng.VMHConv(phi, gam, aggr="+", initialgraph=g), solver="tsit5", n_steps=2, dt=0.05)
        ps0, st = ng.setup(5, node)
        ps = prep(ps0, 5)
        u = torch.as_tensor(S.normal(81, nv).reshape(1, nv).astype(np.float32), device=DEV).requires_grad_(True)
        uT, _ = node(u, ps, st)
        (uT * uT).sum().backward()
        assert "vmh" in plan_flags(node)

    def second():
        nv = 2800
        g, _ = spatial(nv, 82)
        phi, gam = tutorial_mlps(width=20, msg=12, depth=3)
        node = ng.NeuralODE(ng.VMHConv(phi, gam, initialgraph=g), solver="tsit5", n_steps=2, dt=0.05)
        ps0, st = ng.setup(6, node)
        ps = prep(ps0, 6)
        u = torch.as_tensor(S.normal(83, nv).reshape(1, nv).astype(np.float32), device=DEV).requires_grad_(True)
        uT, _ = node(u, ps, st)
        (uT * uT).sum().backward()
        assert "vmh" in plan_flags(node)
        n1, _ = mlp_grad_pairs(ps["ϕ"], [{"weight": 0, "bias": 0}] * 3, phi)
        n2, _ = mlp_grad_pairs(ps["γ"], [{"weight": 0, "bias": 0}] * 3, gam)
        return [uT.detach().clone(), u.grad.clone()] + [p_.grad.clone() for _, p_ in n1 + n2]

    ng.release_cached_memory()
    first()
    gc.collect()
    torch.cuda.synchronize()
    on_parked = second()
    gc.collect()
    torch.cuda.synchronize()
    assert ng.release_cached_memory() > 0
    on_fresh = second()
    for a, b in zip(on_parked, on_fresh):
        assert torch.isfinite(a).all() and torch.equal(a, b)


def test_vmh_resident_plan_reports_an_abort_instead_of_hanging(monkeypatch):
    # a launch whose waits give up (forced: NGPDE_DEBUG_FORCE_ABORT=1 starts it with the abort word set) writes NaN outputs and
    # latches the plan's fault word; the next entry of the plan fails instead of computing on garbage; a fresh plan works
    monkeypatch.delenv("NGPDE_NO_VMH_NODE", raising=False)
    N = 700
    g, _ = spatial(N, 12)
    phi, gam = tutorial_mlps(depth=3)
    node = ng.NeuralODE(ng.VMHConv(phi, gam, initialgraph=g), solver="tsit5", n_steps=2, dt=0.05)
    ps, st = ng.setup(1, node)
    ps = ng.to_device(ps, DEV)
    u = torch.randn(1, N, device=DEV)
    out, _ = node(u, ps, st)
    assert torch.isfinite(out).all() and "vmh" in plan_flags(node)
    monkeypatch.setenv("NGPDE_DEBUG_FORCE_ABORT", "1")
    out, _ = node(u, ps, st)
    monkeypatch.delenv("NGPDE_DEBUG_FORCE_ABORT")
    torch.cuda.synchronize()
    assert torch.isnan(out).all()
    assert any(p.fault() for pool in node._plans.values() for p in pool)
    with pytest.raises(_lib.NgpdeError, match="gave up waiting"):
        node(u, ps, st)
    fresh = ng.NeuralODE(ng.VMHConv(phi, gam, initialgraph=g), solver="tsit5", n_steps=2, dt=0.05)
    out, _ = fresh(u, ps, st)
    assert torch.isfinite(out).all()
