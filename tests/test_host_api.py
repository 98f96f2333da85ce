"""Host-side mirror of the reference's Lux API: state structure, parameter structure, updategraph
identity and error behaviour, as asserted by /root/reference/test/runtests.jl (CPU; no kernels run)."""
import numpy as np
import pytest
import torch

import ngpde_amd as ng


def fixture_graph(**kw):
    return ng.GNNGraph([1, 1, 2, 3], [2, 3, 1, 1], **kw)      # test/runtests.jl:11-13


def test_gcn_setup_state_and_parameters():
    g = fixture_graph()
    l = ng.GCNConv((3, 5), initialgraph=g)
    ps, st = ng.setup(0, l)
    assert st == {"graph": g}                                  # :21
    assert list(ps) == ["weight", "bias"]
    assert tuple(ps["weight"].shape) == (5, 3) and tuple(ps["bias"].shape) == (5, 1)   # src/layers.jl:166-167
    assert ps["weight"].stride() == (1, 5)                     # Julia column-major memory order
    assert l.parameterlength() == 20 and l.statelength() == 1  # src/layers.jl:173-175, :24
    assert ng.GCNConv((3, 5), bias=False).parameterlength() == 15
    assert repr(l) == "GCNConv(3 => 5)" and repr(ng.GCNConv((3, 5), "relu")) == "GCNConv(3 => 5, relu)"
    # positional ctor defaults to glorot_normal, pair ctor to glorot_uniform (src/layers.jl:178 vs :193)
    assert ng.GCNConv(3, 5).init_weight is ng.glorot_normal and l.init_weight is ng.glorot_uniform
    lim = np.sqrt(6 / 8)
    assert float(ps["weight"].abs().max()) <= lim + 1e-6


def test_default_initialgraph_is_emptygraph():
    l = ng.GCNConv((3, 5))
    _, st = ng.setup(0, l)
    assert st["graph"].num_nodes == 0 and st["graph"].num_edges == 0   # src/layers.jl:14,21


def test_initialgraph_accepts_graph_or_thunk():
    g = fixture_graph()
    assert ng.setup(0, ng.GCNConv((3, 5), initialgraph=g))[1]["graph"] == g
    assert ng.setup(0, ng.GCNConv((3, 5), initialgraph=lambda: g))[1]["graph"] is g    # src/utils.jl:16-17
    with pytest.raises(TypeError):
        ng.GCNConv((3, 5), initialgraph=3)


def test_chain_states_and_updategraph_identity():
    # test/runtests.jl:167-185
    g = ng.rand_graph(5, 4, bidirected=False, seed=0)
    l = ng.GCNConv((3, 5), initialgraph=g)
    ps, st = ng.setup(0, l)
    new_g = ng.rand_graph(5, 7, bidirected=False, seed=1)
    new_st = ng.updategraph(st, new_g)
    assert new_st["graph"] is new_g
    model = ng.Chain(ng.GCNConv((3, 5), initialgraph=g), ng.GCNConv((5, 5), initialgraph=g))
    ps, st = ng.setup(0, model)
    assert list(ps) == ["layer_1", "layer_2"] and list(st) == ["layer_1", "layer_2"]
    new_st = ng.updategraph(st, new_g)
    assert new_st["layer_1"]["graph"] is new_st["layer_2"]["graph"] is new_g
    assert ng.updategraph({}, new_g) == {}                     # src/utils.jl:25


def test_updategraph_with_graph_data():
    # test/runtests.jl:188-205
    g = ng.rand_graph(5, 4, bidirected=False, seed=0)
    l = ng.GCNConv((3, 5), initialgraph=g)
    _, st = ng.setup(0, l)
    ndata = np.random.rand(3, g.num_nodes)
    new_st = ng.updategraph(st, ndata=ndata)
    assert new_st["graph"].ndata["x"] is ndata
    model = ng.Chain(ng.GCNConv((3, 5), initialgraph=g), ng.GCNConv((5, 5), initialgraph=g))
    _, st = ng.setup(0, model)
    new_st = ng.updategraph(st, ndata=ndata)
    assert new_st["layer_1"]["graph"].ndata["x"] is new_st["layer_2"]["graph"].ndata["x"] is ndata


def test_graph_data_validation_and_batch():
    g = fixture_graph()
    with pytest.raises(ng.DimensionMismatch):
        ng.GNNGraph(g, ndata=np.zeros((2, 5)))                 # wrong number of node columns
    with pytest.raises(ng.DimensionMismatch):
        ng.GNNGraph([1, 2], [2, 9], num_nodes=3)               # edge outside 1:3
    gh = ng.GNNGraph(g, ndata={"u": np.random.rand(2, 3), "x": np.random.rand(3, 3)}, gdata={"θ": np.random.rand(4)})
    gb = ng.batch([gh, gh.copy()])                             # test/runtests.jl:92
    assert gb.num_nodes == 6 and gb.num_edges == 8 and gb.num_graphs == 2
    assert tuple(gb.ndata["u"].shape) == (2, 6) and tuple(gb.gdata["θ"].shape) == (4, 2)
    s, t = gb.edge_index()
    assert s.tolist() == [1, 1, 2, 3, 4, 4, 5, 6] and t.tolist() == [2, 3, 1, 1, 5, 6, 4, 4]
    assert gh.copy() == gh and gh.copy() is not gh


def test_dimension_errors_raise_before_any_kernel():
    g = fixture_graph()
    l = ng.GCNConv((3, 5), initialgraph=g)
    ps, st = ng.setup(0, l)
    with pytest.raises(ng.DimensionMismatch):
        l(torch.zeros(4, 3), ps, st)                           # wrong feature count
    with pytest.raises(ng.DimensionMismatch):
        l(torch.zeros(3, 7), ps, st)                           # wrong node count
    with pytest.raises(ng.ArgumentError, match="expected 4 but given 2"):
        l(torch.zeros(3, 3), ps, st, torch.ones(2))            # src/layers.jl:207
    with pytest.raises(ng.ArgumentError):
        ng.GCNConv((3, 5), "not_an_activation")


def test_neuralode_container_flattens_params_and_states():
    # graph_node.md:44-52: a single-field Lux container shares its model's ps and st
    g = fixture_graph()
    chain = ng.Chain(ng.GCNConv((16, 16), "relu", initialgraph=g), ng.GCNConv((16, 16), "relu", initialgraph=g))
    node = ng.NeuralODE(chain, solver="tsit5", tspan=(0.0, 1.0), n_steps=50)
    ps, st = ng.setup(0, node)
    assert list(ps) == ["layer_1", "layer_2"] and st["layer_1"]["graph"] == g
    assert abs(node.dt - 0.02) < 1e-12
    with pytest.raises(ng.ArgumentError):
        ng.NeuralODE(chain, solver="rk45")


def test_no_cpu_fallback_for_the_hot_path():
    g = fixture_graph()
    l = ng.GCNConv((3, 5), initialgraph=g)
    ps, st = ng.setup(0, l)
    if torch.cuda.is_available():
        pytest.skip("GPU box: covered by the gpu tests")
    with pytest.raises((ng.NgpdeError, RuntimeError)):
        l(torch.zeros(3, 3), ps, st)                           # CPU tensors are rejected, never silently computed


def test_neuralode_generic_path_host_helpers():
    # the parameter-tree helpers of the generic NeuralODE path (node.py): leaves in insertion order (ComponentArray order), rebuilt
    # trees keep the structure; a weight's cotangent that arrives transposed is combined through its dense view, not copied
    import torch
    from ngpde_amd.node import _dense, _leaves, _rebuild, NeuralODE, TABLEAUS
    tree = {"ϕ": {"layer_1": {"weight": torch.zeros(3, 2), "bias": torch.zeros(3, 1)}, "layer_2": {"weight": torch.ones(1, 3)}},
            "γ": {"weight": torch.full((2, 2), 2.0)}, "empty": {}}
    lv = _leaves(tree)
    assert [tuple(t.shape) for t in lv] == [(3, 2), (3, 1), (1, 3), (2, 2)]
    rb = _rebuild(tree, iter([t + 1 for t in lv]))
    assert list(rb) == ["ϕ", "γ", "empty"] and list(rb["ϕ"]["layer_1"]) == ["weight", "bias"] and rb["empty"] == {}
    assert torch.equal(rb["γ"]["weight"], torch.full((2, 2), 3.0))
    w = torch.arange(6.0).reshape(2, 3)
    assert _dense(w) is w and _dense(w.T).data_ptr() == w.data_ptr() and _dense(w.T).is_contiguous()
    assert _dense(w[:, ::2]).is_contiguous()                            # anything else: a contiguous copy
    # every row of a tableau sums to its node: consistency of the coefficients the combinations use
    a, b = TABLEAUS["tsit5"]
    assert abs(sum(b) - 1.0) < 1e-14 and len(a) == len(b) == 6
    node = NeuralODE(object(), solver="Tsit5", n_steps=7, tspan=(0.0, 2.0), capture=True)
    assert node.solver == "tsit5" and abs(node.dt - 2.0 / 7) < 1e-15 and node.capture


def test_reshuffled_batch_maps_onto_the_first_batch_of_the_same_members(monkeypatch):
    # batches.py: _canonical_batch -- DataLoader(shuffle = true) (VMH.md:120) hands the loop the same clouds in a new order; the solver
    # runs such a batch on the first batch's plan through this node map.  Host logic only: the map is a permutation under which the
    # first batch's node data ARE the reshuffled batch's, a cloud that occurs twice gets two different slots, other members start
    # their own entry, and the switch turns it off.
    from ngpde_amd import batches as node_mod
    monkeypatch.delenv("NGPDE_NO_BATCH_REUSE", raising=False)
    node_mod._CANON_BATCHES.clear()
    rng = np.random.default_rng(0)
    clouds = []
    for n in (5, 7, 4, 6):
        s, t = rng.integers(0, n, size=3 * n), rng.integers(0, n, size=3 * n)
        clouds.append(ng.GNNGraph(s, t, num_nodes=n, index_base=0, ndata={"x": torch.as_tensor(rng.random((2, n)).astype(np.float32))}))
    first = ng.batch([clouds[0], clouds[1], clouds[2], clouds[0]])
    assert node_mod._canonical_batch(first, "cpu") is None            # the first of its kind: remembered, solved as it is
    assert node_mod._canonical_batch(first, "cpu") is None            # ... and again itself
    again = ng.batch([clouds[2], clouds[0], clouds[1], clouds[0]])
    g0, nodemap = node_mod._canonical_batch(again, "cpu")
    assert g0 is first and sorted(nodemap.tolist()) == list(range(first.num_nodes))
    assert torch.equal(torch.as_tensor(again.ndata["x"]), torch.as_tensor(first.ndata["x"])[:, nodemap])
    assert node_mod._canonical_batch(again, "cpu")[1] is nodemap      # kept on the batch
    other = ng.batch([clouds[0], clouds[1], clouds[3], clouds[0]])    # another member: its own entry
    assert node_mod._canonical_batch(other, "cpu") is None
    assert node_mod._canonical_batch(clouds[0], "cpu") is None        # not a batch
    monkeypatch.setenv("NGPDE_NO_BATCH_REUSE", "1")
    assert node_mod._canonical_batch(ng.batch([clouds[1], clouds[0], clouds[2], clouds[0]]), "cpu") is None
    node_mod._CANON_BATCHES.clear()
