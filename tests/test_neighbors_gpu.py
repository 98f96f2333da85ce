"""Neighbour search on the device (ngpde_radius_graph / ngpde_knn_graph / ngpde_spatial_order) against the brute-force
float32 restatement in oracle/ngpde_oracle.py.  Index work: the bar is bit-exact -- the same edge list, in the same
canonical order.  Reference boundary: GNNGraphs.radius_graph / knn_graph, re-exported at
/root/reference/src/NeuralGraphPDE.jl:4 ([UPSTREAM] GraphNeuralNetworks.jl over NearestNeighbors.jl trees)."""
import ctypes as C

import numpy as np
import pytest
import torch

import ngpde_amd as ng
from ngpde_amd import _lib
from ngpde_amd import synth as S
from oracle import ngpde_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def points(seed, dim, n, kind="uniform"):
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        return rng.random((dim, n)).astype(np.float32)
    if kind == "clustered":      # a few dense blobs and a sparse background: very uneven cells
        c = rng.random((dim, 6))
        P = c[:, rng.integers(0, 6, n)] + 0.01 * rng.standard_normal((dim, n))
        P[:, : n // 8] = rng.random((dim, n // 8)) * 4 - 2
        return P.astype(np.float32)
    if kind == "lattice":        # many pairs exactly at the threshold distance: the float rule decides
        side = int(round(n ** (1.0 / dim)))
        ax = [np.arange(side, dtype=np.float32) * np.float32(0.125)] * dim
        return np.stack([a.reshape(-1) for a in np.meshgrid(*ax, indexing="ij")]).astype(np.float32)
    raise ValueError(kind)


def edges(g):
    return g._s0.astype(np.int64), g._t0.astype(np.int64)


@pytest.mark.parametrize("dim", [1, 2, 3])
@pytest.mark.parametrize("kind", ["uniform", "clustered", "lattice"])
@pytest.mark.parametrize("self_loops", [False, True])
def test_radius_graph_matches_oracle(dim, kind, self_loops):
    n = {1: 3000, 2: 4096, 3: 3375}[dim]
    P = points(dim * 7 + 1, dim, n, kind)
    n = P.shape[1]
    r = {"uniform": {1: 0.002, 2: 0.03, 3: 0.09}[dim], "clustered": 0.02, "lattice": 0.125}[kind]
    g = ng.radius_graph(P, r, self_loops=self_loops)
    s, t = edges(g)
    so, to = O.radius_graph(P, r, self_loops=self_loops)
    assert g.num_nodes == n and g.num_edges == so.size
    assert np.array_equal(s, so) and np.array_equal(t, to)
    assert so.size > n // 2                                            # the case is not trivially empty


def test_radius_graph_threshold_is_inclusive_and_float():
    # neighbours at exactly r are kept (d2 <= r*r in float32); one ulp further they are dropped
    r = np.float32(0.3)
    up = np.nextafter(r, np.float32(1))
    assert up * up > r * r                                               # distinguishable after rounding to float32
    x = np.array([0.0, r, -up], dtype=np.float32).reshape(1, -1)
    g = ng.radius_graph(x, float(r))
    s, t = edges(g)
    so, to = O.radius_graph(x, float(r))
    assert np.array_equal(s, so) and np.array_equal(t, to)
    assert set(zip(s.tolist(), t.tolist())) == {(0, 1), (1, 0)}


def test_radius_graph_direction_batches_and_symmetry():
    P = points(3, 2, 3000)
    gi = np.repeat(np.arange(1, 4), 1000)                                # three graphs, 1-based ids as in GNNGraphs
    g_in = ng.radius_graph(P, 0.05, graph_indicator=gi)
    g_out = ng.radius_graph(P, 0.05, graph_indicator=gi, dir="out")
    s, t = edges(g_in)
    so, to = O.radius_graph(P, 0.05, graph_indicator=gi)
    assert np.array_equal(s, so) and np.array_equal(t, to)
    assert g_in.num_graphs == 3
    assert np.array_equal(g_out._s0, t) and np.array_equal(g_out._t0, s)
    assert np.array_equal(gi[s], gi[t])                                  # no edge crosses graphs
    fwd = set(zip(s.tolist(), t.tolist()))
    assert all((b, a) in fwd for a, b in fwd)                            # (a-b)^2 == (b-a)^2 exactly: symmetric
    # a batch equals its members searched one by one (MLUtils.batch of the three graphs)
    parts = [O.radius_graph(P[:, k * 1000:(k + 1) * 1000], 0.05) for k in range(3)]
    assert np.array_equal(s, np.concatenate([p[0] + 1000 * k for k, p in enumerate(parts)]))


def test_radius_graph_edge_cases():
    assert ng.radius_graph(np.zeros((2, 0), np.float32), 0.1).num_edges == 0
    assert ng.radius_graph(np.zeros((2, 1), np.float32), 0.1).num_edges == 0
    assert ng.radius_graph(np.zeros((2, 1), np.float32), 0.1, self_loops=True).num_edges == 1
    same = np.ones((3, 40), np.float32)                                  # coincident points: complete graph, r = 0 included
    g = ng.radius_graph(same, 0.0)
    assert g.num_edges == 40 * 39
    so, to = O.radius_graph(same, 0.0)
    assert np.array_equal(g._s0, so) and np.array_equal(g._t0, to)
    P = points(5, 2, 300)
    g = ng.radius_graph(P, float("inf"))                                 # everything is a neighbour
    assert g.num_edges == 300 * 299
    g = ng.radius_graph(P * 1e4 - 5e3, 1e-3)                             # radius far below the cell cap: grid coarsened
    assert g.num_edges == O.radius_graph(P * 1e4 - 5e3, 1e-3)[0].size
    with pytest.raises(ng.ArgumentError):
        ng.radius_graph(P, -1.0)
    with pytest.raises(ng.ArgumentError):
        ng.radius_graph(np.array([[0.0, np.nan]], np.float32), 0.1)
    with pytest.raises(ng.ArgumentError):
        ng.radius_graph(P, 0.1, graph_indicator=np.zeros(300, np.int32))  # ids are 1-based
    with pytest.raises(ng.NgpdeError):
        ng.radius_graph(np.zeros((4, 10), np.float32), 0.1)              # more than three coordinates
    with pytest.raises(ng.DimensionMismatch):
        ng.radius_graph(P, 0.1, graph_indicator=np.ones(7, np.int32))


def test_radius_graph_capacity_is_checked():
    lib = _lib.load()
    P = torch.as_tensor(points(1, 2, 500).T.copy(), device=DEV)
    ne = C.c_int64()
    _lib.check(lib.ngpde_radius_graph(500, 2, P.data_ptr(), 0.2, None, 1, 0, 0, 0, 0, 0, None, None, C.byref(ne), None))
    m = ne.value
    assert m > 0
    s = torch.full((m,), -7, dtype=torch.int32, device=DEV)
    t = torch.full((m,), -7, dtype=torch.int32, device=DEV)
    st = lib.ngpde_radius_graph(500, 2, P.data_ptr(), 0.2, None, 1, 0, 0, 0, 0, m - 1, s.data_ptr(), t.data_ptr(), C.byref(ne), None)
    assert st == _lib.ERR_INVALID_ARGUMENT and b"output arrays hold" in lib.ngpde_last_error()
    assert int((s != -7).sum()) == 0                                      # nothing was written
    _lib.check(lib.ngpde_radius_graph(500, 2, P.data_ptr(), 0.2, None, 1, 0, 0, 0, 1, m, s.data_ptr(), t.data_ptr(), C.byref(ne), None))
    so, to = O.radius_graph(P.cpu().numpy().T, 0.2)
    assert np.array_equal(s.cpu().numpy(), so + 1) and np.array_equal(t.cpu().numpy(), to + 1)   # index_base = 1


@pytest.mark.parametrize("dim", [1, 2, 3])
@pytest.mark.parametrize("kind", ["uniform", "clustered", "lattice"])
@pytest.mark.parametrize("k,self_loops", [(1, False), (6, False), (6, True), (33, False)])
def test_knn_graph_matches_oracle(dim, kind, k, self_loops):
    n = {1: 2000, 2: 2500, 3: 2197}[dim]
    P = points(dim * 11 + k, dim, n, kind)
    n = P.shape[1]
    g = ng.knn_graph(P, k, self_loops=self_loops)
    s, t = edges(g)
    so, to = O.knn_graph(P, k, self_loops=self_loops)
    assert g.num_edges == n * k
    assert np.array_equal(t, to)
    assert np.array_equal(s, so)


def test_knn_graph_batches_direction_and_limits():
    P = points(9, 2, 1200)
    gi = np.repeat(np.arange(1, 5), 300)
    g = ng.knn_graph(P, 8, graph_indicator=gi)
    s, t = edges(g)
    so, to = O.knn_graph(P, 8, graph_indicator=gi)
    assert np.array_equal(s, so) and np.array_equal(t, to) and g.num_graphs == 4
    g_out = ng.knn_graph(P, 8, graph_indicator=gi, dir="out")
    assert np.array_equal(g_out._s0, t) and np.array_equal(g_out._t0, s)
    assert np.bincount(t, minlength=1200).tolist() == [8] * 1200           # in-degree exactly k
    # k = 128 (the LDS list at its limit) and k = 0
    P2 = points(10, 3, 400)
    g = ng.knn_graph(P2, 128)
    assert np.array_equal(g._s0, O.knn_graph(P2, 128)[0])
    assert ng.knn_graph(P2, 0).num_edges == 0
    with pytest.raises(ng.NgpdeError):
        ng.knn_graph(P2, 129)
    with pytest.raises(ng.ArgumentError):
        ng.knn_graph(P2[:, :5], 5)                                            # needs k + 1 points without self loops
    assert ng.knn_graph(P2[:, :5], 5, self_loops=True).num_edges == 25
    with pytest.raises(ng.ArgumentError):                                     # one of the graphs is too small
        ng.knn_graph(P2[:, :30], 12, graph_indicator=np.r_[np.ones(20, np.int32), np.full(10, 2, np.int32)])


@pytest.mark.parametrize("dim", [1, 2, 3])
def test_spatial_order_matches_oracle(dim):
    n = 5000
    P = points(20 + dim, dim, n, "clustered")
    gi = np.sort(np.random.default_rng(4).integers(1, 4, n)).astype(np.int32)
    pts = torch.as_tensor(P.T.copy(), device=DEV)
    out = torch.empty(n, dtype=torch.int32, device=DEV)
    lib = _lib.load()
    for ind in (None, gi):
        gd = None if ind is None else torch.as_tensor(ind, device=DEV)
        _lib.check(lib.ngpde_spatial_order(n, dim, pts.data_ptr(), _lib.ptr(gd), 1 if ind is None else 3, 1, out.data_ptr(), None))
        assert np.array_equal(out.cpu().numpy(), O.spatial_order(P, ind, id_base=1))


def test_c2_graph_from_points_on_device():
    """BASELINE config 2 is a radius graph: the device search at full size against properties that need no O(n^2) oracle
    (symmetry, no self loops, every edge within r, the degree histogram of a cKDTree count in float64 away from the
    threshold), and the spatial schedule drives the GCN layer to the same result as the BFS schedule."""
    from scipy.spatial import cKDTree
    pts, s_ref, t_ref = S.closest_pairs_graph(16384, 65536, 2)
    P = pts.T.astype(np.float32)
    d_ref = np.sqrt(((pts[s_ref] - pts[t_ref]) ** 2).sum(axis=1))
    r = float(np.float32(d_ref.max()) * np.float32(1.0000005))               # just above the 65536-th closest pair
    g = ng.radius_graph(P, r)
    s, t = edges(g)
    assert np.array_equal(np.lexsort((s, t)), np.arange(s.size))             # canonical order: by target, sources ascending
    assert not np.any(s == t)
    key = s * 16384 + t
    assert np.array_equal(np.sort(key), np.sort(t * 16384 + s))              # symmetric
    P64 = P.astype(np.float64).T
    d = np.sqrt(((P64[s] - P64[t]) ** 2).sum(axis=1))
    assert d.max() <= r * (1 + 1e-6)
    tree = cKDTree(P64)
    lo = tree.query_ball_point(P64, r * (1 - 1e-5), return_length=True) - 1
    hi = tree.query_ball_point(P64, r * (1 + 1e-5), return_length=True) - 1
    deg = np.bincount(t, minlength=16384)
    assert np.all(deg >= lo) and np.all(deg <= hi)
    assert set(zip(s_ref.tolist(), t_ref.tolist())) <= set(zip(s.tolist(), t.tolist()))   # holds the closest-pairs graph

    # same layer output under the BFS and the spatial schedule (the order only changes which workgroup owns which rows)
    g2 = ng.radius_graph(P, r, locality="spatial")
    assert np.array_equal(g2._s0, g._s0)
    assert not np.array_equal(g2.node_order(), g.node_order())
    x = torch.as_tensor(S.normal(5, 64 * 16384).reshape(16384, 64).T.astype(np.float32).copy(), device=DEV)
    ys = []
    for gg in (g, g2):
        layer = ng.GCNConv((64, 64), "relu", initialgraph=gg)
        ps, st = ng.setup(0, layer)
        ps = ng.to_device(ps, DEV)
        y, _ = layer(x, ps, st)
        ys.append(y.cpu().numpy())
    assert np.allclose(ys[0], ys[1], rtol=1e-5, atol=1e-6)


def test_order_must_be_a_permutation():
    lib = _lib.load()
    s = torch.as_tensor([0, 1, 2], dtype=torch.int32, device=DEV)
    t = torch.as_tensor([1, 2, 0], dtype=torch.int32, device=DEV)
    out = C.c_void_p()
    for bad in ([0, 0, 1], [0, 1, 3], [-1, 1, 2]):
        o = torch.as_tensor(bad, dtype=torch.int32, device=DEV)
        st = lib.ngpde_graph_create_device(3, 3, s.data_ptr(), t.data_ptr(), 32, 0, 1, o.data_ptr(), None, C.byref(out))
        assert st == _lib.ERR_INVALID_ARGUMENT and b"permutation" in lib.ngpde_last_error()
