"""CPU tests of the oracle itself: the reference's known-answer test, tableau order conditions,
structural assertions of test/runtests.jl, and finite-difference checks of every hand-derived VJP."""
import numpy as np
import pytest

from oracle import ngpde_oracle as O
from ngpde_amd import synth as S

RNG = np.random.default_rng(0)


def fixture_graph(**kw):
    # test/runtests.jl:11-13
    return O.Graph([1, 1, 2, 3], [2, 3, 1, 1], **kw)


def rand_mlp(sizes, acts, rng=RNG):
    return [dict(weight=rng.normal(size=(o, i)) * 0.5, bias=rng.normal(size=(o, 1)) * 0.1, act=a)
            for i, o, a in zip(sizes[:-1], sizes[1:], acts)]


# ---- pins ------------------------------------------------------------------------------------

def test_spectralconv_known_answer_float32():
    # /root/reference/test/runtests.jl:153-162
    n = 100
    g = O.spectral_graph(n, np.float32)
    x = np.linspace(0, 2 * np.pi, n + 1, dtype=np.float32)[1:]
    e1 = O.spectral_conv(np.sin(x), g, n) - np.cos(x)
    e2 = O.spectral_conv(np.cos(x), g, n) + np.sin(x)
    assert e1.dtype == np.float32
    assert float((e1 ** 2).sum()) < 1e-3
    assert float((e2 ** 2).sum()) < 1e-3


def test_spectralconv_docstring_residuals_float64():
    # src/layers.jl:590-630: residuals ~1e-15 .. 4e-13
    n = 100
    g = O.spectral_graph(n)
    x = np.linspace(0, 2 * np.pi, n + 1)[1:]
    assert np.abs(O.spectral_conv(np.sin(x), g, n) - np.cos(x)).max() < 2e-12
    assert np.abs(O.spectral_conv(np.cos(x), g, n) + np.sin(x)).max() < 2e-12


def test_direction_convention_is_pinned():
    # flipping source/target must break the known-answer test
    n = 100
    g = O.spectral_graph(n)
    flipped = O.Graph(g.t, g.s, num_nodes=n, edata=g.edata["e"], index_base=0)
    x = np.linspace(0, 2 * np.pi, n + 1)[1:]
    assert ((O.spectral_conv(np.sin(x), flipped, n) - np.cos(x)) ** 2).sum() > 1.0


def test_tsit5_tableau_order_conditions():
    tb = O.TSIT5
    for ci, row in zip(tb["c"], tb["a"]):
        assert abs(sum(row) - ci) < 1e-14
    assert abs(sum(tb["b"]) - 1.0) < 1e-14
    rhs = lambda u: (u, None)
    errs = []
    for dt in (0.05, 0.1, 0.2):
        u, _ = O.rk_solve(rhs, np.ones(1), tb, dt, 1)
        errs.append(abs(u[0] - np.exp(dt)))
    # 5th order: local error = C6 dt^6 + C7 dt^7 with small C6 (Tsitouras minimised it):
    # measured 5.2e-13 / 2.3e-11 / 1.7e-10 ~ |4.3e-5 dt^6 - dt^7/5040|
    for dt, e in zip((0.05, 0.1, 0.2), errs):
        model = abs(4.3e-5 * dt ** 6 - dt ** 7 / 5040)
        assert e < 1e-4 * dt ** 6
        assert 0.5 * model < e < 2.0 * model


# ---- structural assertions of test/runtests.jl ------------------------------------------------

def test_shapes_like_reference_tests():
    g = fixture_graph()
    x = RNG.normal(size=(3, 3)).astype(np.float32)
    y, _ = O.gcn_conv(x, RNG.normal(size=(5, 3)).astype(np.float32), np.zeros((5, 1), np.float32), g)
    assert y.shape == (5, 3) and y.dtype == np.float32                       # :23
    gh = fixture_graph(ndata={"x": RNG.random((3, 3))})
    u = RNG.normal(size=(4, 3))
    y, _ = O.explicit_edge_conv(u, rand_mlp([11, 5], ["identity"]), gh)      # :36
    assert y.shape == (5, 3)
    y, _ = O.vmh_conv(u, rand_mlp([11, 5], ["identity"]), rand_mlp([9, 7], ["identity"]), gh)  # :53
    assert y.shape == (7, 3)
    gm = fixture_graph(ndata={"u": RNG.random((2, 3)), "x": RNG.random((3, 3))}, gdata={"θ": RNG.random(4)})
    h = RNG.normal(size=(5, 3))
    y, _ = O.mppde_conv(h, rand_mlp([19, 5], ["identity"]), rand_mlp([14, 7], ["identity"]), gm)  # :72
    assert y.shape == (7, 3)
    ge = fixture_graph(edata={"u": RNG.random((2, 4)), "x": RNG.random((3, 4))}, gdata={"θ": RNG.random(4)})
    y, _ = O.mppde_conv(h, rand_mlp([19, 5], ["identity"]), rand_mlp([14, 7], ["identity"]), ge)  # :86
    assert y.shape == (7, 3)
    gb = O.batch([gm, gm.copy()])                                            # :92
    y, _ = O.mppde_conv(RNG.normal(size=(5, 6)), rand_mlp([19, 5], ["identity"]),
                        rand_mlp([14, 7], ["identity"]), gb)
    assert y.shape == (7, 6)
    gn = fixture_graph(ndata={"u": RNG.random((2, 3)), "x": RNG.random((3, 3))})
    y, _ = O.mppde_conv(h, rand_mlp([15, 5], ["identity"]), rand_mlp([10, 7], ["identity"]), gn)  # :119
    assert y.shape == (7, 3)


def test_batched_graph_equals_per_graph():
    gm = fixture_graph(ndata={"u": RNG.random((2, 3)), "x": RNG.random((3, 3))}, gdata={"θ": RNG.random(4)})
    g2 = fixture_graph(ndata={"u": RNG.random((2, 3)), "x": RNG.random((3, 3))}, gdata={"θ": RNG.random(4)})
    phi, psi = rand_mlp([19, 8, 5], ["swish", "swish"]), rand_mlp([14, 7], ["identity"])
    h1, h2 = RNG.normal(size=(5, 3)), RNG.normal(size=(5, 3))
    yb, _ = O.mppde_conv(np.concatenate([h1, h2], 1), phi, psi, O.batch([gm, g2]))
    y1, _ = O.mppde_conv(h1, phi, psi, gm)
    y2, _ = O.mppde_conv(h2, phi, psi, g2)
    np.testing.assert_allclose(yb, np.concatenate([y1, y2], 1), rtol=1e-13, atol=1e-13)


def test_mean_of_empty_neighbourhood_is_zero():
    g = O.Graph([1, 2], [2, 1], num_nodes=4)   # nodes 3,4 isolated
    m = O.scatter("mean", np.ones((2, 2)), g.t, 4)
    assert np.all(m[:, 2:] == 0) and np.all(np.isfinite(m))


def test_product_scatter_pullback_with_zero_entries():
    # scatter(*): every entry receives dout times the product of the OTHER entries of its destination -- also where entries are exactly
    # zero (a relu message): one zero in a destination -> that entry gets the product of the rest, the rest get 0; two zeros -> all 0.
    # Against the definition, entry by entry, and against dout * out / M where nothing is zero.
    rng = np.random.default_rng(3)
    n, E, d = 7, 40, 3
    idx = rng.integers(0, n - 1, size=E)                 # destination n - 1 stays empty (neutral element 1)
    M = rng.normal(size=(d, E))
    M[:, rng.choice(E, size=9, replace=False)] = 0.0
    M[1, idx == idx[0]] = 0.0                            # a destination with several zeros in one feature
    out = O.scatter("*", M, idx, n)
    assert (out[:, n - 1] == 1.0).all()
    dout = rng.normal(size=(d, n))
    got = O.scatter_pullback("*", M, idx, n, out, dout)
    ref = np.zeros_like(M)
    for e in range(E):
        mates = [k for k in range(E) if idx[k] == idx[e] and k != e]
        ref[:, e] = dout[:, idx[e]] * np.prod(M[:, mates], axis=1)
    assert np.abs(got - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())
    M2 = np.where(M == 0, 0.7, M)
    out2 = O.scatter("*", M2, idx, n)
    assert np.allclose(O.scatter_pullback("*", M2, idx, n, out2, dout), O.gather(dout * out2, idx) / M2, rtol=1e-12, atol=1e-14)


def test_gcn_matches_dense_formula():
    # Y = act(W X C (A+I) C + b),  C = diag(1/sqrt(indeg+1))   
    N, D = 7, 4
    s = RNG.integers(1, N + 1, 15)
    t = RNG.integers(1, N + 1, 15)
    g = O.Graph(s, t, num_nodes=N)
    A = np.zeros((N, N))
    np.add.at(A, (s - 1, t - 1), 1.0)      # A[s, t]
    A += np.eye(N)
    c = 1 / np.sqrt(A.sum(0))
    X = RNG.normal(size=(D, N))
    W, b = RNG.normal(size=(5, D)), RNG.normal(size=(5, 1))
    y, _ = O.gcn_conv(X, W, b, g, "tanh")
    np.testing.assert_allclose(y, np.tanh(W @ (X * c) @ A * c + b), rtol=1e-12, atol=1e-12)
    W2 = RNG.normal(size=(2, D))           # Dout < Din branch (:220)
    y2, _ = O.gcn_conv(X, W2, None, g, "relu")
    np.testing.assert_allclose(y2, np.maximum(W2 @ (X * c) @ A * c, 0), rtol=1e-12, atol=1e-12)


# ---- finite-difference checks of the VJPs -----------------------------------------------------

def fd_check(f, x, analytic, eps=1e-6, ntries=6, tol=2e-6):
    """f: array -> scalar loss."""
    rng = np.random.default_rng(1)
    for _ in range(ntries):
        idx = tuple(rng.integers(0, s) for s in x.shape)
        xp, xm = x.copy(), x.copy()
        xp[idx] += eps
        xm[idx] -= eps
        num = (f(xp) - f(xm)) / (2 * eps)
        assert abs(num - analytic[idx]) <= tol * max(1.0, abs(num)), (idx, num, analytic[idx])


def rand_graph(N, E, rng=RNG, **kw):
    return O.Graph(rng.integers(1, N + 1, E), rng.integers(1, N + 1, E), num_nodes=N, **kw)


@pytest.mark.parametrize("dims", [(4, 6), (6, 3)])
@pytest.mark.parametrize("weighted", [False, True])
def test_gcn_backward_fd(dims, weighted):
    Din, Dout = dims
    N, E = 9, 20
    g = rand_graph(N, E)
    ew = RNG.random(E) + 0.5 if weighted else None
    X, W, b = RNG.normal(size=(Din, N)), RNG.normal(size=(Dout, Din)), RNG.normal(size=(Dout, 1))
    R = RNG.normal(size=(Dout, N))
    y, c = O.gcn_conv(X, W, b, g, "swish", edge_weight=ew)
    gr = O.gcn_conv_backward(c, R)
    fd_check(lambda x: (O.gcn_conv(x, W, b, g, "swish", edge_weight=ew)[0] * R).sum(), X, gr["x"])
    fd_check(lambda w: (O.gcn_conv(X, w, b, g, "swish", edge_weight=ew)[0] * R).sum(), W, gr["weight"])
    fd_check(lambda bb: (O.gcn_conv(X, W, bb, g, "swish", edge_weight=ew)[0] * R).sum(), b, gr["bias"])
    if weighted:   # the message factor and the weighted-degree normalisation (src/layers.jl:206-231), every edge
        fd_check(lambda w_: (O.gcn_conv(X, W, b, g, "swish", edge_weight=w_)[0] * R).sum(), ew, gr["edge_weight"], ntries=E)


@pytest.mark.parametrize("aggr", ["mean", "+", "max", "*"])
def test_mppde_backward_fd(aggr):
    N, E, G = 8, 24, 2
    base = rand_graph(4, 12, ndata={"u": RNG.random((2, 4)), "x": RNG.random((1, 4))},
                      edata={"w": RNG.random((1, 12))}, gdata={"θ": RNG.random(3)})
    g = O.batch([base, base.copy(gdata={"θ": RNG.random(3)})])
    h = 5
    phi = rand_mlp([2 * h + 3 + 1 + 3, 7, 6], ["swish", "tanh"])
    psi = rand_mlp([h + 6 + 3, 4], ["identity"])
    X = RNG.normal(size=(h, N))
    R = RNG.normal(size=(4, N))
    y, c = O.mppde_conv(X, phi, psi, g, aggr)
    gr = O.mppde_conv_backward(c, R)
    fd_check(lambda x: (O.mppde_conv(x, phi, psi, g, aggr)[0] * R).sum(), X, gr["x"])

    def with_w(w, which, k):
        p2 = [dict(L) for L in (phi if which == "phi" else psi)]
        p2[k]["weight"] = w
        return (O.mppde_conv(X, p2 if which == "phi" else phi, psi if which == "phi" else p2, g, aggr)[0] * R).sum()
    fd_check(lambda w: with_w(w, "phi", 0), phi[0]["weight"], gr["phi"][0]["weight"])
    fd_check(lambda w: with_w(w, "phi", 1), phi[1]["weight"], gr["phi"][1]["weight"])
    fd_check(lambda w: with_w(w, "psi", 0), psi[0]["weight"], gr["psi"][0]["weight"])


def test_vmh_and_edgeconv_backward_fd():
    N, E = 7, 18
    g = rand_graph(N, E, ndata={"x": RNG.random((2, N))})
    h = 3
    phi = rand_mlp([2 * h + 2, 6, 5], ["tanh", "tanh"])
    gamma = rand_mlp([h + 5, 4], ["tanh"])
    X = RNG.normal(size=(h, N))
    R = RNG.normal(size=(4, N))
    y, c = O.vmh_conv(X, phi, gamma, g)
    gr = O.vmh_conv_backward(c, R)
    fd_check(lambda x: (O.vmh_conv(x, phi, gamma, g)[0] * R).sum(), X, gr["x"])
    R2 = RNG.normal(size=(5, N))
    y, c = O.explicit_edge_conv(X, phi, g)
    gr = O.explicit_edge_conv_backward(c, R2)
    fd_check(lambda x: (O.explicit_edge_conv(x, phi, g)[0] * R2).sum(), X, gr["x"])

    def with_w(w):
        p2 = [dict(L) for L in phi]
        p2[0]["weight"] = w
        return (O.explicit_edge_conv(X, p2, g)[0] * R2).sum()
    fd_check(with_w, phi[0]["weight"], gr["phi"][0]["weight"])


def test_gno_backward_fd_and_reshape_convention():
    N, E, cin, cout = 6, 14, 3, 4
    g = rand_graph(N, E, ndata={"a": RNG.random((1, N)), "x": RNG.random((2, N))})
    phi = rand_mlp([6, 5, cin * cout], ["relu", "identity"])
    W, b = RNG.normal(size=(cout, cin)), RNG.normal(size=(cout, 1))
    X = RNG.normal(size=(cin, N))
    R = RNG.normal(size=(cout, N))
    y, c = O.gno_conv(X, phi, W, b, g, cin, cout, "tanh")
    # literal check of K_e[o,i] = phi_out[o + out*i]  (src/layers.jl:527, column-major reshape)
    kin = np.concatenate([c["g"].ndata["a"][:, g.t], c["g"].ndata["x"][:, g.t],
                          c["g"].ndata["a"][:, g.s], c["g"].ndata["x"][:, g.s]], 0)
    Wk, _ = O.mlp_forward(phi, kin)
    m = np.zeros((cout, E))
    for e in range(E):
        K = Wk[:, e].reshape(cout, cin, order="F")
        m[:, e] = K @ X[:, g.s[e]]
    np.testing.assert_allclose(m, c["m"], rtol=1e-12, atol=1e-12)
    gr = O.gno_conv_backward(c, R)
    fd_check(lambda x: (O.gno_conv(x, phi, W, b, g, cin, cout, "tanh")[0] * R).sum(), X, gr["x"])

    def with_w(w):
        p2 = [dict(L) for L in phi]
        p2[1]["weight"] = w
        return (O.gno_conv(X, p2, W, b, g, cin, cout, "tanh")[0] * R).sum()
    fd_check(with_w, phi[1]["weight"], gr["phi"][1]["weight"])
    fd_check(lambda w: (O.gno_conv(X, phi, w, b, g, cin, cout, "tanh")[0] * R).sum(), W, gr["weight"])


def test_gat_backward_fd():
    N, E, H, C, Din = 7, 16, 2, 3, 5
    g = rand_graph(N, E)
    W = RNG.normal(size=(H * C, Din))
    a = RNG.normal(size=(2 * C, H))
    b = RNG.normal(size=(H * C,))
    X = RNG.normal(size=(Din, N))
    R = RNG.normal(size=(H * C, N))
    y, c = O.gat_conv(X, W, a, b, g, H, C, "tanh")
    assert y.shape == (H * C, N)
    # softmax rows sum to one per (head, target)
    np.testing.assert_allclose(O.scatter("+", c["alpha"], c["g"].t, N), 1.0, rtol=1e-12)
    gr = O.gat_conv_backward(c, R)
    f = lambda x=X, w=W, aa=a: (O.gat_conv(x, w, aa, b, g, H, C, "tanh")[0] * R).sum()
    fd_check(lambda x: f(x=x), X, gr["x"])
    fd_check(lambda w: f(w=w), W, gr["weight"])
    fd_check(lambda aa: f(aa=aa), a, gr["a"])


@pytest.mark.parametrize("tab", ["euler", "tsit5"])
def test_node_adjoint_fd(tab):
    N, E, D = 10, 30, 4
    g = rand_graph(N, E)
    params = [dict(weight=RNG.normal(size=(D, D)) * 0.5, bias=RNG.normal(size=(D, 1)) * 0.1) for _ in range(2)]
    u0 = RNG.normal(size=(D, N))
    tb = O.TABLEAUS[tab]
    uT, du0, acc = O.gcn2_node_loss_and_grads(params, g, u0, tb, 0.1, 3, "tanh")
    loss = lambda u: O.gcn2_node_loss_and_grads(params, g, u, tb, 0.1, 3, "tanh")[0].sum()
    fd_check(loss, u0, du0)

    def loss_w(w, layer):
        p2 = [dict(p) for p in params]
        p2[layer]["weight"] = w
        return O.gcn2_node_loss_and_grads(p2, g, u0, tb, 0.1, 3, "tanh")[0].sum()
    fd_check(lambda w: loss_w(w, 0), params[0]["weight"], acc[0]["weight"])
    fd_check(lambda w: loss_w(w, 1), params[1]["weight"], acc[1]["weight"])


def test_generators_are_deterministic():
    a = S.splitmix64(7, 4)
    assert a.dtype == np.uint64 and len(set(a.tolist())) == 4
    # splitmix64 reference value for seed 0, first output
    assert int(S.splitmix64(0, 1)[0]) == 0xE220A8397B1DCDAF
    pts, s, t = S.closest_pairs_graph(512, 2048, 3)
    assert s.size == 4096 and set(zip(s.tolist(), t.tolist())) == set(zip(t.tolist(), s.tolist()))
    assert np.all(s != t)
    pts2, s2, t2 = S.closest_pairs_graph(512, 2048, 3)
    assert np.array_equal(s, s2) and np.array_equal(t, t2)
    # brute-force check: these are the 2048 closest pairs
    d2 = ((pts[:, None, :] - pts[None, :, :]) ** 2).sum(-1)
    iu = np.triu_indices(512, 1)
    thr = np.sort(d2[iu])[2047]
    assert np.all(((pts[s] - pts[t]) ** 2).sum(-1) <= thr + 1e-15)


# ---- derived-graph handle restatement (integer work): structural properties on random and edge-case graphs ------------------

@pytest.mark.parametrize("n,m,seed", [(70, 300, 0), (33, 90, 1), (5, 0, 2), (64, 1200, 3), (3, 4, 4)])
def test_derived_graph_properties(n, m, seed):
    rng = np.random.default_rng(seed)
    s, t = rng.integers(0, n, m), rng.integers(0, n, m)
    w = (0.5 + rng.random(m)).astype(np.float32)
    d = O.derived_graph(s, t, n, None, True, w, True)
    assert sorted(d["order"].tolist()) == list(range(n))                       # a permutation of the nodes
    for tag, key, other in (("t", t, s), ("s", s, t)):
        c = d[tag]
        assert c["rowptr"][0] == 0 and c["rowptr"][-1] == m and (np.diff(c["rowptr"]) >= 0).all()
        assert sorted(c["eid"].tolist()) == list(range(m))
        for r in range(n):
            e = c["eid"][c["rowptr"][r]:c["rowptr"][r + 1]]
            assert (key[e] == r).all() and (np.diff(e) > 0).all()              # COO order inside a row (NNlib's scatter order)
            assert np.array_equal(c["col"][c["rowptr"][r]:c["rowptr"][r + 1]], other[e])
        assert np.array_equal(c["ent"][:, 0], c["col"])
        coef = c["ent"][:, 1].copy().view(np.float32)
        assert np.array_equal(coef, (w[c["eid"]] * d["c"][c["col"]]).astype(np.float32))
    assert np.array_equal(d["s"]["xpos"][d["t"]["xpos"]], np.arange(m))         # the two lists index each other
    deg = np.zeros(n); np.add.at(deg, t, w.astype(np.float64))
    np.testing.assert_allclose(d["c"], 1 / np.sqrt(deg + 1), rtol=1e-6)          # src/layers.jl:224-225 with padded weights
    # every slot byte of a fitting tile points at the halo entry of that entry's column
    for tag in ("t", "s"):
        c = d[tag]
        for tl in range(c["tile_info"].shape[0]):
            if c["tile_info"][tl, 0] == 0:
                continue
            for k in range(O.TILE_ROWS):
                pos = tl * O.TILE_ROWS + k
                if pos >= n:
                    continue
                v = d["order"][pos]
                assert c["halo"][tl, k, 0] == v
                for j in range(c["rowptr"][v + 1] - c["rowptr"][v]):
                    assert c["halo"][tl, c["slots"][pos, j], 0] == c["col"][c["rowptr"][v] + j]
                    assert c["slot_w"][pos, j] == w[c["eid"][c["rowptr"][v] + j]]


def test_locality_order_keeps_clusters_compact():
    # a 2-D grid graph: BFS-grown 32-node clusters touch far fewer distinct rows than index order would
    k = 24
    idx = np.arange(k * k).reshape(k, k)
    pairs = np.concatenate([np.stack([idx[:, :-1].ravel(), idx[:, 1:].ravel()]), np.stack([idx[:-1].ravel(), idx[1:].ravel()])], axis=1)
    s, t = np.concatenate([pairs[0], pairs[1]]), np.concatenate([pairs[1], pairs[0]])
    perm = np.random.default_rng(0).permutation(k * k)                           # hide the geometry in the labels
    d = O.derived_graph(perm[s], perm[t], k * k, None, True, None, False)
    ident = O.derived_graph(perm[s], perm[t], k * k, np.arange(k * k, dtype=np.int32), True, None, False)
    assert d["t"]["halo_ok"]
    assert d["t"]["tile_info"][:, 0].mean() < 0.75 * np.where(ident["t"]["tile_info"][:, 0] > 0, ident["t"]["tile_info"][:, 0], 96).mean()


def test_optimiser_rules_closed_forms():
    # Adam, t = 1: bias-corrected moments are g and g^2  =>  x1 = x0 - eta g / (|g| + eps)   [UPSTREAM Optimisers.jl]
    g = np.array([3.0, -0.5, 1e-3, 0.0], np.float32)
    x, st = O.adam_step(np.zeros(4, np.float32), g, O.adam_init(g), 0.1)
    np.testing.assert_allclose(x, -0.1 * g / (np.abs(g) + 1e-8), rtol=1e-5, atol=1e-9)
    # constant gradient: every later step has the same size eta (m/c1 = g, v/c2 = g^2 for all t)
    x2, st = O.adam_step(x, g, st, 0.1)
    np.testing.assert_allclose(x2 - x, x, rtol=1e-4, atol=1e-9)
    # Rprop: first step has no remembered gradient -> step size unchanged; same sign grows by 1.2, a flip halves and skips
    xr, rs = O.rprop_step(np.zeros(3, np.float32), np.array([1.0, -2.0, 0.0], np.float32), O.rprop_init(np.zeros(3, np.float32), 0.1))
    np.testing.assert_allclose(xr, [-0.1, 0.1, 0.0], rtol=1e-6)
    xr2, rs = O.rprop_step(xr, np.array([5.0, 3.0, 1.0], np.float32), rs)
    np.testing.assert_allclose(rs["step"], [0.12, 0.05, 0.1], rtol=1e-6)
    np.testing.assert_allclose(xr2 - xr, [-0.12, 0.0, -0.1], rtol=1e-6, atol=1e-9)


# ---- neighbour search restatement (radius_graph / knn_graph / spatial_order) -------------------------------------


@pytest.mark.parametrize("dim", [1, 2, 3])
def test_radius_and_knn_restatement_against_kdtree(dim):
    """the brute-force float32 restatement against scipy's cKDTree on the same points in float64, with the radius
    chosen in a gap of the pair-distance distribution so that float32 rounding cannot move a pair across it"""
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(dim)
    n = 700
    P = rng.random((dim, n)).astype(np.float32)
    P64 = P.T.astype(np.float64)
    tree = cKDTree(P64)
    r0 = {1: 0.004, 2: 0.06, 3: 0.15}[dim]
    dd = np.sort(np.sqrt(((P64[:, None, :] - P64[None, :, :]) ** 2).sum(-1)).reshape(-1))
    k0 = np.searchsorted(dd, r0)
    gaps = dd[k0:k0 + 200][1:] - dd[k0:k0 + 200][:-1]
    j = int(np.argmax(gaps))
    r = float(0.5 * (dd[k0 + j] + dd[k0 + j + 1]))
    assert gaps[j] > 1e-6 * r
    s, t = O.radius_graph(P, r)
    pairs = tree.query_pairs(r, output_type="ndarray")
    want = set(map(tuple, pairs.tolist())) | set(map(tuple, pairs[:, ::-1].tolist()))
    assert set(zip(s.tolist(), t.tolist())) == want
    assert np.array_equal(np.lexsort((s, t)), np.arange(s.size))               # by target, sources ascending
    so, to = O.radius_graph(P, r, dir="out")
    assert np.array_equal(so, t) and np.array_equal(to, s)
    sl, tl = O.radius_graph(P, r, self_loops=True)
    assert sl.size == s.size + n
    # k-NN: same neighbours, nearest first (distinct distances with probability one)
    k = 7
    s2, t2 = O.knn_graph(P, k)
    _, idx = tree.query(P64, k=k + 1)
    assert np.array_equal(t2, np.repeat(np.arange(n), k))
    assert np.array_equal(s2.reshape(n, k), idx[:, 1:])
    s3, _ = O.knn_graph(P, k, self_loops=True)
    assert np.array_equal(s3.reshape(n, k), idx[:, :k])
    # graph_indicator: the batch is the union of its members
    gi = np.repeat([1, 2], n // 2)
    sb, tb = O.radius_graph(P, r, graph_indicator=gi)
    a = O.radius_graph(P[:, :n // 2], r)
    b = O.radius_graph(P[:, n // 2:], r)
    assert np.array_equal(sb, np.r_[a[0], b[0] + n // 2]) and np.array_equal(tb, np.r_[a[1], b[1] + n // 2])
    with pytest.raises(ValueError):
        O.knn_graph(P[:, :5], 5)


def test_spatial_order_restatement_properties():
    rng = np.random.default_rng(0)
    for dim in (1, 2, 3):
        P = rng.random((dim, 4096)).astype(np.float32)
        gi = np.sort(rng.integers(0, 3, 4096))
        o = O.spatial_order(P, gi)
        assert sorted(o.tolist()) == list(range(4096))
        assert np.array_equal(gi[o], np.sort(gi))                             # graph by graph
        if dim == 1:
            assert np.all(np.diff(P[0, O.spatial_order(P)]) >= -1e-6)         # the coordinate itself (30-bit quantisation)
    # Hilbert curve on the full 2^k x 2^k lattice: consecutive cells are edge neighbours
    k = 5
    xs, ys = np.meshgrid(np.arange(1 << k), np.arange(1 << k), indexing="ij")
    d = O._hilbert2(xs.reshape(-1), ys.reshape(-1), k)
    assert sorted(d.tolist()) == list(range(1 << (2 * k)))
    o = np.argsort(d)
    step = np.abs(np.diff(xs.reshape(-1)[o])) + np.abs(np.diff(ys.reshape(-1)[o]))
    assert np.all(step == 1)
    # locality: on uniform points 32 consecutive nodes span far less than a random 32-subset
    P = rng.random((2, 16384)).astype(np.float32)
    o = O.spatial_order(P)
    tiles = P[:, o].reshape(2, -1, 32)
    span = (tiles.max(axis=2) - tiles.min(axis=2)).max(axis=0)
    assert np.median(span) < 0.1


# ---- the pre-scaled form of the solver plan is the same function ------------------------------------------------


@pytest.mark.parametrize("tab,activation", [("euler", "relu"), ("tsit5", "tanh"), ("tsit5", "swish"), ("tsit5", "relu")])
def test_prescaled_pipeline_algebra(tab, activation):
    """the HIP solver plan keeps u, the layer outputs and the adjoint products multiplied by
    c[row].  In float64 that form must reproduce the plain solve and its discrete adjoint to rounding, on a DIRECTED
    graph (forward walks the in-lists, the pullback the out-lists) with multi-edges and pre-existing self loops."""
    rng = np.random.default_rng(7)
    n, d, m = 40, 6, 170
    s, t = rng.integers(0, n, m), rng.integers(0, n, m)
    g = O.Graph(s, t, num_nodes=n, index_base=0)
    params = [dict(weight=rng.normal(size=(d, d)) * 0.5, bias=rng.normal(size=(d, 1)) * 0.2) for _ in range(2)]
    u0 = rng.normal(size=(d, n))
    ref = O.gcn2_node_loss_and_grads(params, g, u0, O.TABLEAUS[tab], 0.1, 3, activation)
    pre = O.gcn2_node_prescaled(params, g, u0, O.TABLEAUS[tab], 0.1, 3, activation)
    assert np.allclose(pre[0], ref[0], rtol=1e-11, atol=1e-12)
    assert np.allclose(pre[1], ref[1], rtol=1e-10, atol=1e-12)
    for k in range(2):
        assert np.allclose(pre[2][k]["weight"], ref[2][k]["weight"], rtol=1e-10, atol=1e-11)
        assert np.allclose(pre[2][k]["bias"], ref[2][k]["bias"], rtol=1e-10, atol=1e-11)
