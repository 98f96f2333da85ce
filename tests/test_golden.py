"""CPU checks on the committed golden vectors (tests/golden/*.npz, generator tests/golden/make_golden.py):

  1. the numpy oracle regenerates every stored array (pins the oracle against drift);
  2. an INDEPENDENT torch float64 transcription of the reference call sites (/root/reference/src/layers.jl), with
     gradients from torch.autograd, reproduces every stored output and gradient -- the oracle's hand-derived pullbacks
     take no part in that computation;
  3. the SpectralConv vectors meet the reference's own known-answer criterion (test/runtests.jl:153-162).

Nothing here touches the GPU or /root/reference.
"""
import glob
import importlib.util
import json
import os

import numpy as np
import pytest
import torch

from oracle import ngpde_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, "*.npz")))


def load(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    return meta, {k: z[k] for k in z.files if k != "meta"}


def test_fixture_set_is_complete():
    kinds = {load(c)[0]["layer"] for c in CASES}
    assert kinds == {"gcn", "edgeconv", "vmh", "mppde", "gno", "gat", "spectral", "node_gcn2"}
    assert {"gcn_fix3", "edgeconv_fix3", "vmh_fix3", "mppde_fix3", "gno_fix3"} <= set(CASES)   # the reference's 3-node fixture


def test_oracle_regenerates_golden(tmp_path):
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLD, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    mg.HERE = str(tmp_path)
    mg.main()
    assert sorted(os.path.basename(p)[:-4] for p in glob.glob(str(tmp_path / "*.npz"))) == CASES
    for c in CASES:
        old, new = np.load(os.path.join(GOLD, c + ".npz")), np.load(str(tmp_path / (c + ".npz")))
        assert sorted(old.files) == sorted(new.files), c
        for k in old.files:
            if k == "meta":
                assert bytes(old[k]) == bytes(new[k]), c
            elif old[k].dtype.kind == "f":
                np.testing.assert_allclose(new[k], old[k], rtol=1e-12, atol=1e-13, err_msg=f"{c}:{k}")
            else:
                assert np.array_equal(old[k], new[k]), (c, k)


# ---- torch float64 transcription of the reference semantics (autograd supplies every gradient) ----------------------------

T = lambda a: a if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a), dtype=torch.float64)

ACT = {"identity": lambda z: z, "relu": torch.relu, "tanh": torch.tanh, "sigmoid": torch.sigmoid,
       "swish": lambda z: z * torch.sigmoid(z)}


def scatter_t(op, M, idx, n):
    if op == "*":                                          # scatter(*): neutral element 1 for an empty neighbourhood
        return torch.ones(M.shape[0], n, dtype=M.dtype).index_reduce(1, idx, M, "prod", include_self=True)
    out = torch.zeros(M.shape[0], n, dtype=M.dtype).index_add(1, idx, M)
    if op == "mean":
        cnt = torch.zeros(n, dtype=M.dtype).index_add(0, idx, torch.ones(idx.numel(), dtype=M.dtype))
        out = out / torch.clamp(cnt, min=1.0)            # mean over an empty neighbourhood is 0
    elif op != "+":
        raise NotImplementedError(op)
    return out


def mlp_t(layers, x):
    for W, b, a in layers:
        x = W @ x
        if b is not None:
            x = x + b.reshape(-1, 1)
        x = ACT[a](x)
    return x


class Params:
    """leaf tensors (requires_grad) by fixture key"""

    def __init__(self, d):
        self.d, self.leaves = d, {}

    def __call__(self, key):
        if key not in self.d:
            return None
        if key not in self.leaves:
            self.leaves[key] = T(self.d[key]).requires_grad_(True)
        return self.leaves[key]

    def mlp(self, prefix, acts):
        return [(self(f"{prefix}.{i}.weight"), self(f"{prefix}.{i}.bias"), a) for i, a in enumerate(acts)]


def graph_t(d):
    s, t, n = torch.as_tensor(d["g.s"]), torch.as_tensor(d["g.t"]), int(d["g.n"])
    feats = lambda tag: {k.split(".", 2)[2]: T(v) for k, v in d.items() if k.startswith(f"g.{tag}.")}
    return s, t, n, feats("ndata"), feats("edata"), feats("gdata")


def vcat(parts, ncols):
    parts = list(parts)
    return torch.cat(parts, dim=0) if parts else torch.zeros(0, ncols, dtype=torch.float64)


def fwd_gcn(meta, d, P, x):                                                     # src/layers.jl:200-239
    s, t, n, *_ = graph_t(d)
    loops = torch.arange(n)
    s2, t2 = torch.cat([s, loops]), torch.cat([t, loops])                       # add_self_loops (:211)
    w = torch.cat([T(d["g.edge_weight"]), torch.ones(n, dtype=torch.float64)]) if meta["weighted"] else torch.ones(s2.numel(), dtype=torch.float64)
    deg = torch.zeros(n, dtype=torch.float64).index_add(0, t2, w)               # degree(g; dir=:in, edge_weight) (:224)
    c = 1.0 / torch.sqrt(deg)
    h = x * c
    h = scatter_t("+", h[:, s2] * w, t2, n)                                     # copy_xj / e_mul_xj (:227-232)
    h = h * c
    return ACT[meta["act"]](P("p.weight") @ h + P("p.bias"))


def fwd_edgeconv(meta, d, P, x):                                                # :98-112
    s, t, n, nd, _, _ = graph_t(d)
    pos = nd["x"]
    inp = torch.cat([x[:, t], x[:, s], pos[:, s] - pos[:, t]], dim=0)
    return scatter_t(meta["aggr"], mlp_t(P.mlp("phi", meta["phi"]), inp), t, n)


def fwd_vmh(meta, d, P, x):                                                     # :312-332
    s, t, n, nd, _, _ = graph_t(d)
    pos = nd["x"]
    inp = torch.cat([x[:, t], x[:, s] - x[:, t], pos[:, s] - pos[:, t]], dim=0)
    m = scatter_t(meta["aggr"], mlp_t(P.mlp("phi", meta["phi"]), inp), t, n)
    return mlp_t(P.mlp("gamma", meta["gamma"]), torch.cat([x, m], dim=0))


def fwd_mppde(meta, d, P, x):                                                   # :390-422
    s, t, n, nd, ed, gd = graph_t(d)
    E = s.numel()
    theta = vcat(gd.values(), 1)
    dd, e = vcat(nd.values(), n), vcat(ed.values(), E)
    inp = torch.cat([x[:, t], x[:, s], dd[:, t] - dd[:, s], e, theta.repeat_interleave(E, dim=1)], dim=0)
    m = scatter_t(meta["aggr"], mlp_t(P.mlp("phi", meta["phi"]), inp), t, n)
    return mlp_t(P.mlp("psi", meta["psi"]), torch.cat([x, m, theta.repeat_interleave(n, dim=1)], dim=0))


def fwd_gno(meta, d, P, x):                                                     # :509-547
    s, t, n, nd, ed, _ = graph_t(d)
    E, cin, cout = s.numel(), meta["cin"], meta["cout"]
    sf = vcat(nd.values(), n)
    K = mlp_t(P.mlp("phi", meta["phi"]), torch.cat([sf[:, t], sf[:, s], vcat(ed.values(), E)], dim=0))
    # reshape(K, out, in, E) in column-major order: K3[o, i, e] = K[o + out * i, e]
    K3 = K.reshape(cin, cout, E).permute(1, 0, 2)
    m = torch.einsum("oie,ie->oe", K3, x[:, s])                                 # batched_mul (:529)
    agg = scatter_t(meta["aggr"], m, t, n)
    return ACT[meta["act"]](P("p.linear.weight") @ x + agg + P("p.linear.bias"))


def fwd_gat(meta, d, P, x):                                                     # GraphNeuralNetworks.jl GATConv
    s, t, n, *_ = graph_t(d)
    H, C = meta["heads"], meta["c"]
    loops = torch.arange(n)
    s2, t2 = torch.cat([s, loops]), torch.cat([t, loops])
    Wx = (P("p.weight") @ x).reshape(H, C, n)                                   # row h*C + c of W x <-> (c, h) column-major
    a = P("p.a")                                                                # (2C x H)
    ai, aj = a[:C].T, a[C:].T                                                   # [H][C]
    logit = torch.nn.functional.leaky_relu((Wx[:, :, t2] * ai[:, :, None]).sum(1) + (Wx[:, :, s2] * aj[:, :, None]).sum(1), 0.2)
    mx = torch.full((H, n), -float("inf"), dtype=torch.float64).scatter_reduce(1, t2.expand(H, -1), logit, "amax")
    ex = torch.exp(logit - mx[:, t2])
    alpha = ex / torch.zeros(H, n, dtype=torch.float64).index_add(1, t2, ex)[:, t2]
    out = torch.zeros(H, C, n, dtype=torch.float64).index_add(2, t2, Wx[:, :, s2] * alpha[:, None, :])
    z = out.reshape(H * C, n) if meta["concat"] else out.mean(dim=0)
    return ACT[meta["act"]](z + P("p.bias").reshape(-1, 1))


FWD = {"gcn": fwd_gcn, "edgeconv": fwd_edgeconv, "vmh": fwd_vmh, "mppde": fwd_mppde, "gno": fwd_gno, "gat": fwd_gat}
LAYER_CASES = [c for c in CASES if load(c)[0]["layer"] in FWD]


def grad_key(k):
    return "d." + (k[2:] if k.startswith("p.") else k)


@pytest.mark.parametrize("case", LAYER_CASES)
def test_autograd_transcription_matches_golden(case):
    meta, d = load(case)
    P = Params(d)
    x = T(d["x"]).requires_grad_(True)
    ew = None
    if "d.edge_weight" in d:           # the edge_weight ARGUMENT of GCNConv is a differentiable input (src/layers.jl:206-231)
        ew = T(d["g.edge_weight"]).clone().requires_grad_(True)
        d = dict(d)
        d["g.edge_weight"] = ew
    y = FWD[meta["layer"]](meta, d, P, x)
    np.testing.assert_allclose(y.detach().numpy(), d["y"], rtol=1e-10, atol=1e-12, err_msg=case + ": y")
    (y * T(d["R"])).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), d["d.x"], rtol=1e-9, atol=1e-11, err_msg=case + ": dx")
    if ew is not None:
        np.testing.assert_allclose(ew.grad.numpy(), d["d.edge_weight"], rtol=1e-9, atol=1e-11, err_msg=case + ": d edge_weight")
    stored = {k for k in d if k.startswith("d.") and k not in ("d.x", "d.edge_weight")}
    assert stored == {grad_key(k) for k in P.leaves}, case
    for k, leaf in P.leaves.items():
        np.testing.assert_allclose(leaf.grad.numpy(), d[grad_key(k)].reshape(leaf.shape), rtol=1e-9, atol=1e-11, err_msg=f"{case}: {k}")


@pytest.mark.parametrize("case", [c for c in CASES if c.startswith("node_gcn2")])
def test_autograd_through_solver_matches_golden(case):
    # loss = sum(u(T)) differentiated by backprop through every stage of the fixed-step solve
    meta, d = load(case)
    tab = O.TABLEAUS[meta["tableau"]]
    P = Params(d)
    gmeta = dict(act=meta["act"], weighted=False)

    def rhs(u):
        for i in range(2):
            Pi = lambda key, i=i: P(key.replace("p.", f"p.{i}."))
            u = fwd_gcn(gmeta, d, Pi, u)
        return u

    u = T(d["u0"]).requires_grad_(True)
    u0 = u
    for _ in range(meta["nsteps"]):
        ks = []
        for i in range(len(tab["c"])):
            ui = u
            for j, aij in enumerate(tab["a"][i]):
                if aij != 0.0:
                    ui = ui + meta["dt"] * aij * ks[j]
            ks.append(rhs(ui))
        u = u + meta["dt"] * sum(bi * k for bi, k in zip(tab["b"], ks) if bi != 0.0)
    np.testing.assert_allclose(u.detach().numpy(), d["uT"], rtol=1e-10, atol=1e-12)
    u.sum().backward()
    np.testing.assert_allclose(u0.grad.numpy(), d["d.u0"], rtol=1e-9, atol=1e-11)
    for k, leaf in P.leaves.items():
        np.testing.assert_allclose(leaf.grad.numpy(), d[grad_key(k)].reshape(leaf.shape), rtol=1e-9, atol=1e-11, err_msg=k)


def test_spectral_known_answer():
    meta, d = load("spectral_n100")
    for f in ("sin", "cos"):
        assert np.sum((d[f"y_{f}_oracle"] - d[f"y_{f}_analytic"]) ** 2) < meta["tol_sum_abs2"]   # test/runtests.jl:158,161
