"""GPU parity against the committed golden vectors (tests/golden/*.npz): every layer type through the Lux-style layer
API -> C ABI -> HIP kernels, on the reference's 3-node fixture graph (test/runtests.jl:11-13) and a 64-node radius
graph, outputs and all gradients.  Tolerances are SURVEY.md section 8(d)'s: |y - y_ref| <= 1e-4 max|y_ref| + 1e-5 per
layer call, gradients 2e-4 relative (fp32 kernels against float64 vectors)."""
import glob
import json
import os

import numpy as np
import pytest
import torch

import ngpde_amd as ng

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, "*.npz")))


def load(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    return json.loads(bytes(z["meta"]).decode()), {k: z[k] for k in z.files if k != "meta"}


def close(a, ref, rtol, atol=1e-5, what=""):
    a = a.detach().cpu().double().numpy()
    ref = np.asarray(ref, dtype=np.float64).reshape(a.shape)
    err = np.abs(a - ref).max() if ref.size else 0.0
    bound = rtol * (np.abs(ref).max() if ref.size else 0.0) + atol
    assert err <= bound, f"{what}: max err {err:.3e} > {bound:.3e}"


def f32(a):
    return torch.as_tensor(np.asarray(a, dtype=np.float32))


def graph_of(d):
    feats = lambda tag: {k.split(".", 2)[2]: d[k].astype(np.float32) for k in d if k.startswith(f"g.{tag}.")}
    kw = {}
    for tag in ("ndata", "edata", "gdata"):
        if feats(tag):
            kw[tag] = feats(tag)
    return ng.GNNGraph(d["g.s"], d["g.t"], num_nodes=int(d["g.n"]), index_base=0, **kw)


def mlp_layer(d, prefix, acts):
    dense = []
    for i, a in enumerate(acts):
        w = d[f"{prefix}.{i}.weight"]
        dense.append(ng.Dense(w.shape[1], w.shape[0], a, bias=f"{prefix}.{i}.bias" in d))
    return dense[0] if len(dense) == 1 else ng.Chain(*dense)


def mlp_params(d, prefix, layer):
    """params of a Dense / Chain in the package's (Lux) naming, and the list of (tensor, gradient key)"""
    pairs = []

    def one(i):
        p = {"weight": f32(d[f"{prefix}.{i}.weight"]).to(DEV).requires_grad_(True)}
        pairs.append((p["weight"], f"d.{prefix}.{i}.weight"))
        if f"{prefix}.{i}.bias" in d:
            p["bias"] = f32(d[f"{prefix}.{i}.bias"]).to(DEV).requires_grad_(True)
            pairs.append((p["bias"], f"d.{prefix}.{i}.bias"))
        return p

    if isinstance(layer, ng.Dense):
        return one(0), pairs
    return {n: one(i) for i, n in enumerate(layer.names())}, pairs


def build(meta, d):
    g = graph_of(d)
    kind = meta["layer"]
    leaf = lambda key: f32(d[key]).to(DEV).requires_grad_(True)
    if kind == "gcn":
        l = ng.GCNConv((meta["din"], meta["dout"]), meta["act"], initialgraph=g)
        ps = {"weight": leaf("p.weight"), "bias": leaf("p.bias")}
        return l, ps, [(ps["weight"], "d.weight"), (ps["bias"], "d.bias")]
    if kind == "gat":
        l = ng.GATConv((meta["din"], meta["c"]), meta["act"], heads=meta["heads"], concat=meta["concat"], initialgraph=g)
        ps = {"weight": leaf("p.weight"), "a": leaf("p.a"), "bias": leaf("p.bias")}
        return l, ps, [(ps["weight"], "d.weight"), (ps["a"], "d.a"), (ps["bias"], "d.bias")]
    phi = mlp_layer(d, "phi", meta["phi"])
    pphi, pairs = mlp_params(d, "phi", phi)
    if kind == "edgeconv":
        return ng.ExplicitEdgeConv(phi, initialgraph=g, aggr=meta["aggr"]), pphi, pairs
    if kind in ("vmh", "mppde"):
        second = "gamma" if kind == "vmh" else "psi"
        sec = mlp_layer(d, second, meta[second])
        psec, pairs2 = mlp_params(d, second, sec)
        if kind == "vmh":
            return ng.VMHConv(phi, sec, initialgraph=g, aggr=meta["aggr"]), {"ϕ": pphi, "γ": psec}, pairs + pairs2
        return ng.MPPDEConv(phi, sec, initialgraph=g, aggr=meta["aggr"]), {"ϕ": pphi, "ψ": psec}, pairs + pairs2
    if kind == "gno":
        l = ng.GNOConv((meta["cin"], meta["cout"]), phi, meta["act"], initialgraph=g, aggr=meta["aggr"])
        lin = {"weight": leaf("p.linear.weight"), "bias": leaf("p.linear.bias")}
        return l, {"linear": lin, "ϕ": pphi}, pairs + [(lin["weight"], "d.linear.weight"), (lin["bias"], "d.linear.bias")]
    raise AssertionError(kind)


LAYER_CASES = [c for c in CASES if load(c)[0]["layer"] not in ("spectral", "node_gcn2")]


@pytest.mark.parametrize("case", LAYER_CASES)
def test_layer_matches_golden(case):
    meta, d = load(case)
    layer, ps, pairs = build(meta, d)
    _, st = ng.setup(0, layer)
    x = f32(d["x"]).to(DEV).requires_grad_(True)
    ew = None
    if meta["layer"] == "gcn" and meta["weighted"]:
        ew = f32(d["g.edge_weight"]).to(DEV).requires_grad_(True)       # the edge_weight argument is differentiable (src/layers.jl:206-231)
        y, _ = layer(x, ps, st, ew)
    else:
        y, _ = layer(x, ps, st)
    close(y, d["y"], rtol=1e-4, what=case + ": y")
    y.backward(f32(d["R"]).to(DEV))
    close(x.grad, d["d.x"], rtol=2e-4, what=case + ": dx")
    if ew is not None:
        close(ew.grad, d["d.edge_weight"], rtol=2e-4, what=case + ": d edge_weight")
    for tensor, key in pairs:
        close(tensor.grad, d[key], rtol=2e-4, what=f"{case}: {key}")


def test_spectral_matches_reference_known_answer():
    meta, d = load("spectral_n100")
    l = ng.SpectralConv(meta["n"])
    ps, st = ng.setup(0, l)
    for f in ("sin", "cos"):
        y, _ = l(f32(d[f"u_{f}"]).to(DEV), ps, st)
        err = y.detach().cpu().double().numpy() - d[f"y_{f}_analytic"]
        assert np.sum(err ** 2) < meta["tol_sum_abs2"]                           # test/runtests.jl:158,161
        close(y, d[f"y_{f}_oracle"], rtol=1e-4, what=f"spectral {f}")


@pytest.mark.parametrize("case", [c for c in CASES if c.startswith("node_gcn2")])
def test_node_solve_and_adjoint_match_golden(case):
    meta, d = load(case)
    g = graph_of(d)
    dim = d["u0"].shape[0]
    rhs = ng.Chain(ng.GCNConv((dim, dim), meta["act"], initialgraph=g), ng.GCNConv((dim, dim), meta["act"], initialgraph=g))
    node = ng.NeuralODE(rhs, solver=meta["tableau"], n_steps=meta["nsteps"], dt=meta["dt"])
    _, st = ng.setup(0, node)
    ps = {f"layer_{i + 1}": {"weight": f32(d[f"p.{i}.weight"]).to(DEV).requires_grad_(True),
                             "bias": f32(d[f"p.{i}.bias"]).to(DEV).requires_grad_(True)} for i in range(2)}
    u0 = f32(d["u0"]).to(DEV).requires_grad_(True)
    uT, _ = node(u0, ps, st)
    close(uT, d["uT"], rtol=2e-4, what="u(T)")
    uT.sum().backward()
    close(u0.grad, d["d.u0"], rtol=5e-4, atol=1e-4, what="du0")
    for i in range(2):
        close(ps[f"layer_{i + 1}"]["weight"].grad, d[f"d.{i}.weight"], rtol=5e-4, atol=1e-4, what=f"dW{i + 1}")
        close(ps[f"layer_{i + 1}"]["bias"].grad, d[f"d.{i}.bias"], rtol=5e-4, atol=1e-4, what=f"db{i + 1}")
