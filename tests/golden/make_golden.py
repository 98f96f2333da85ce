"""Generates the golden input/output vectors under tests/golden/*.npz.

The reference (Julia) cannot run in this image (no julia binary), so -- except for the
SpectralConv known-answer case, whose expected output is the ANALYTIC derivative the reference's own test uses
(/root/reference/test/runtests.jl:153-162) -- these vectors are outputs of the float64 numpy restatement
oracle/ngpde_oracle.py on fixed, formula-generated inputs (no RNG: they regenerate bit-identically anywhere).
They are committed so that
  * tests/test_golden.py (CPU) pins the oracle against drift, and checks every stored output AND gradient against an
    independent torch float64 autograd transcription of the same reference call sites (the oracle's hand-derived
    pullbacks never enter that computation);
  * tests/test_golden_gpu.py checks the HIP path against the same stored numbers through the layer API.

Graphs: `fix3` = the 3-node / 4-edge fixture of test/runtests.jl:11-13; `rad64` = 64 points of the R2 low-discrepancy
sequence in the unit square joined within radius 0.22 (directed both ways, ordered by (source, target)).

usage:  python tests/golden/make_golden.py        (rewrites every .npz next to this file)
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ngpde_oracle as O  # noqa: E402


def val(shape, seed, scale=1.0):
    """deterministic, RNG-free test data"""
    n = int(np.prod(shape))
    k = np.arange(n, dtype=np.float64)
    return (scale * (0.6 * np.sin(0.37 * k + 1.7 * seed) + 0.4 * np.cos(0.11 * k * (1 + seed % 3) + seed))).reshape(shape)


def dense(i, o, act, seed, bias=True):
    return dict(weight=val((o, i), seed, 1.0 / np.sqrt(i)), bias=val((o, 1), seed + 50, 0.3) if bias else None, act=act)


def graph_fix3():
    return dict(s=np.array([0, 0, 1, 2]), t=np.array([1, 2, 0, 0]), n=3)


def graph_rad64():
    k = np.arange(1, 65, dtype=np.float64)
    pts = np.stack([np.mod(k * 0.7548776662466927, 1.0), np.mod(k * 0.5698402909980532, 1.0)])
    d2 = ((pts[:, :, None] - pts[:, None, :]) ** 2).sum(axis=0)
    s, t = np.nonzero((d2 < 0.22 ** 2) & ~np.eye(64, dtype=bool))
    return dict(s=s, t=t, n=64, pts=pts)


def ograph(g, **kw):
    return O.Graph(g["s"], g["t"], num_nodes=g["n"], index_base=0, **kw)


def flat_mlp(prefix, layers, out):
    for i, L in enumerate(layers):
        out[f"{prefix}.{i}.weight"] = L["weight"]
        if L["bias"] is not None:
            out[f"{prefix}.{i}.bias"] = L["bias"]


def flat_mlp_grads(prefix, grads, out):
    for i, G in enumerate(grads):
        out[f"d.{prefix}.{i}.weight"] = G["weight"]
        if G.get("bias") is not None:
            out[f"d.{prefix}.{i}.bias"] = G["bias"]


def save(name, meta, arrays):
    arrays = {k: np.asarray(v) for k, v in arrays.items() if v is not None}
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrays)
    print(f"{name}: {sum(a.nbytes for a in arrays.values())} bytes")


def graph_arrays(g, ndata=None, edata=None, gdata=None, edge_weight=None):
    out = {"g.s": g["s"].astype(np.int64), "g.t": g["t"].astype(np.int64), "g.n": np.int64(g["n"])}
    for tag, d in (("ndata", ndata), ("edata", edata), ("gdata", gdata)):
        for k, v in (d or {}).items():
            out[f"g.{tag}.{k}"] = v
    if edge_weight is not None:
        out["g.edge_weight"] = edge_weight
    return out


def acts(layers):
    return [L["act"] for L in layers]


def case_gcn(gname, g, din, dout, act, weighted, seed, tag=""):
    E = g["s"].size
    ew = (0.5 + np.abs(val((E,), seed + 3))) if weighted else None
    og = ograph(g)
    x, W, b = val((din, g["n"]), seed), val((dout, din), seed + 1, 1 / np.sqrt(din)), val((dout, 1), seed + 2, 0.3)
    y, c = O.gcn_conv(x, W, b, og, act, True, edge_weight=ew)
    R = val(y.shape, seed + 9)
    gr = O.gcn_conv_backward(c, R)
    arr = graph_arrays(g, edge_weight=ew)
    arr.update({"x": x, "p.weight": W, "p.bias": b, "R": R, "y": y, "d.x": gr["x"], "d.weight": gr["weight"], "d.bias": gr["bias"]})
    if weighted:
        arr["d.edge_weight"] = gr["edge_weight"]        # the edge_weight ARGUMENT is differentiable (src/layers.jl:206-231)
    save(f"gcn_{gname}" + ("_weighted" if weighted else "") + tag, dict(layer="gcn", act=act, din=din, dout=dout, weighted=weighted), arr)


def case_edgeconv(gname, g, h, dpos, phi, aggr, seed):
    nd = {"x": g.get("pts") if dpos == 2 and "pts" in g else val((dpos, g["n"]), seed + 4)}
    og = ograph(g, ndata=nd)
    x = val((h, g["n"]), seed)
    y, c = O.explicit_edge_conv(x, phi, og, aggr)
    R = val(y.shape, seed + 9)
    gr = O.explicit_edge_conv_backward(c, R)
    arr = graph_arrays(g, ndata=nd)
    arr.update({"x": x, "R": R, "y": y, "d.x": gr["x"]})
    flat_mlp("phi", phi, arr)
    flat_mlp_grads("phi", gr["phi"], arr)
    save(f"edgeconv_{gname}", dict(layer="edgeconv", aggr=aggr, phi=acts(phi)), arr)


def case_vmh(gname, g, h, dpos, phi, gamma, aggr, seed):
    nd = {"x": g.get("pts") if dpos == 2 and "pts" in g else val((dpos, g["n"]), seed + 4)}
    og = ograph(g, ndata=nd)
    x = val((h, g["n"]), seed)
    y, c = O.vmh_conv(x, phi, gamma, og, aggr)
    R = val(y.shape, seed + 9)
    gr = O.vmh_conv_backward(c, R)
    arr = graph_arrays(g, ndata=nd)
    arr.update({"x": x, "R": R, "y": y, "d.x": gr["x"]})
    flat_mlp("phi", phi, arr); flat_mlp("gamma", gamma, arr)
    flat_mlp_grads("phi", gr["phi"], arr); flat_mlp_grads("gamma", gr["gamma"], arr)
    save(f"vmh_{gname}", dict(layer="vmh", aggr=aggr, phi=acts(phi), gamma=acts(gamma)), arr)


def case_mppde(gname, g, h, phi, psi, aggr, seed, with_edata):
    E = g["s"].size
    nd = {"u": val((1, g["n"]), seed + 4), "x": val((1, g["n"]), seed + 5)}
    ed = {"e": val((1, E), seed + 6)} if with_edata else None
    gd = {"θ": val((2, 1), seed + 7)}
    og = ograph(g, ndata=nd, edata=ed, gdata=gd)
    x = val((h, g["n"]), seed)
    y, c = O.mppde_conv(x, phi, psi, og, aggr)
    R = val(y.shape, seed + 9)
    gr = O.mppde_conv_backward(c, R)
    arr = graph_arrays(g, ndata=nd, edata=ed, gdata=gd)
    arr.update({"x": x, "R": R, "y": y, "d.x": gr["x"]})
    flat_mlp("phi", phi, arr); flat_mlp("psi", psi, arr)
    flat_mlp_grads("phi", gr["phi"], arr); flat_mlp_grads("psi", gr["psi"], arr)
    save(f"mppde_{gname}", dict(layer="mppde", aggr=aggr, phi=acts(phi), psi=acts(psi)), arr)


def case_gno(gname, g, cin, cout, phi, act, aggr, seed):
    nd = {"a": val((1, g["n"]), seed + 4), "x": g["pts"] if "pts" in g else val((2, g["n"]), seed + 5)}
    og = ograph(g, ndata=nd)
    x = val((cin, g["n"]), seed)
    W, b = val((cout, cin), seed + 1, 1 / np.sqrt(cin)), val((cout, 1), seed + 2, 0.3)
    y, c = O.gno_conv(x, phi, W, b, og, cin, cout, act, aggr)
    R = val(y.shape, seed + 9)
    gr = O.gno_conv_backward(c, R)
    arr = graph_arrays(g, ndata=nd)
    arr.update({"x": x, "p.linear.weight": W, "p.linear.bias": b, "R": R, "y": y, "d.x": gr["x"],
                "d.linear.weight": gr["weight"], "d.linear.bias": gr["bias"]})
    flat_mlp("phi", phi, arr)
    flat_mlp_grads("phi", gr["phi"], arr)
    save(f"gno_{gname}", dict(layer="gno", aggr=aggr, act=act, cin=cin, cout=cout, phi=acts(phi)), arr)


def case_gat(gname, g, din, heads, c, act, concat, seed, tag=""):
    og = ograph(g)
    x = val((din, g["n"]), seed)
    W, a = val((c * heads, din), seed + 1, 1 / np.sqrt(din)), val((2 * c, heads), seed + 2, 0.5)
    b = val((c * heads if concat else c,), seed + 3, 0.3)
    y, cc = O.gat_conv(x, W, a, b, og, heads, c, act, concat=concat)
    R = val(y.shape, seed + 9)
    gr = O.gat_conv_backward(cc, R)
    arr = graph_arrays(g)
    arr.update({"x": x, "p.weight": W, "p.a": a, "p.bias": b, "R": R, "y": y, "d.x": gr["x"], "d.weight": gr["weight"],
                "d.a": gr["a"], "d.bias": gr["bias"]})
    save(f"gat_{gname}" + ("" if concat else "_mean") + tag, dict(layer="gat", act=act, din=din, heads=heads, c=c, concat=concat), arr)


def case_spectral():
    # the reference's own known-answer test: expected values are the analytic derivatives (test/runtests.jl:153-162)
    n = 100
    xs = np.linspace(0.0, 2.0 * np.pi, n + 1)[1:]
    save("spectral_n100", dict(layer="spectral", n=n, tol_sum_abs2=1e-3),
         {"u_sin": np.sin(xs), "u_cos": np.cos(xs), "y_sin_analytic": np.cos(xs), "y_cos_analytic": -np.sin(xs),
          "y_sin_oracle": O.spectral_conv(np.sin(xs), O.spectral_graph(n), n),
          "y_cos_oracle": O.spectral_conv(np.cos(xs), O.spectral_graph(n), n)})


def case_node(g, d, tableau, nsteps, dt, seed):
    og = ograph(g)
    params = [dict(weight=val((d, d), seed + i, 1 / np.sqrt(d)), bias=val((d, 1), seed + 20 + i, 0.2)) for i in range(2)]
    u0 = val((d, g["n"]), seed + 5)
    uT, du0, grads = O.gcn2_node_loss_and_grads(params, og, u0, O.TABLEAUS[tableau], dt, nsteps, "relu")
    arr = graph_arrays(g)
    arr.update({"u0": u0, "uT": uT, "d.u0": du0})
    for i in range(2):
        arr[f"p.{i}.weight"], arr[f"p.{i}.bias"] = params[i]["weight"], params[i]["bias"]
        arr[f"d.{i}.weight"], arr[f"d.{i}.bias"] = grads[i]["weight"], grads[i]["bias"]
    save(f"node_gcn2_{tableau}", dict(layer="node_gcn2", tableau=tableau, nsteps=nsteps, dt=dt, act="relu", loss="sum(u(T))"), arr)


def main():
    f3, r64 = graph_fix3(), graph_rad64()
    # GCNConv(3 => 5) on the fixture (test/runtests.jl:15-24), wider + relu on the radius graph, an edge-weighted call (:227-231)
    case_gcn("fix3", f3, 3, 5, "identity", False, 1)
    case_gcn("rad64", r64, 8, 6, "relu", False, 2)
    case_gcn("rad64", r64, 8, 6, "tanh", True, 3)                      # Dout < Din: W applied first (:220-223)
    case_gcn("rad64", r64, 6, 8, "swish", True, 21, tag="_wide")        # Dout >= Din (:235-237)
    case_gcn("rad64", r64, 16, 16, "relu", True, 22, tag="_fused")      # a width the fused kernels take
    # ExplicitEdgeConv(Dense(4+4+3, 5)) (test/runtests.jl:27-37)
    case_edgeconv("fix3", f3, 4, 3, [dense(11, 5, "identity", 4)], "mean", 4)
    case_edgeconv("rad64", r64, 6, 2, [dense(14, 16, "tanh", 5), dense(16, 9, "tanh", 6)], "+", 5)
    # VMHConv (test/runtests.jl:40-54)
    case_vmh("fix3", f3, 5, 3, [dense(5 + 5 + 3, 5, "identity", 7)], [dense(5 + 5, 7, "identity", 8)], "mean", 7)
    case_vmh("rad64", r64, 4, 2, [dense(10, 12, "tanh", 9), dense(12, 8, "identity", 10)],
             [dense(12, 10, "swish", 11), dense(10, 3, "identity", 12)], "mean", 9)
    # MPPDEConv (test/runtests.jl:57-102): phi in = 2h + 2 + #edata + 2, psi in = h + m + 2
    case_mppde("fix3", f3, 5, [dense(5 + 5 + 2 + 2, 5, "identity", 13)], [dense(5 + 5 + 2, 7, "identity", 14)], "mean", 13, False)
    case_mppde("rad64", r64, 8, [dense(8 + 8 + 2 + 1 + 2, 16, "swish", 15), dense(16, 12, "swish", 16)],
               [dense(8 + 12 + 2, 16, "swish", 17), dense(16, 8, "identity", 18)], "mean", 15, True)
    # GNOConv (test/runtests.jl:123-151): phi in = 2 * (1 + 2) = 6
    case_gno("fix3", f3, 5, 7, [dense(6, 35, "identity", 19)], "identity", "mean", 19)
    case_gno("rad64", r64, 6, 5, [dense(6, 16, "relu", 20), dense(16, 30, "identity", 21)], "tanh", "mean", 20)
    # GAT-style aggregation
    case_gat("rad64", r64, 8, 2, 3, "relu", True, 22)
    case_gat("rad64", r64, 8, 2, 3, "identity", False, 23)
    # BASELINE config 3's shape (64 => 4 heads x 16): on the GPU this is the one-launch layer (ngpde_gat_layer_forward)
    case_gat("rad64", r64, 64, 4, 16, "tanh", True, 26, tag="_c3shape")
    # product aggregation (src/layers.jl:49) and a GNO shape whose message runs on the matrix pipe (out = 16, k = 16)
    case_edgeconv("rad64_prod", r64, 4, 2, [dense(10, 8, "tanh", 27), dense(8, 5, "tanh", 28)], "*", 27)
    case_gno("rad64_mfma", r64, 8, 16, [dense(6, 16, "relu", 29), dense(16, 128, "identity", 30)], "identity", "mean", 29)
    case_spectral()
    case_node(r64, 8, "tsit5", 3, 0.1, 24)
    case_node(r64, 8, "euler", 4, 0.05, 25)


if __name__ == "__main__":
    main()
