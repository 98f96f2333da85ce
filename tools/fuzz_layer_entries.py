"""Randomised cases of the layer-level C entries (ngpde_edge_layer_*, ngpde_gno_layer_*: api_layers.hip) against the same layers composed
from the primitives (tests/composed.py): outputs and every gradient bit for bit, training and inference.  What varies: the layer kind,
the graph (closest pairs / random local lists with isolated nodes and rows of more than 32 entries / batched periodic meshes / no edges
at all), node counts across partial tiles, state as a matrix or a NamedTuple, extra node data, edge features, per-graph data, MLP
depths 1 - 5 and widths 1 - 72 (64 => 64 tails reach the specialised message kernels), activations, aggregations + / mean / max / min / *.
The entry's plan (fused message launch or primitives, one-launch pullback or saved pre-activations, chain / pair Dense launches, the
three GNO message forms) is whatever make_plan decides for the case.  With ORACLE=1 the entry's output, input gradient and phi's
gradients of every ExplicitEdgeConv / VMHConv / MPPDEConv case with a matrix state are also compared with
the float64 oracle (oracle/ngpde_oracle.py) at the suite's tolerances (1e-4 / 5e-4 of the largest value).
usage: [ORACLE=1] python3 tools/fuzz_layer_entries.py [cases=60] [seed=1] [verbose=0]      exit code 1 when a case differs"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ngpde_amd as ng  # noqa: E402
from ngpde_amd import synth as S  # noqa: E402
import composed  # noqa: E402
from oracle import ngpde_oracle as O  # noqa: E402  (the checker)

DEV = "cuda"
ACTS = ["identity", "relu", "tanh", "sigmoid", "swish", "gelu", "leakyrelu", "elu", "softplus"]
AGGRS = ["+", "mean", "max", "min", "*"]
WIDTHS = [1, 2, 3, 5, 8, 12, 16, 24, 32, 40, 48, 60, 64, 72]


def grad_leaves(t):
    for v in t.values():
        if isinstance(v, dict):
            yield from grad_leaves(v)
        else:
            yield v


def prep(ps0, seed):
    """device copy of the freshly initialised parameters with random biases and requires_grad"""
    rng = np.random.default_rng(seed)
    ps = ng.to_device(ps0, DEV)

    def walk(d):
        for k, v in d.items():
            if isinstance(v, dict):
                walk(v)
            else:
                if k == "bias":
                    d[k] = torch.as_tensor(rng.normal(size=tuple(v.shape)).astype(np.float32) * 0.3, device=DEV)
                d[k].requires_grad_(True)
    walk(ps)
    return ps


def random_graph(rng, kind_hint):
    """(s, t, N, n_graphs)"""
    form = rng.choice(["pairs", "local", "mesh", "empty"], p=[0.45, 0.3, 0.2, 0.05]) if kind_hint != "mppde" else rng.choice(["mesh", "pairs"], p=[0.7, 0.3])
    if form == "mesh":
        n, traj = int(rng.choice([32, 48, 100, 256])), int(rng.integers(1, 6))
        s, t = S.periodic_mesh_batch(n, traj)
        return s, t, n * traj, traj
    N = int(rng.choice([33, 64, 97, 200, 511, 700, 1025]))
    if form == "pairs":
        _, s, t = S.closest_pairs_graph(N, int(rng.integers(1, 4)) * N, seed=int(rng.integers(1, 1000)))
        return s, t, N, 1
    if form == "empty":
        return np.zeros(0, np.int64), np.zeros(0, np.int64), N, 1
    ss, tt = [], []
    for i in range(N):
        k = int(rng.choice([0, 1, 2, 3, 5, 8, 40], p=[0.1, 0.2, 0.2, 0.2, 0.15, 0.1, 0.05]))      # isolated nodes; a few rows beyond 32 entries
        for j in rng.integers(0, N, size=k):
            ss.append(int(j)); tt.append(i)
    if not ss:
        ss, tt = [0], [1]
    return np.array(ss), np.array(tt), N, 1


def mlp(rng, din, dout, depth, last_act="identity"):
    dims = [din] + [int(rng.choice(WIDTHS)) for _ in range(depth - 1)] + [dout]
    return ng.Chain(*[ng.Dense(dims[l], dims[l + 1], str(rng.choice(ACTS)) if l + 1 < depth else last_act) for l in range(depth)]) if depth > 1 else \
        ng.Dense(din, dout, last_act)


def make_case(rng):
    kind = str(rng.choice(["edgeconv", "vmh", "mppde", "gno"], p=[0.25, 0.3, 0.3, 0.15]))
    s, t, N, n_graphs = random_graph(rng, kind)
    E = int(s.size)
    pos_dim = int(rng.integers(1, 4))
    ndata = {"x": rng.normal(size=(pos_dim, N)).astype(np.float32)}
    extra = int(rng.integers(0, 3)) if kind != "vmh" else 0
    for k in range(extra):
        ndata[f"f{k}"] = rng.normal(size=(1 + k, N)).astype(np.float32)
    kw = dict(ndata=ndata)
    aggr = str(rng.choice(AGGRS, p=[0.3, 0.4, 0.1, 0.1, 0.1]))
    if kind == "edgeconv":
        dh = int(rng.choice([1, 3, 6, 16, 64]))
        g = ng.GNNGraph(s, t, num_nodes=N, index_base=0, **kw)
        dother = sum(1 + k for k in range(extra))
        phi = mlp(rng, 2 * (dh + dother) + pos_dim, int(rng.choice(WIDTHS)), int(rng.integers(1, 5)), str(rng.choice(["identity", "tanh"])))
        return f"edgeconv N={N} E={E} dh={dh} aggr={aggr}", ng.ExplicitEdgeConv(phi, initialgraph=g, aggr=aggr), torch.randn(dh, N, device=DEV)
    if kind == "vmh":
        g = ng.GNNGraph(s, t, num_nodes=N, index_base=0, **kw)
        if rng.random() < 0.3:
            x = {"u": torch.randn(int(rng.integers(1, 4)), N, device=DEV), "v": torch.randn(int(rng.integers(1, 3)), N, device=DEV)}
            dh = sum(int(v.shape[0]) for v in x.values())
        else:
            dh = int(rng.choice([1, 2, 8, 64]))
            x = torch.randn(dh, N, device=DEV)
        dm = int(rng.choice(WIDTHS))
        phi = mlp(rng, 2 * dh + pos_dim, dm, int(rng.integers(1, 6)))
        gam = mlp(rng, dh + dm, int(rng.choice([1, dh, 7])), int(rng.integers(1, 5)))
        return f"vmh N={N} E={E} dh={dh} dm={dm} aggr={aggr}", ng.VMHConv(phi, gam, initialgraph=g, aggr=aggr), x
    if kind == "mppde":
        h = int(rng.choice([4, 10, 16, 32, 64, 64]))
        de = int(rng.choice([0, 0, 1, 3])) if E > 0 else 0
        dth = int(rng.choice([0, 1, 2]))
        if de:
            kw["edata"] = {"e": rng.normal(size=(de, E)).astype(np.float32)}
        if dth:
            kw["gdata"] = {"θ": rng.normal(size=(dth, n_graphs)).astype(np.float32)}
        g = ng.GNNGraph(s, t, num_nodes=N, index_base=0, num_graphs=n_graphs, **kw)
        dd = pos_dim + sum(1 + k for k in range(extra))
        dm = h if rng.random() < 0.5 else int(rng.choice(WIDTHS))
        depth = int(rng.integers(1, 5))
        phi = mlp(rng, 2 * h + dd + de + dth, dm, depth, str(rng.choice(["identity", "swish"])))
        if h == 64 and dm == 64 and depth == 2 and rng.random() < 0.7:      # BASELINE config 4's message MLP: the specialised kernels
            a1, a2 = str(rng.choice(["swish", "relu", "tanh"])), str(rng.choice(["swish", "identity"]))
            phi = ng.Chain(ng.Dense(2 * h + dd + de + dth, 64, a1), ng.Dense(64, 64, a2))
        psi = mlp(rng, h + dm + dth, int(rng.choice([h, 5])), int(rng.integers(1, 4)))
        return f"mppde N={N} E={E} h={h} dm={dm} de={de} dth={dth} graphs={n_graphs} aggr={aggr}", ng.MPPDEConv(phi, psi, initialgraph=g, aggr=aggr), \
            torch.randn(N, h, device=DEV).T
    cin, cout = int(rng.choice([4, 8, 16])), int(rng.choice([4, 8, 16, 32]))
    de = int(rng.choice([0, 2])) if E > 0 else 0
    if de:
        kw["edata"] = {"e": rng.normal(size=(de, E)).astype(np.float32)}
    g = ng.GNNGraph(s, t, num_nodes=N, index_base=0, **kw)
    ds = pos_dim + sum(1 + k for k in range(extra))
    k = int(rng.choice([4, 16, 64]))
    depth = int(rng.integers(1, 4))
    bias = bool(rng.random() < 0.7)
    if depth == 1:
        phi = ng.Dense(2 * ds + de, cin * cout)
    elif depth == 2:
        phi = ng.Chain(ng.Dense(2 * ds + de, k, str(rng.choice(["relu", "identity", "tanh"]))), ng.Dense(k, cin * cout, bias=bool(rng.random() < 0.8)))
    else:
        phi = ng.Chain(ng.Dense(2 * ds + de, 12, "tanh"), ng.Dense(12, k, "swish"), ng.Dense(k, cin * cout))
    gaggr = aggr if aggr != "*" else "mean"
    return f"gno N={N} E={E} {cin}=>{cout} depth={depth} aggr={gaggr}", ng.GNOConv((cin, cout), phi, str(rng.choice(["relu", "swish", "identity"])), initialgraph=g,
                                                                                  aggr=gaggr, bias=bias), torch.randn(cin, N, device=DEV)


def both_ways(layer, x, seed, training):
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


def omlp(layer, ps):
    pairs = [(layer, ps)] if isinstance(layer, ng.Dense) else [(l, ps[n]) for n, l in zip(layer.names(), layer.chain)]
    return [dict(weight=p["weight"].detach().cpu().double().numpy(), bias=p["bias"].detach().cpu().double().numpy() if "bias" in p else None,
                 act=l.activation) for l, p in pairs]


def off(a, ref, rtol, atol):
    a = a.detach().cpu().double().numpy()
    ref = np.asarray(ref, dtype=np.float64).reshape(a.shape)
    err = float(np.abs(a - ref).max()) if ref.size else 0.0
    bound = rtol * (float(np.abs(ref).max()) if ref.size else 0.0) + atol
    return None if err <= bound else f"max err {err:.3e} > {bound:.3e}"


def against_oracle(layer, x, seed):
    """[] or the list of quantities beyond tolerance; None when the case is not one the oracle leg covers"""
    if isinstance(x, dict) or isinstance(layer, ng.GNOConv):
        return None
    g = layer.initialgraph()
    s, t = g.edge_index(0)
    f64 = lambda d: {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)).astype(np.float64) for k, v in d.items()} or None
    og = O.Graph(np.asarray(s), np.asarray(t), num_nodes=g.num_nodes, index_base=0, num_graphs=g.num_graphs,
                 ndata=f64(g.ndata), edata=f64(g.edata), gdata=f64(g.gdata))
    ps0, st = ng.setup(seed, layer)
    ps = prep(ps0, seed)
    xs = x.detach().clone().requires_grad_(True)
    y, _ = layer(xs, ps, st)
    R = np.random.default_rng(seed + 1).normal(size=tuple(y.shape))
    if isinstance(layer, ng.ExplicitEdgeConv) and layer.aggr in ("max", "min"):   # a node without incoming edges keeps -inf / +inf (NNlib's scatter)
        R = R * torch.isfinite(y).cpu().numpy()
        y = torch.where(torch.isfinite(y), y, torch.zeros_like(y))
    (y * torch.as_tensor(R.astype(np.float32), device=DEV)).sum().backward()
    x64 = x.detach().cpu().double().numpy()
    if isinstance(layer, ng.ExplicitEdgeConv):
        yo, c = O.explicit_edge_conv(x64, omlp(layer.ϕ, ps), og, layer.aggr)
        gr, sub = O.explicit_edge_conv_backward(c, R), ps
    elif isinstance(layer, ng.VMHConv):
        yo, c = O.vmh_conv(x64, omlp(layer.ϕ, ps["ϕ"]), omlp(layer.γ, ps["γ"]), og, aggr=layer.aggr)
        gr, sub = O.vmh_conv_backward(c, R), ps["ϕ"]
    else:
        yo, c = O.mppde_conv(x64, omlp(layer.ϕ, ps["ϕ"]), omlp(layer.ψ, ps["ψ"]), og, aggr=layer.aggr)
        gr, sub = O.mppde_conv_backward(c, R), ps["ϕ"]
    if isinstance(layer, ng.ExplicitEdgeConv) and layer.aggr in ("max", "min"):
        yo = np.where(np.isfinite(yo), yo, 0.0)
    if not np.isfinite(yo).all() or np.abs(yo).max() > 1e6:
        return None    # (a product over many messages can overflow float32; an empty max / min inside VMH / MPPDE feeds inf to the update)
    bad = []
    for what, a, ref, rt, at in [("y", y, yo, 1e-4, 1e-5), ("dx", xs.grad, gr["x"], 5e-4, 1e-4)]:
        m = off(a, ref, rt, at)
        if m:
            bad.append((what, m))
    phi = layer.ϕ
    pairs = [("", sub)] if isinstance(phi, ng.Dense) else [(n + ".", sub[n]) for n in phi.names()]
    for (pref, p), og_l in zip(pairs, gr["phi"]):
        m = off(p["weight"].grad, og_l["weight"], 5e-4, 2e-4)
        if m:
            bad.append(("d phi." + pref + "weight", m))
    return bad


def differing(outs):
    a, b = outs
    bad = []
    for k, (u, v) in enumerate(zip(a, b)):
        if (u is None) != (v is None):
            bad.append((k, "one side has no gradient"))
        elif u is not None and not (u.shape == v.shape and torch.equal(u.contiguous().view(torch.int32), v.contiguous().view(torch.int32))):
            bad.append((k, f"max diff {float((u - v).abs().nan_to_num(0.0).max()):.3e}"))
    return bad


n_oracle = [0]


def run(cases, seed, verbose=False):
    rng = np.random.default_rng(seed)
    failures = []
    for case in range(cases):
        name, layer, x = make_case(rng)
        try:
            bad = differing(both_ways(layer, x, 100 + case, True)) + differing(both_ways(layer, x, 100 + case, False))
            if os.environ.get("ORACLE") == "1":
                ob = against_oracle(layer, x, 100 + case)
                n_oracle[0] += ob is not None
                bad += ob or []
        except Exception as e:  # noqa: BLE001 -- a case that raises on one path only is a finding too
            bad = [(-1, f"{type(e).__name__}: {e}")]
        if bad:
            failures.append((case, name, bad))
        if verbose or bad:
            print(f"case {case}: {name}: {'ok' if not bad else bad}", flush=True)
    return failures


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    sd = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    fails = run(n, sd, verbose=len(sys.argv) > 3 and sys.argv[3] == "1")
    print(f"fuzz_layer_entries: {n} cases, seed {sd}: {len(fails)} differing" + (f"; {n_oracle[0]} of them also against the float64 oracle" if n_oracle[0] else ""))
    sys.exit(1 if fails else 0)
