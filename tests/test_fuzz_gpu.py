"""Randomised layer checks against the float64 oracle: the one-launch GAT layer (values and all gradients: partial tiles, isolated
nodes, 1 / 2 / 4 heads, all activations) and GCNConv with a trainable edge_weight argument (any widths, both orders of W, with and
without self loops).  Fixed seeds: the cases are the same in every run."""
import numpy as np
import pytest
import torch

import ngpde_amd as ng
from oracle import ngpde_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("seed", [2, 11])
def test_random_gat_and_weighted_gcn_layers_against_the_oracle(seed):
    CASES = 40
    rng = np.random.default_rng(seed)
    ACTS = ["identity", "relu", "tanh", "sigmoid", "swish", "gelu", "leakyrelu", "elu", "softplus"]


    def rel(a, ref):
        a = a.detach().cpu().double().numpy().reshape(np.asarray(ref).shape)
        return float(np.abs(a - ref).max() / (np.abs(ref).max() + 1e-12))


    def local_graph(n, max_deg, reach):
        ss, tt = [], []
        for i in range(n):
            k = int(rng.integers(0, max_deg + 1))
            for off in rng.choice(np.arange(-reach, reach + 1), size=min(k, 2 * reach), replace=False):
                if off != 0:
                    ss.append((i + off) % n); tt.append(i)
        if not ss:
            ss, tt = [0], [min(1, n - 1)]
        return np.array(ss), np.array(tt)


    failures = []
    for case in range(CASES):
        n = int(rng.choice([5, 31, 33, 64, 100, 257, 500]))
        act = str(rng.choice(ACTS)); loops = bool(rng.integers(0, 4) > 0)
        s, t = local_graph(n, int(rng.integers(1, 14)), int(rng.integers(1, min(8, max(2, n // 2)))))
        g, og = ng.GNNGraph(s, t, num_nodes=n, index_base=0), O.Graph(s, t, num_nodes=n, index_base=0)
        if case % 2 == 0:      # GAT layer
            H = int(rng.choice([1, 2, 4])); C = 64 // H; bias = bool(rng.integers(0, 2))
            l = ng.GATConv((64, C), act, heads=H, add_self_loops=loops, bias=bias, initialgraph=g)
            ps, st = ng.setup(case, l)
            ps = ng.to_device(ps, DEV)
            if bias: ps["bias"] = torch.randn_like(ps["bias"]) * 0.2
            for v in ps.values(): v.requires_grad_(True)
            x = torch.randn(64, n, device=DEV, requires_grad=True)
            y, _ = l(x, ps, st)
            R = rng.normal(size=(64, n))
            (y * torch.as_tensor(R, dtype=torch.float32, device=DEV)).sum().backward()
            pw = lambda k: ps[k].detach().cpu().double().numpy() if k in ps else None
            yo, c = O.gat_conv(x.detach().cpu().double().numpy(), pw("weight"), pw("a"), pw("bias"), og, H, C, act, concat=True, add_self_loops_=loops)
            gr = O.gat_conv_backward(c, R)
            errs = {"y": rel(y, yo), "dx": rel(x.grad, gr["x"]), "dW": rel(ps["weight"].grad, gr["weight"]), "da": rel(ps["a"].grad, gr["a"])}
            if bias: errs["db"] = rel(ps["bias"].grad, gr["bias"])
            name = f"GAT H={H} bias={bias}"
        else:                  # GCNConv with a trainable edge_weight
            din, dout = int(rng.choice([64, 32, 10, 5, 16, 48])), int(rng.choice([64, 32, 10, 12, 16]))
            if not loops:      # every node needs an incoming edge without self loops
                s, t = np.concatenate([s, np.arange(n)]), np.concatenate([t, (np.arange(n) + 1) % n])
                g, og = ng.GNNGraph(s, t, num_nodes=n, index_base=0), O.Graph(s, t, num_nodes=n, index_base=0)
            ew0 = rng.random(s.size) + 0.5
            l = ng.GCNConv((din, dout), act, initialgraph=g, add_self_loops=loops)
            ps, st = ng.setup(case, l)
            ps = ng.to_device(ps, DEV)
            ps["bias"] = torch.randn_like(ps["bias"]) * 0.2
            for v in ps.values(): v.requires_grad_(True)
            x = torch.randn(din, n, device=DEV, requires_grad=True)
            ew = torch.as_tensor(ew0.astype(np.float32), device=DEV).requires_grad_(True)
            y, _ = l(x, ps, st, ew)
            R = rng.normal(size=(dout, n))
            (y * torch.as_tensor(R, dtype=torch.float32, device=DEV)).sum().backward()
            W, b = ps["weight"].detach().cpu().double().numpy(), ps["bias"].detach().cpu().double().numpy()
            yo, cache = O.gcn_conv(x.detach().cpu().double().numpy(), W, b, og, act, loops, edge_weight=ew0.astype(np.float32).astype(np.float64))
            go = O.gcn_conv_backward(cache, R)
            errs = {"y": rel(y, yo), "dx": rel(x.grad, go["x"]), "dW": rel(ps["weight"].grad, go["weight"]), "dew": rel(ew.grad, go["edge_weight"])}
            name = f"GCN+ew {din}=>{dout}"
        worst = max(errs.values())
        okc = worst <= 5e-4 and all(np.isfinite(v) for v in errs.values())
        if not okc:
            failures.append((case, name, n, act, loops, errs))
    assert not failures, failures



@pytest.mark.parametrize("seed", [3, 17])
def test_random_small_dense_layers_against_float64(seed):
    # the one-launch Dense kernels for widths up to 64 (dense_small_bwd.hip: forward in one contraction pass, the whole pullback in one
    # launch) and the general kernels around their limits: random row counts (fewer than a tile, ragged, up to 70 000), widths 1 .. 70,
    # one to three blocks of a virtual vcat of which per-graph blocks carry no gradient, every activation, with and without bias --
    # y, dW, db and the blocks' gradients against a float64 restatement with torch
    import composed as F          # the primitives' autograd wrappers (tests/composed.py)
    rng = np.random.default_rng(seed)
    ACTS = ["identity", "relu", "tanh", "sigmoid", "swish", "gelu", "leakyrelu", "elu", "softplus"]
    REF = {"identity": lambda z: z, "relu": torch.relu, "tanh": torch.tanh, "sigmoid": torch.sigmoid, "swish": lambda z: z * torch.sigmoid(z),
           "gelu": lambda z: torch.nn.functional.gelu(z, approximate="tanh"), "leakyrelu": lambda z: torch.nn.functional.leaky_relu(z, 0.01),
           "elu": torch.nn.functional.elu, "softplus": torch.nn.functional.softplus}
    failures = []
    for case in range(40):
        n = int(rng.choice([1, 7, 63, 64, 65, 200, 1000, 3000, 18000, 65536, 65537, 70000]))
        nb = int(rng.integers(1, 4))
        widths = [int(rng.integers(1, 30)) for _ in range(nb)]
        if rng.integers(0, 3) == 0:
            widths = [int(rng.choice([16, 17, 60, 64, 65, 70]))]
        divs = [1] + [int(rng.choice([1, max(n // 3, 1)])) for _ in range(len(widths) - 1)]
        dout = int(rng.choice([1, 3, 16, 40, 60, 64, 65]))
        act = str(rng.choice(ACTS))
        bias = bool(rng.integers(0, 2))
        blocks = [torch.as_tensor(rng.normal(size=((n + rd - 1) // rd, w)), dtype=torch.float32, device=DEV).requires_grad_(rd == 1)
                  for w, rd in zip(widths, divs)]
        din = sum(widths)
        wt = torch.as_tensor(rng.normal(size=(din, dout)) / np.sqrt(din), dtype=torch.float32, device=DEV).requires_grad_(True)
        b = torch.as_tensor(rng.normal(size=dout), dtype=torch.float32, device=DEV).requires_grad_(True) if bias else None
        R = torch.as_tensor(rng.normal(size=(n, dout)), dtype=torch.float32, device=DEV)
        y = F.dense(blocks, wt, b, ng.layers._act_code(act)[1], row_divs=divs, n=n)
        (y * R).sum().backward()
        # float64 restatement
        blocks64 = [bl.detach().double().requires_grad_(rd == 1) for bl, rd in zip(blocks, divs)]
        X = torch.cat([bl.repeat_interleave(rd, dim=0)[:n] if rd > 1 else bl for bl, rd in zip(blocks64, divs)], dim=1)
        w64 = wt.detach().double().requires_grad_(True)
        b64 = b.detach().double().requires_grad_(True) if bias else None
        y64 = REF[act](X @ w64 + (b64 if bias else 0.0))
        (y64 * R.double()).sum().backward()

        def rel(a, ref):
            return float((a.double() - ref).abs().max() / (ref.abs().max() + 1e-12))
        errs = {"y": rel(y.detach(), y64.detach()), "dW": rel(wt.grad, w64.grad)}
        if bias:
            errs["db"] = rel(b.grad, b64.grad)
        for k, (bl, bl64, rd) in enumerate(zip(blocks, blocks64, divs)):
            if rd == 1:
                errs[f"dx{k}"] = rel(bl.grad, bl64.grad)
        bad = {k: v for k, v in errs.items() if not v < (2e-5 if k == "y" else 3e-4)}
        if bad:
            failures.append((case, n, widths, divs, dout, act, bias, bad))
    assert not failures, failures


def test_random_vmh_node_shapes_plan_against_generic_solver():
    # tools/fuzz_vmh_node.py, a short fixed run: random depths, widths, activations, aggregation, coordinates, sizes on both sides of the
    # tile-round boundary, saveat -- NeuralODE(VMHConv) on the device-resident plan against the generic solver; a case above 5e-5 is
    # settled by a torch float64 autograd transcription (relu kinks).  One child process; the tool exits 1 on a mismatch
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CASES="10", SEED="5")
    for k in ("NGPDE_NO_VMH_NODE", "ONLY", "VERBOSE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_vmh_node.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    last = r.stdout.strip().splitlines()[-1]
    assert last.startswith("10 cases") and last.endswith("0 mismatches"), last
    if not os.environ.get("NGPDE_NO_VMH_NODE") and not os.environ.get("NGPDE_NO_PERSISTENT"):
        assert int(last.split(" cases, ")[1].split(" on the plan")[0]) >= 5, last


def test_random_layer_entries_against_the_composed_layers_and_the_oracle(monkeypatch):
    # tools/fuzz_layer_entries.py, a short fixed run: ExplicitEdgeConv / VMHConv / MPPDEConv / GNOConv of random shapes through the
    # layer-level C entries (ngpde_edge_layer_*, ngpde_gno_layer_*) -- every output and gradient bit for bit equal to the layer composed
    # from the primitives, training and inference, and (matrix state, + / mean / *) within the suite's tolerances of the float64 oracle.
    # Seed 1 holds two edgeless graphs with multi-layer message MLPs (a Dense over zero rows takes NULL blocks: found by this fuzz).
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("fuzz_layer_entries", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                                     "tools", "fuzz_layer_entries.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    monkeypatch.setenv("ORACLE", "1")
    failures = fz.run(40, 1)
    assert not failures, failures
    assert fz.n_oracle[0] >= 15
