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

