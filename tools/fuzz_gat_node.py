"""Randomised comparison of the device-resident GAT solver with the generic solver (bit for bit in u(T), du0): node counts from one
partial tile up, 1 / 2 / 4 heads, all activations, Euler / Tsit5, batches of 1..3 members, graphs with isolated nodes and without
self loops.  usage: python tools/fuzz_gat_node.py [CASES] [SEED]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ngpde_amd as ng

CASES = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
DEV = "cuda"
ACTS = ["identity", "relu", "tanh", "sigmoid", "swish", "gelu", "leakyrelu", "elu", "softplus"]


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


bad = 0
for case in range(CASES):
    n = int(rng.choice([5, 31, 32, 33, 64, 100, 257, 700, 1500]))
    H = int(rng.choice([1, 2, 4])); act = str(rng.choice(ACTS)); solver = str(rng.choice(["euler", "tsit5"]))
    steps = int(rng.integers(1, 4)); K = int(rng.choice([1, 1, 2, 3])); loops = bool(rng.integers(0, 4) > 0); bias = bool(rng.integers(0, 2))
    s, t = local_graph(n, int(rng.integers(1, 12)), int(rng.integers(1, min(7, max(2, n // 2)))))
    if os.environ.get('ONLY') and int(os.environ['ONLY']) != case:
        torch.randn(1); continue
    g1 = ng.GNNGraph(s, t, num_nodes=n, index_base=0)
    g = ng.batch([g1] * K) if K > 1 else g1
    l = ng.GATConv((64, 64 // H), act, heads=H, add_self_loops=loops, bias=bias, initialgraph=g)
    ps0, _ = ng.setup(case, l)
    ps0 = ng.to_device(ps0, DEV)
    u0 = torch.randn(64, n * K, device=DEV)
    R = torch.randn(64, n * K, device=DEV)
    outs = []
    for resident in (True, False):
        if resident: os.environ.pop("NGPDE_NO_PERSISTENT", None)
        else: os.environ["NGPDE_NO_PERSISTENT"] = "1"
        node = ng.NeuralODE(l, solver=solver, n_steps=steps, dt=0.05)
        _, st = ng.setup(case, node)
        ps = {k: v.detach().clone().requires_grad_(True) for k, v in ps0.items()}
        u = u0.clone().requires_grad_(True)
        uT, _ = node(u, ps, st)
        (uT * R).sum().backward()
        plans = [p for pool in node._plans.values() for p in pool]
        outs.append((uT.detach(), u.grad, {k: v.grad for k, v in ps.items()}, plans))
    a, b = outs
    used = bool(a[3]) and all("gat" in p.flags() for p in a[3])
    okb = torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and not any(p.fault() for p in a[3])
    okp = all(torch.allclose(a[2][k], b[2][k], rtol=1e-4, atol=1e-4 * float(b[2][k].abs().max() + 1e-6)) for k in a[2])
    if not (okb and okp):
        bad += 1
        print('   u(T) max diff', float((a[0] - b[0]).abs().max()), 'nan', bool(torch.isnan(a[0]).any()), bool(torch.isnan(b[0]).any()), '| du0 max diff', float((a[1] - b[1]).abs().max()),
              'cols differing', int(((a[1] != b[1]).any(dim=0)).sum()), 'of', a[1].shape[1], '| dparams', {k: float((a[2][k] - b[2][k]).abs().max()) for k in a[2]})
    print(f"case {case}: n={n} K={K} H={H} act={act} {solver}x{steps} loops={loops} bias={bias} edges={s.size} resident={used} bitwise={okb} params={okp}", flush=True)
os.environ.pop("NGPDE_NO_PERSISTENT", None)
print(f"{CASES} cases, {bad} bad")
