"""Randomised comparison of the persistent GCN solver in all its forms (one tile per workgroup, two members per workgroup, tile pairs,
tile rounds; relu sign bits or pre-activation tape; graphs with edge weights; d = 16 / 32 widened) with the replayed plan
(NGPDE_NO_PERSISTENT=1): u(T) and du0 bit for bit at d = 64 (to rounding for the widened widths, whose replayed plan runs other
kernels), parameter gradients to rounding.  usage: python tools/fuzz_gcn_node.py [CASES] [SEED]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ngpde_amd as ng
from ngpde_amd import synth as S

CASES = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
DEV = "cuda"
bad = 0
for case in range(CASES):
    n = int(rng.choice([40, 333, 1000, 4097, 9000, 16000, 16384, 16385, 20001, 30000, 33000, 36001, 50000]))
    act = str(rng.choice(["relu", "relu", "tanh", "swish", "identity", "leakyrelu"]))
    solver = str(rng.choice(["euler", "tsit5"])); steps = int(rng.integers(1, 4))
    K = int(rng.choice([1, 1, 2, 3, 5])) if n <= 16384 else 1
    bias = bool(rng.integers(0, 2))
    d = int(rng.choice([64, 64, 64, 32, 16]))
    weighted = bool(rng.integers(0, 3) == 0) and K == 1
    _, s, t = S.closest_pairs_graph(n, int(n * rng.choice([2, 4, 5])), seed=int(rng.integers(1, 1000)))
    ew = (0.25 + rng.random(s.size)).astype(np.float32) if weighted else None
    g1 = ng.GNNGraph(s, t, num_nodes=n, index_base=0, edge_weight=ew)
    g = ng.batch([g1] * K) if K > 1 else g1
    rhs = ng.Chain(ng.GCNConv((d, d), act, bias=bias, initialgraph=g, use_edge_weight=weighted),
                   ng.GCNConv((d, d), act, bias=bias, initialgraph=g, use_edge_weight=weighted))
    ps0, _ = ng.setup(case, rhs)
    ps0 = ng.to_device(ps0, DEV)
    if bias:
        for lp in ps0.values(): lp["bias"] = torch.randn_like(lp["bias"]) * 0.1
    u0 = torch.randn(d, n * K, device=DEV); R = torch.randn(d, n * K, device=DEV)
    outs = []
    for persistent in (True, False):
        if persistent: os.environ.pop("NGPDE_NO_PERSISTENT", None)
        else: os.environ["NGPDE_NO_PERSISTENT"] = "1"
        node = ng.NeuralODE(rhs, solver=solver, n_steps=steps, dt=0.04)
        _, st = ng.setup(case, node)
        ps = {l: {k: v.detach().clone().requires_grad_(True) for k, v in lp.items()} for l, lp in ps0.items()}
        u = u0.clone().requires_grad_(True)
        uT, _ = node(u, ps, st)
        (uT * R).sum().backward()
        plans = [p for pool in node._plans.values() for p in pool]
        outs.append((uT.detach(), u.grad, [v.grad for lp in ps.values() for v in lp.values()], plans))
    a, b = outs
    fl = sorted(a[3][0].flags()) if a[3] else []
    if d == 64:
        okb = torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    else:   # (relu kinks: a few du0 entries may differ by more than rounding between two correct plans)
        okb = torch.allclose(a[0], b[0], rtol=1e-4, atol=1e-5) and float(((a[1] - b[1]).abs() > 1e-4 + 1e-3 * b[1].abs()).double().mean()) < 2e-3
    okb = okb and not any(p.fault() for p in a[3])
    okp = all(torch.allclose(x, y, rtol=1e-4, atol=1e-4 * float(y.abs().max() + 1e-6)) for x, y in zip(a[2], b[2]))
    if not (okb and okp):
        bad += 1
        print("   u(T) max diff", float((a[0] - b[0]).abs().max()), "du0 max diff", float((a[1] - b[1]).abs().max()), "nan", bool(torch.isnan(a[0]).any()))
    print(f"case {case}: n={n} d={d} weighted={weighted} K={K} act={act} {solver}x{steps} bias={bias} edges={s.size} plan={fl} bitwise={okb} params={okp}", flush=True)
    del outs, a, b
    torch.cuda.empty_cache()
os.environ.pop("NGPDE_NO_PERSISTENT", None)
print(f"{CASES} cases, {bad} bad")
