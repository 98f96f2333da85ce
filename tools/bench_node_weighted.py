"""The C2 solve (Tsit5 x 50, forward + adjoint) on a graph WITH stored edge weights, GCNConv(use_edge_weight=true): the tile-round
persistent kernels with the slot weights in LDS against the replayed plan (NGPDE_NO_PERSISTENT=1).  `python tools/bench_node_weighted.py [N]`"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ngpde_amd as ng
from ngpde_amd import synth as S

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
d = 64
_, s, t = S.closest_pairs_graph(N, 4 * N, seed=2)
ew = (0.25 + np.random.default_rng(1).random(s.size)).astype(np.float32)
g = ng.GNNGraph(s, t, num_nodes=N, index_base=0, edge_weight=ew)
for persistent in (True, False):
    if persistent: os.environ.pop("NGPDE_NO_PERSISTENT", None)
    else: os.environ["NGPDE_NO_PERSISTENT"] = "1"
    rhs = ng.Chain(ng.GCNConv((d, d), "relu", initialgraph=g, use_edge_weight=True), ng.GCNConv((d, d), "relu", initialgraph=g, use_edge_weight=True))
    node = ng.NeuralODE(rhs, solver="tsit5", n_steps=50, dt=0.02)
    ps, st = ng.setup(0, node)
    ps = ng.to_device(ps, "cuda")
    for lp in ps.values():
        for v in lp.values(): v.requires_grad_(True)
    u = torch.randn(N, d, device="cuda").T.requires_grad_(True)
    R = torch.randn(N, d, device="cuda").T
    def step():
        u.grad = None
        for lp in ps.values():
            for v in lp.values(): v.grad = None
        y, _ = node(u, ps, st)
        y.backward(R)
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    plan = next(iter(node._plans.values()))[0]
    print(json.dumps(dict(weighted=True, nodes=N, flags=sorted(plan.flags()), ms_per_solve_and_adjoint=round(ms, 3),
                          ode_steps_per_s=round(50 / ms * 1e3, 1))), flush=True)
