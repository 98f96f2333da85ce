"""The C2 solve (16 384 nodes / 65 536 closest pairs, Tsit5 x 50, forward + adjoint) at widths d = 16 / 32 / 64 / 128: the widened
persistent plan (d = 16, 32 run zero-padded on the 64-wide kernels) against the native-width replayed plan (NGPDE_NO_WIDEN=1).
One JSON line per (d, mode).  `python tools/bench_node_widths.py [N]`"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ngpde_amd as ng
from ngpde_amd import synth as S

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
_, s, t = S.closest_pairs_graph(N, 4 * N, seed=2)
g = ng.GNNGraph(s, t, num_nodes=N, index_base=0)
for d in (16, 32, 64, 128):
    for widen in (True, False):
        if d >= 64 and not widen: continue
        if widen: os.environ.pop("NGPDE_NO_WIDEN", None)
        else: os.environ["NGPDE_NO_WIDEN"] = "1"
        rhs = ng.Chain(ng.GCNConv((d, d), "relu", initialgraph=g), ng.GCNConv((d, d), "relu", initialgraph=g))
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
        print(json.dumps(dict(d=d, nodes=N, flags=sorted(plan.flags()), ms_per_solve_and_adjoint=round(ms, 3),
                              ode_steps_per_s=round(50 / ms * 1e3, 1))), flush=True)
