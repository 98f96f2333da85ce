"""BASELINE config 1's shape -- the graph_node.md tutorial: NeuralODE(2 x GCNConv(d => d, relu)) on a Cora-sized graph with Cora's degree
skew (2 708 nodes, hubs of degree > 100: its tiles do not fit the LDS halo, so the solver runs the replayed plan) -- beside a graph of
the same size WITHOUT hubs (closest pairs: the persistent plan).  Time per solve + adjoint and per ODE step.  env: STEPS (10), REPS (20)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ngpde_amd as ng
from ngpde_amd import synth as S

dev = "cuda:0"
N, PAIRS = 2708, 5278
steps, reps = int(os.environ.get("STEPS", 10)), int(os.environ.get("REPS", 20))
graphs = {}
s, t = S.preferential_pairs_graph(N, PAIRS, seed=1)
graphs["cora-like (hubs)"] = ng.GNNGraph(s, t, num_nodes=N, index_base=0)
_, s2, t2 = S.closest_pairs_graph(N, PAIRS, seed=1)
graphs["closest pairs (no hubs)"] = ng.GNNGraph(s2, t2, num_nodes=N, index_base=0)
for name, g in graphs.items():
    for d in (16, 32, 64):
        rhs = ng.Chain(ng.GCNConv((d, d), "relu", initialgraph=g), ng.GCNConv((d, d), "relu", initialgraph=g))
        node = ng.NeuralODE(rhs, solver="tsit5", n_steps=steps, dt=1.0 / steps)
        ps, st = ng.setup(0, node)
        ps = ng.to_device(ps, dev)
        for k in ("layer_1", "layer_2"):
            for v in ps[k].values():
                v.requires_grad_(True)
        u = torch.randn(d, N, device=dev, requires_grad=True)
        ts = []
        for rep in range(reps + 3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            uT, _ = node(u, ps, st)
            uT.sum().backward()
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        ms = 1e3 * float(np.median(ts[3:]))
        flags = sorted({f for p in node._plans.values() for f in (p.flags() if hasattr(p, "flags") else [])})
        print(f"{name:24s} d {d:3d}: {ms:7.3f} ms per solve + adjoint (Tsit5 x {steps}) = {1e3 * ms / steps:6.1f} us per ODE step = {steps / ms * 1e3:7.0f} ODE-steps/s  plan {flags}", flush=True)
