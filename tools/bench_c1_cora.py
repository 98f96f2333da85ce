"""BASELINE config 1's shape -- the graph_node.md tutorial: NeuralODE(2 x GCNConv(d => d, relu)) on a Cora-sized graph with Cora's degree
skew (2 708 nodes, hubs of degree > 100: its tiles do not fit the handle's 96-row halo lists; the solver runs the persistent kernels'
hub geometry, NGPDE_NO_PERSISTENT=1 the replayed plan with the per-row gather) -- beside a graph of the same size WITHOUT hubs (closest
pairs: the persistent plan in the 96-row geometry).  Device time per solve + adjoint through the C ABI (the plan's two entries, as
bench.py's headline drives them) and per ODE step.  env: STEPS (10), REPS (20), TABLEAU (tsit5)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ngpde_amd as ng
from ngpde_amd import _lib, synth as S
from ngpde_amd.node import _Plan

dev = "cuda:0"
N, PAIRS = 2708, 5278
steps, reps, tab = int(os.environ.get("STEPS", 10)), int(os.environ.get("REPS", 20)), os.environ.get("TABLEAU", "tsit5")
lib, p = _lib.load(), _lib.ptr
graphs = {}
s, t = S.preferential_pairs_graph(N, PAIRS, seed=1)
graphs["cora-like (hubs)"] = ng.GNNGraph(s, t, num_nodes=N, index_base=0)
_, s2, t2 = S.closest_pairs_graph(N, PAIRS, seed=1)
graphs["closest pairs (no hubs)"] = ng.GNNGraph(s2, t2, num_nodes=N, index_base=0)
stream = torch.cuda.current_stream().cuda_stream
for name, g in graphs.items():
    for d in (16, 32, 64):
        plan = _Plan(g.handle((True, None, False)), d, _lib.ACT["relu"], tab, steps, 1.0 / steps, True)
        u, seed = torch.randn(N, d, device=dev), torch.ones(N, d, device=dev)
        w = [torch.as_tensor(S.glorot_uniform(50 + k, d, d).astype(np.float32), device=dev) for k in range(2)]
        b = [torch.zeros(d, device=dev) for _ in range(2)]
        uT, du = torch.empty_like(u), torch.empty_like(u)
        gr = [torch.empty_like(w[0]), torch.empty_like(b[0]), torch.empty_like(w[1]), torch.empty_like(b[1])]

        def solve():
            _lib.check(lib.ngpde_node_gcn2_forward(plan.ptr, p(u), p(w[0]), p(b[0]), p(w[1]), p(b[1]), p(uT), stream))
            _lib.check(lib.ngpde_node_gcn2_backward(plan.ptr, p(seed), p(du), p(gr[0]), p(gr[1]), p(gr[2]), p(gr[3]), stream))
        for _ in range(3):
            solve()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            solve()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        assert not plan.fault()
        print(f"{name:24s} d {d:3d}: {ms:7.3f} ms per solve + adjoint ({tab} x {steps}) = {1e3 * ms / steps:6.1f} us per ODE step = {steps / ms * 1e3:7.0f} ODE-steps/s  plan {sorted(plan.flags())}", flush=True)
