"""Race screen of the device-resident GAT solver: REPS solves + adjoints of BASELINE config 3 as ODE right-hand side (and of a batch
of MEMBERS copies), every output of every replay compared bit for bit with the first.  usage: python tools/soak_gat_node.py [REPS]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ngpde_amd as ng
from ngpde_amd import synth as S
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 100
K = int(os.environ.get("MEMBERS", 1))
_, s, t = S.closest_pairs_graph(16384, 65536, seed=2)
g1 = ng.GNNGraph(s, t, num_nodes=16384, index_base=0)
g = ng.batch([g1] * K) if K > 1 else g1
l = ng.GATConv((64, 16), "relu", heads=4, initialgraph=g)
node = ng.NeuralODE(l, solver="tsit5", n_steps=50, dt=0.02)
ps, st = ng.setup(5, node)
ps = ng.to_device(ps, "cuda")
for v in ps.values(): v.requires_grad_(True)
u = torch.randn(64, 16384 * K, device="cuda", requires_grad=True)
first, diff = None, 0
for rep in range(REPS):
    for v in list(ps.values()) + [u]: v.grad = None
    uT, _ = node(u, ps, st)
    uT.sum().backward()
    got = [uT.detach().clone(), u.grad.clone()] + [ps[k].grad.clone() for k in sorted(ps)]
    if first is None: first = got
    elif not all(torch.equal(a, b) for a, b in zip(got, first)): diff += 1
plans = [p for pool in node._plans.values() for p in pool]
print(f"GAT solver, {K} member(s): {REPS} replays, {diff} differing from the first; plans {[sorted(p.flags()) for p in plans]}; fault={any(p.fault() for p in plans)}; finite={all(bool(torch.isfinite(x).all()) for x in first)}")
