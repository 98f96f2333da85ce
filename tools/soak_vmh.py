"""Race screen of the device-resident NeuralODE(VMHConv) plan: REPS solves + adjoints on the same inputs, every output compared bit for bit
with the first one's (no atomics, fixed summation orders: any difference is a missed hand-off).  env: N (3000: one half tile per workgroup;
9000 / 24000: tile rounds), STEPS (20), REPS (40)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ngpde_amd as ng
from ngpde_amd import synth as S

dev = "cuda:0"
nv, steps, reps = int(os.environ.get("N", 3000)), int(os.environ.get("STEPS", 20)), int(os.environ.get("REPS", 40))
pts = torch.as_tensor(S.uniform01(41, 2 * nv).reshape(2, nv).astype(np.float32), device=dev)
gv = ng.GNNGraph(ng.knn_graph(pts, 6), ndata={"x": pts})
phi = ng.Chain(ng.Dense(4, 60, "tanh"), ng.Dense(60, 60, "tanh"), ng.Dense(60, 60, "tanh"), ng.Dense(60, 40))
gam = ng.Chain(ng.Dense(41, 60, "tanh"), ng.Dense(60, 60, "tanh"), ng.Dense(60, 60, "tanh"), ng.Dense(60, 1))
node = ng.NeuralODE(ng.VMHConv(phi, gam, initialgraph=gv), solver="tsit5", n_steps=steps, dt=0.2 / steps)
ps, st = ng.setup(4, node)
ps = ng.to_device(ps, dev)


def leaves(t):
    for v in t.values():
        if isinstance(v, dict):
            yield from leaves(v)
        else:
            yield v


for v in leaves(ps):
    v.requires_grad_(True)
u = torch.as_tensor(S.normal(42, nv).reshape(1, nv).astype(np.float32), device=dev).requires_grad_(True)
R = torch.as_tensor(S.normal(43, nv).reshape(1, nv).astype(np.float32), device=dev)
first, bad = None, 0
for rep in range(reps):
    for v in [u] + list(leaves(ps)):
        v.grad = None
    uT, _ = node(u, ps, st)
    (uT * R).sum().backward()
    out = [uT.detach().clone(), u.grad.clone()] + [v.grad.clone() for v in leaves(ps)]
    if first is None:
        first = out
    else:
        bad += sum(0 if torch.equal(a, b) else 1 for a, b in zip(first, out))
flags = sorted({f for pool in node._plans.values() for p in pool for f in p.flags()})
fault = [p.fault() for pool in node._plans.values() for p in pool]
print(f"nodes {nv}, Tsit5 x {steps}, {reps} solves + adjoints on plan {flags}: {bad} differing outputs of {(reps - 1) * len(first)}, fault {fault}")
sys.exit(1 if bad or any(fault) else 0)
