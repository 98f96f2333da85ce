"""NeuralODE(VMHConv) of docs/src/tutorials/VMH.md:75-89 at the tutorial's size (bench.py's secondary.VMH_node_tsit5x20 workload), a few
solves + adjoints for `rocprofv3 --kernel-trace --stats`: which launches does a right-hand side of this shape consist of?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ngpde_amd as ng
from ngpde_amd import synth as S
dev = "cuda"
nv, kv, steps = 3000, 6, int(os.environ.get("STEPS", 5))
pts = torch.as_tensor(S.uniform01(41, 2 * nv).reshape(2, nv).astype(np.float32), device=dev)
gv = ng.GNNGraph(ng.knn_graph(pts, kv), ndata={"x": pts})
phi = ng.Chain(ng.Dense(4, 60, "tanh"), ng.Dense(60, 60, "tanh"), ng.Dense(60, 60, "tanh"), ng.Dense(60, 40))
gam = ng.Chain(ng.Dense(41, 60, "tanh"), ng.Dense(60, 60, "tanh"), ng.Dense(60, 60, "tanh"), ng.Dense(60, 1))
node = ng.NeuralODE(ng.VMHConv(phi, gam, initialgraph=gv), solver="tsit5", n_steps=steps, dt=0.2 / steps, capture=os.environ.get("CAPTURE", "0") == "1")
ps, st = ng.setup(4, node)
ps = ng.to_device(ps, dev)
def leaves(d):
    out = []
    for v in d.values():
        out += leaves(v) if isinstance(v, dict) else [v]
    return out
for v in leaves(ps): v.requires_grad_(True)
u = torch.as_tensor(S.normal(42, nv).reshape(1, nv).astype(np.float32), device=dev).requires_grad_(True)
for rep in range(3):
    for v in [u] + leaves(ps): v.grad = None
    uT, _ = node(u, ps, st)
    uT.sum().backward()
torch.cuda.synchronize()
print("done", steps, "steps per solve, 3 solves")
