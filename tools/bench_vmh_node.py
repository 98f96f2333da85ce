"""NeuralODE(VMHConv(phi, gamma)) of docs/src/tutorials/VMH.md:75-89 at the tutorial's size (bench.py's secondary.VMH_node_tsit5x20 leg
alone): ms per 20-step Tsit5 solve + adjoint through the captured generic solver, and the launch count of one replay."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ngpde_amd as ng
from ngpde_amd import synth as S

dev = "cuda:0"
nv, kv, steps_v = 3000, 6, 20
pts = torch.as_tensor(S.uniform01(41, 2 * nv).reshape(2, nv).astype(np.float32), device=dev)
gv = ng.GNNGraph(ng.knn_graph(pts, kv), ndata={"x": pts})
phi = ng.Chain(ng.Dense(4, 60, "tanh"), ng.Dense(60, 60, "tanh"), ng.Dense(60, 60, "tanh"), ng.Dense(60, 40))
gam = ng.Chain(ng.Dense(41, 60, "tanh"), ng.Dense(60, 60, "tanh"), ng.Dense(60, 60, "tanh"), ng.Dense(60, 1))
node = ng.NeuralODE(ng.VMHConv(phi, gam, initialgraph=gv), solver="tsit5", n_steps=steps_v, dt=0.2 / steps_v, capture=True)
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


def solve():
    for v in [u] + list(leaves(ps)):
        v.grad = None
    uT, _ = node(u, ps, st)
    uT.sum().backward()


for _ in range(3):
    solve()
torch.cuda.synchronize()
t0 = time.perf_counter()
R = 5
for _ in range(R):
    solve()
torch.cuda.synchronize()
ms = 1e3 * (time.perf_counter() - t0) / R
print(f"VMH node tsit5x{steps_v}: {ms:.3f} ms per solve + adjoint = {steps_v / (ms * 1e-3):.1f} ODE-steps/s "
      f"({ms * 1e3 / (6 * steps_v):.1f} us per right-hand side + pullback)")
