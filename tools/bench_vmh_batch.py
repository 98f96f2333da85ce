"""NeuralODE(VMHConv) on a batch of point clouds as one block-diagonal graph (docs/src/tutorials/VMH.md:120-134 batches 24 clouds of 3 000
points): the device-resident plan in tile rounds against the captured generic solver.  env: NB (24), NV (3000), STEPS (20), REPS (2)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ngpde_amd as ng
from ngpde_amd import synth as S

dev = "cuda:0"
nb, nv, steps, reps = int(os.environ.get("NB", 24)), int(os.environ.get("NV", 3000)), int(os.environ.get("STEPS", 20)), int(os.environ.get("REPS", 2))
clouds = []
for kb in range(nb):
    pk = torch.as_tensor(S.uniform01(200 + kb, 2 * nv).reshape(2, nv).astype(np.float32), device=dev)
    clouds.append(ng.GNNGraph(ng.knn_graph(pk, 6), ndata={"x": pk}))
gb = ng.batch(clouds)
phi = ng.Chain(ng.Dense(4, 60, "tanh"), ng.Dense(60, 60, "tanh"), ng.Dense(60, 60, "tanh"), ng.Dense(60, 40))
gam = ng.Chain(ng.Dense(41, 60, "tanh"), ng.Dense(60, 60, "tanh"), ng.Dense(60, 60, "tanh"), ng.Dense(60, 1))
N = nb * nv
u0 = torch.as_tensor(S.normal(42, N).reshape(1, N).astype(np.float32), device=dev)


def leaves(t):
    for v in t.values():
        if isinstance(v, dict):
            yield from leaves(v)
        else:
            yield v


for mode in ("plan", "generic"):
    if mode == "generic":
        os.environ["NGPDE_NO_VMH_NODE"] = "1"
    node = ng.NeuralODE(ng.VMHConv(phi, gam, initialgraph=gb), solver="tsit5", n_steps=steps, dt=0.2 / steps, capture=(mode == "generic"))
    ps, st = ng.setup(4, node)
    ps = ng.to_device(ps, dev)
    for v in leaves(ps):
        v.requires_grad_(True)
    u = u0.clone().requires_grad_(True)
    ts = []
    for rep in range(reps + 1):
        for v in [u] + list(leaves(ps)):
            v.grad = None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        uT, _ = node(u, ps, st)
        uT.sum().backward()
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    flags = sorted({f for pool in node._plans.values() for p in pool for f in p.flags()})
    print(f"{mode}: {nb} clouds x {nv} points, {min(ts[1:]):.2f} ms per solve + adjoint = {nb * steps / (min(ts[1:]) * 1e-3):.0f} trajectory ODE-steps/s, plan {flags}",
          flush=True)
    del node
