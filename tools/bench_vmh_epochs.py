"""The VMH tutorial's training step with a freshly shuffled DataLoader batch every epoch (VMH.md:120-141): a new block-diagonal graph per
step, i.e. a new graph handle and a new device-resident plan per step.  Time of batch + updategraph and of the first / second solve +
adjoint on the new graph.  env: NB (24), EPOCHS (5); NGPDE_NO_VMH_NODE=1 for the generic solver"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ngpde_amd as ng
from ngpde_amd import synth as S

dev = "cuda:0"
nb, nv, steps, epochs = int(os.environ.get("NB", 24)), 3000, 20, int(os.environ.get("EPOCHS", 5))
clouds = []
for kb in range(nb):
    pk = torch.as_tensor(S.uniform01(200 + kb, 2 * nv).reshape(2, nv).astype(np.float32), device=dev)
    clouds.append(ng.GNNGraph(ng.knn_graph(pk, 6), ndata={"x": pk}))
phi = ng.Chain(ng.Dense(4, 60, "tanh"), ng.Dense(60, 60, "tanh"), ng.Dense(60, 60, "tanh"), ng.Dense(60, 40))
gam = ng.Chain(ng.Dense(41, 60, "tanh"), ng.Dense(60, 60, "tanh"), ng.Dense(60, 60, "tanh"), ng.Dense(60, 1))
node = ng.NeuralODE(ng.VMHConv(phi, gam), solver="tsit5", n_steps=steps, dt=0.01)
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
u = torch.randn(1, nb * nv, device=dev, requires_grad=True)
rng = np.random.default_rng(0)
for epoch in range(epochs):
    perm = rng.permutation(nb)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    gb = ng.batch([clouds[i] for i in perm])
    st2 = ng.updategraph(st, gb)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    uT, _ = node(u, ps, st2)
    uT.sum().backward()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    uT, _ = node(u, ps, st2)
    uT.sum().backward()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    flags = sorted({f for pool in node._plans.values() for p in pool for f in p.flags()})
    print(f"epoch {epoch}: batch + updategraph {1e3 * (t1 - t0):.1f} ms, first solve + adjoint on the new graph {1e3 * (t2 - t1):.1f} ms, second {1e3 * (t3 - t2):.1f} ms  {flags}", flush=True)
