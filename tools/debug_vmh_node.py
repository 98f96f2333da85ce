"""NeuralODE(VMHConv) on the device-resident plan (ngpde_node_vmh_*) against the generic solver (NGPDE_NO_VMH_NODE=1): outputs and
gradients side by side, and the time per solve + adjoint.  env: N (3000), K (6), STEPS (20), TAB (tsit5), REPS (5)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ngpde_amd as ng
from ngpde_amd import synth as S

dev = "cuda:0"
nv, kv, steps = int(os.environ.get("N", 3000)), int(os.environ.get("K", 6)), int(os.environ.get("STEPS", 20))
tab, reps = os.environ.get("TAB", "tsit5"), int(os.environ.get("REPS", 5))
pts = torch.as_tensor(S.uniform01(41, 2 * nv).reshape(2, nv).astype(np.float32), device=dev)
gv = ng.GNNGraph(ng.knn_graph(pts, kv), ndata={"x": pts})
phi = ng.Chain(ng.Dense(4, 60, "tanh"), ng.Dense(60, 60, "tanh"), ng.Dense(60, 60, "tanh"), ng.Dense(60, 40))
gam = ng.Chain(ng.Dense(41, 60, "tanh"), ng.Dense(60, 60, "tanh"), ng.Dense(60, 60, "tanh"), ng.Dense(60, 1))
u0 = torch.as_tensor(S.normal(42, nv).reshape(1, nv).astype(np.float32), device=dev)
R = torch.as_tensor(S.normal(43, nv).reshape(1, nv).astype(np.float32), device=dev)


def leaves(t):
    for v in t.values():
        if isinstance(v, dict):
            yield from leaves(v)
        else:
            yield v


def run(resident):
    if resident:
        os.environ.pop("NGPDE_NO_VMH_NODE", None)
    else:
        os.environ["NGPDE_NO_VMH_NODE"] = "1"
    node = ng.NeuralODE(ng.VMHConv(phi, gam, initialgraph=gv), solver=tab, n_steps=steps, dt=0.2 / steps, capture=not resident)
    ps, st = ng.setup(4, node)
    ps = ng.to_device(ps, dev)
    rng = np.random.default_rng(3)
    for v in leaves(ps):
        if v.shape[-1] == 1:
            v.copy_(torch.as_tensor(rng.normal(size=tuple(v.shape)).astype(np.float32) * 0.2))   # non-zero biases
        v.requires_grad_(True)
    u = u0.clone().requires_grad_(True)
    ts = []
    for rep in range(reps):
        for v in [u] + list(leaves(ps)):
            v.grad = None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        uT, _ = node(u, ps, st)
        (uT * R).sum().backward()
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    plans = [p for pool in node._plans.values() for p in pool]
    print(("resident" if resident else "generic "), [sorted(p.flags()) for p in plans], f"{min(ts):.3f} ms per solve + adjoint = {steps / (min(ts) * 1e-3):.0f} ODE-steps/s",
          "fault", [p.fault() for p in plans], flush=True)
    return [uT.detach().clone(), u.grad.clone()] + [v.grad.clone() for v in leaves(ps)]


a = run(False)
b = run(True)
names = ["uT", "du0"] + [f"dparam{k}" for k in range(len(a) - 2)]
for nm, x, y in zip(names, a, b):
    err = float((x - y).abs().max())
    print(f"{nm}: shape {tuple(x.shape)} max|generic| {float(x.abs().max()):.4e} max diff {err:.3e} rel {err / max(float(x.abs().max()), 1e-30):.2e} nan={bool(torch.isnan(y).any())}")
