"""Randomised NeuralODE(VMHConv) on the device-resident plan against the generic solver (NGPDE_NO_VMH_NODE=1): depths, widths, activations,
aggregation, coordinates, graph sizes on both sides of the one-tile / tile-round boundary, saveat.  No oracle involved.  env: CASES (40), SEED"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ngpde_amd as ng
from ngpde_amd import synth as S

dev = "cuda:0"
cases, seed = int(os.environ.get("CASES", 40)), int(os.environ.get("SEED", 1))
rng = np.random.default_rng(seed)


def leaves(t):
    for v in t.values():
        if isinstance(v, dict):
            yield from leaves(v)
        else:
            yield v


def mlp(din, widths, dout, act):
    dims = [din] + widths + [dout]
    return ng.Chain(*[ng.Dense(dims[k], dims[k + 1], act if k + 2 < len(dims) else "identity") for k in range(len(dims) - 1)])


worst, bad, taken = 0.0, 0, 0
for case in range(cases):
    pd = int(rng.integers(1, 4))
    nv = int(rng.choice([300, 1000, 2900, 4100, 5000, 8200, 9500]))
    k = int(rng.integers(3, 9))
    act = str(rng.choice(["tanh", "relu", "sigmoid"]))
    aggr = str(rng.choice(["mean", "+"]))
    solver = str(rng.choice(["tsit5", "euler"]))
    steps = int(rng.integers(1, 4))
    depth_p, depth_g = int(rng.integers(2, 5)), int(rng.integers(2, 5))
    mw = int(rng.integers(1, 64))
    wp = [int(rng.integers(1, 65)) for _ in range(depth_p - 1)]
    wg = [int(rng.integers(1, 65)) for _ in range(depth_g - 1)]
    save = bool(rng.integers(0, 2)) and steps > 1
    pts = torch.as_tensor(S.uniform01(1000 + case, pd * nv).reshape(pd, nv).astype(np.float32), device=dev)
    g = ng.GNNGraph(ng.knn_graph(pts, k), ndata={"x": pts})
    phi, gam = mlp(2 + pd, wp, mw, act), mlp(1 + mw, wg, 1, act)
    u0 = torch.as_tensor(S.normal(2000 + case, nv).reshape(1, nv).astype(np.float32), device=dev)
    res, flags = {}, {}
    for mode in ("plan", "generic"):
        if mode == "generic":
            os.environ["NGPDE_NO_VMH_NODE"] = "1"
        else:
            os.environ.pop("NGPDE_NO_VMH_NODE", None)
        kw = dict(saveat=0.03) if save else {}
        node = ng.NeuralODE(ng.VMHConv(phi, gam, aggr=aggr, initialgraph=g), solver=solver, n_steps=steps, dt=0.03, **kw)
        ps, st = ng.setup(7 + case, node)
        ps = ng.to_device(ps, dev)
        prng = np.random.default_rng(case)
        for v in leaves(ps):
            if v.shape[-1] == 1:
                v.copy_(torch.as_tensor(prng.normal(size=tuple(v.shape)).astype(np.float32) * 0.2))
            v.requires_grad_(True)
        u = u0.clone().requires_grad_(True)
        out, _ = node(u, ps, st)
        R = torch.as_tensor(np.random.default_rng(5 + case).normal(size=tuple(out.shape)).astype(np.float32), device=dev)
        (out * R).sum().backward()
        res[mode] = [out.detach().clone(), u.grad.clone()] + [v.grad.clone() for v in leaves(ps)]
        flags[mode] = sorted({f for pool in node._plans.values() for p in pool for f in p.flags()})
    os.environ.pop("NGPDE_NO_VMH_NODE", None)
    err = 0.0
    for a, b in zip(res["plan"], res["generic"]):
        if not torch.isfinite(a).all():
            err = float("inf")
            break
        err = max(err, float((a - b).abs().max()) / max(float(b.abs().max()), 1e-20))
    on_plan = "vmh" in flags["plan"]
    taken += on_plan
    tag = "" if err <= 5e-5 else "   <-- MISMATCH"
    bad += err > 5e-5
    worst = max(worst, err if np.isfinite(err) else 1e9)
    print(f"case {case:3d}: N {nv:5d} k {k} pd {pd} {act:7s} {aggr:4s} {solver:5s} x{steps} phi {[2 + pd] + wp + [mw]} gamma {[1 + mw] + wg + [1]} saveat {save}: "
          f"{'plan' if on_plan else 'generic (not taken)'} rel err {err:.2e}{tag}", flush=True)
print(f"{cases} cases, {taken} on the plan, worst relative difference {worst:.2e}, {bad} mismatches")
sys.exit(1 if bad else 0)
