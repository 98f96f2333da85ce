"""Randomised NeuralODE(VMHConv) on the device-resident plan against the generic solver (NGPDE_NO_VMH_NODE=1): depths, widths, activations,
aggregation, coordinates, graph sizes on both sides of the one-tile / tile-round boundary, saveat.  No oracle involved.
relu: the two solvers round pre-activations differently, and ONE unit within float32 rounding of its kink changes a random-signed
sum over N rows by about 1/sqrt(N) of its size (seed 2, case 22: a single gamma unit of node 2909, |z| < 4e-6, moves every parameter
gradient of the GENERIC solver by 1 % against torch float64 autograd, the plan by 1e-3; forward values agree to 1e-6).  So relu
gradients above 5e-5 are labelled "kink".  A parameter gradient is judged against max(its own largest entry, 1 % of the largest
parameter gradient): gamma's last bias gradient is ONE random-signed sum that can cancel.  Any case above 5e-5 is then settled by a
torch float64 autograd transcription of the same solve on the CPU and both distances are printed: a MISMATCH is the plan being
further from it than 5e-5 AND than twice the generic solver's own distance (relu: a difference above 1e-1, or in the outputs --
either solver may be the one that flips).
env: CASES (40), SEED, ONLY (case numbers, comma separated), VERBOSE (per-output differences)"""
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


def float64_autograd(g, pts, phi, gam, act, aggr, solver, steps, save, plist, u0, R):
    """the same solve and its gradients by torch float64 autograd on the CPU (src/layers.jl:308-332, fixed-step tableau): the arbiter
    of a flagged case.  plist: the float32 parameters in leaves() order, (out x in) weights and (out x 1) biases alternating."""
    from ngpde_amd.node import TABLEAUS
    s_, t_ = [torch.as_tensor(np.asarray(a.cpu() if hasattr(a, "cpu") else a)).long() for a in g.edge_index(index_base=0)]
    par = [p.detach().cpu().double().clone().requires_grad_(True) for p in plist]
    x, nv = pts.cpu().double(), pts.shape[1]
    f = {"tanh": torch.tanh, "relu": torch.relu, "sigmoid": torch.sigmoid}[act]
    n_phi = len(phi.chain)
    deg = torch.zeros(nv, dtype=torch.float64).index_add(0, t_, torch.ones(t_.numel(), dtype=torch.float64)).clamp(min=1)

    def run(ws, h):
        for li in range(len(ws) // 2):
            h = ws[2 * li] @ h + ws[2 * li + 1]
            h = f(h) if 2 * li + 2 < len(ws) else h
        return h

    def rhs(h):
        hi, hj = h[:, t_], h[:, s_]
        m = run(par[:2 * n_phi], torch.cat([hi, hj - hi, x[:, s_] - x[:, t_]], dim=0))
        agg = torch.zeros((m.shape[0], nv), dtype=torch.float64).index_add(1, t_, m)
        return run(par[2 * n_phi:], torch.cat([h, agg / deg if aggr == "mean" else agg], dim=0))

    a, b = TABLEAUS[solver]
    U = u0.detach().cpu().double().clone().requires_grad_(True)
    cur, saves = U, [U]
    for _ in range(steps):
        ks = []
        for i in range(len(b)):
            ks.append(rhs(cur + sum(0.03 * a[i][j] * ks[j] for j in range(i)) if i else cur))
        cur = cur + sum(0.03 * b[i] * ks[i] for i in range(len(b)))
        saves.append(cur)
    out = torch.stack(saves, dim=2) if save else cur
    (out * R.cpu().double()).sum().backward()
    return [out.detach(), U.grad] + [p.grad for p in par]


worst, bad, taken, kinks = 0.0, 0, 0, 0
only = os.environ.get("ONLY")
only = None if only is None else {int(v) for v in only.split(",")}
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
    if only is not None and case not in only:
        continue
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
        plist = [v.detach() for v in leaves(ps)]
        flags[mode] = sorted({f for pool in node._plans.values() for p in pool for f in p.flags()})
    os.environ.pop("NGPDE_NO_VMH_NODE", None)
    err, errs = 0.0, []
    gscale = max(float(b.abs().max()) for b in res["generic"][2:])      # a scalar gradient (gamma's last bias) is a random-signed sum that can cancel to ~0
    for idx, (a, b) in enumerate(zip(res["plan"], res["generic"])):
        if not torch.isfinite(a).all():
            err = float("inf")
            errs.append(err)
            break
        e1 = float((a - b).abs().max()) / max(float(b.abs().max()), 1e-2 * gscale if idx >= 2 else 1e-20)
        errs.append(e1)
        err = max(err, e1)
    if os.environ.get("VERBOSE"):
        print("      per output (u, du0, parameters...):", " ".join(f"{e:.1e}" for e in errs))
    on_plan = "vmh" in flags["plan"]
    taken += on_plan
    kink = act == "relu" and errs[0] <= 5e-5 and 5e-5 < err <= 1e-1
    wrong, tag = False, ""
    if err > 5e-5:                       # flagged: which of the two is off?  The plan must be as close to float64 as the generic solver is
        ref = float64_autograd(g, pts, phi, gam, act, aggr, solver, steps, save, plist, u0, R)
        per = {m: [float((a.cpu().double().reshape(b.shape) - b).abs().max()) / max(float(b.abs().max()), 1e-2 * gscale if idx >= 2 else 1e-20)
                   for idx, (a, b) in enumerate(zip(res[m], ref))] for m in ("plan", "generic")}
        dist = {m: max(v) for m, v in per.items()}
        if os.environ.get("VERBOSE"):
            for m, v in per.items():
                print(f"      {m:7s} vs float64:", " ".join(f"{e:.1e}" for e in v))
        wrong = not np.isfinite(err) or (not kink and dist["plan"] > max(5e-5, 2 * dist["generic"]))
        tag = (f"   vs float64 autograd: plan {dist['plan']:.1e} generic {dist['generic']:.1e}" + ("  (relu kink)" if kink else "")
               + ("   <-- MISMATCH" if wrong else ""))
    bad += wrong
    kinks += kink
    worst = max(worst, err if np.isfinite(err) else 1e9)
    print(f"case {case:3d}: N {nv:5d} k {k} pd {pd} {act:7s} {aggr:4s} {solver:5s} x{steps} phi {[2 + pd] + wp + [mw]} gamma {[1 + mw] + wg + [1]} saveat {save}: "
          f"{'plan' if on_plan else 'generic (not taken)'} rel err {err:.2e}{tag}", flush=True)
print(f"{cases} cases, {taken} on the plan, worst relative difference {worst:.2e}, {kinks} relu kink cases, {bad} mismatches")
sys.exit(1 if bad else 0)
