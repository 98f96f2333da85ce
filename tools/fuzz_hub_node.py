"""Randomised graphs with hubs on the persistent solver's hub geometry against the replayed plan of the same build (NGPDE_NO_PERSISTENT=1):
preferential-attachment graphs of 200 - 8 000 nodes and 1.2 - 3 pairs per node (hubs of degree 30 - 250), sometimes with extra one-way
edges into or out of the largest hub, d = 16 / 32 / 64, smooth activations (relu kinks at hubs spread: tests/test_hub_plan_gpu.py), Euler
or Tsit5.  u(T), du0 and the parameter gradients must agree to 2e-5 / 1e-4 relative (another summation order in hub rows: not bitwise),
no fault; a graph whose tiles do not fit the geometry's caps (in practice: a hub of more than 224 distinct in+out neighbours, which the
extra one-way edges produce in about one case of seven) must fall back to the replayed plan, bitwise, and the line prints the setup's reason.
usage: python3 tools/fuzz_hub_node.py [cases=30] [seed=1]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ngpde_amd as ng  # noqa: E402
from ngpde_amd import _lib, synth as S  # noqa: E402

DEV = "cuda:0"


def solve(g, d, act, tab, nsteps, dt, params, u0, R, weighted=False):
    kw = dict(initialgraph=g, use_edge_weight=True) if weighted else dict(initialgraph=g)
    rhs = ng.Chain(ng.GCNConv((d, d), act, **kw), ng.GCNConv((d, d), act, **kw))
    node = ng.NeuralODE(rhs, solver=tab, n_steps=nsteps, dt=dt)
    _, st = ng.setup(0, node)
    ps = {f"layer_{k + 1}": {"weight": torch.as_tensor(params[k]["weight"].astype(np.float32), device=DEV).requires_grad_(True),
                             "bias": torch.as_tensor(params[k]["bias"].astype(np.float32), device=DEV).requires_grad_(True)}
          for k in range(2)}
    u = torch.as_tensor(u0.astype(np.float32), device=DEV).requires_grad_(True)
    uT, _ = node(u, ps, st)
    plan = next(iter(node._plans.values()))[0]
    (uT * torch.as_tensor(R.astype(np.float32), device=DEV)).sum().backward()
    grads = [ps[f"layer_{k + 1}"][n].grad.clone() for k in range(2) for n in ("weight", "bias")]
    return uT.detach().clone(), u.grad.clone(), grads, plan.flags(), plan.fault()


def rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def main(cases, seed):
    rng = np.random.default_rng(seed)
    bad, n_hub = [], 0
    for case in range(cases):
        N = int(rng.choice([200, 700, 1500, 2708, 4000, 8000]))
        pairs = int(N * rng.uniform(1.2, 3.0))
        s, t = S.preferential_pairs_graph(N, pairs, seed=int(rng.integers(1, 10000)))
        deg = np.bincount(t, minlength=N)
        hub = int(deg.argmax())
        extra = int(rng.choice([0, 0, 40, 120]))
        if extra:      # one-way edges at the largest hub: its row grows in one direction's lists only
            others = rng.choice(np.setdiff1d(np.arange(N), [hub]), size=extra, replace=False)
            if rng.random() < 0.5:
                s, t = np.concatenate([s, np.full(extra, hub)]), np.concatenate([t, others])
            else:
                s, t = np.concatenate([s, others]), np.concatenate([t, np.full(extra, hub)])
        d = int(rng.choice([16, 32, 64]))
        act = str(rng.choice(["tanh", "swish", "sigmoid"]))
        tab, nsteps = ("euler", 4) if rng.random() < 0.4 else ("tsit5", 2)
        params = [dict(weight=S.glorot_uniform(int(rng.integers(1, 1000)), d, d), bias=rng.normal(size=(d, 1)) * 0.1) for _ in range(2)]
        # round 6: a third of the cases with stored edge weights (use_edge_weight = true), a third as a batch of 2 - 3 same-structure members
        weighted = rng.random() < 0.33
        members = int(rng.choice([2, 3])) if (rng.random() < 0.33 and N <= 4000) else 1
        ew = (0.25 + rng.random(s.size)).astype(np.float32) if weighted else None
        g = ng.GNNGraph(s, t, num_nodes=N, index_base=0, edge_weight=ew)
        if members > 1:
            g = ng.batch([g] + [g.copy() for _ in range(members - 1)])
        u0, R = rng.normal(size=(d, members * N)), rng.normal(size=(d, members * N))
        os.environ.pop("NGPDE_NO_PERSISTENT", None)
        a = solve(g, d, act, tab, nsteps, 0.1, params, u0, R, weighted)
        why = "" if "hub_geometry" in a[3] else " (" + _lib.load().ngpde_last_error().decode()[:120] + ")"
        os.environ["NGPDE_NO_PERSISTENT"] = "1"
        b = solve(g, d, act, tab, nsteps, 0.1, params, u0, R, weighted)
        os.environ.pop("NGPDE_NO_PERSISTENT", None)
        is_hub = "hub_geometry" in a[3]
        n_hub += is_hub
        errs = [rel(a[0], b[0]), rel(a[1], b[1])] + [rel(x, y) for x, y in zip(a[2], b[2])]
        ok = (not a[4]) and errs[0] <= 2e-5 and errs[1] <= 1e-4 and max(errs[2:]) <= 2e-4 and "persistent_fwd" not in b[3]
        print(f"case {case}: N={N} E={s.size} max degree {int(np.bincount(t, minlength=N).max())} d={d} {act} {tab}{' weighted' if weighted else ''}{' x' + str(members) if members > 1 else ''}: "
              f"{'hub geometry' if is_hub else 'plan ' + str(sorted(a[3])) + why}; u(T) {errs[0]:.1e} du0 {errs[1]:.1e} grads {max(errs[2:]):.1e}"
              f"{'' if ok else '   <-- FAIL'}", flush=True)
        if not ok:
            bad.append(case)
    print(f"fuzz_hub_node: {cases} cases, seed {seed}: {len(bad)} failing, {n_hub} on the hub geometry")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 30, int(sys.argv[2]) if len(sys.argv) > 2 else 1))
