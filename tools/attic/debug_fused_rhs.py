"""Fused right-hand-side plan (one hand-off per evaluation, node_fused_rhs.hip) vs the one-hop persistent plan, output by output,
on the bench workload cut to STEPS steps (env: STEPS, TAB, N, PAIRS, REPS)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ngpde_amd as ng
from ngpde_amd import _lib, synth as S
from ngpde_amd.node import _Plan

N, PAIRS, D = int(os.environ.get("N", 16384)), int(os.environ.get("PAIRS", 65536)), 64
STEPS, TAB, REPS = int(os.environ.get("STEPS", 2)), os.environ.get("TAB", "tsit5"), int(os.environ.get("REPS", 5))
dev = "cuda:0"
_, s, t = S.closest_pairs_graph(N, PAIRS, seed=2)
g = ng.GNNGraph(s, t, num_nodes=N, index_base=0)
lib, p = _lib.load(), _lib.ptr
dv = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32), device=dev)
u0 = dv(S.normal(1000, D * N).reshape(N, D))
w1, w2 = dv(S.glorot_uniform(11, D, D).T), dv(S.glorot_uniform(12, D, D).T)
b1, b2 = dv(S.normal(5, D) * 0.1), dv(S.normal(6, D) * 0.1)
seed = dv(S.normal(7, D * N).reshape(N, D))
stream = torch.cuda.current_stream().cuda_stream


def run(fused):
    if fused:
        os.environ["NGPDE_FUSED_RHS"] = "1"
    else:
        os.environ.pop("NGPDE_FUSED_RHS", None)
    plan = _Plan(g.handle((True, None, False)), D, _lib.ACT["relu"], TAB, STEPS, 1.0 / 50, True)
    outs = [torch.empty_like(u0), torch.empty_like(u0), torch.empty_like(w1), torch.empty_like(b1), torch.empty_like(w2), torch.empty_like(b2)]
    tf, tb = [], []
    for rep in range(REPS):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _lib.check(lib.ngpde_node_gcn2_forward(plan.ptr, p(u0), p(w1), p(b1), p(w2), p(b2), p(outs[0]), stream))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        _lib.check(lib.ngpde_node_gcn2_backward(plan.ptr, p(seed), p(outs[1]), p(outs[2]), p(outs[3]), p(outs[4]), p(outs[5]), stream))
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        tf.append(1e3 * (t1 - t0))
        tb.append(1e3 * (t2 - t1))
    print(("fused  " if fused else "one-hop"), sorted(plan.flags()), f"fwd {min(tf):.3f} ms  bwd {min(tb):.3f} ms  fault={plan.fault()}", flush=True)
    return [o.clone() for o in outs]


a = run(False)
b = run(True)
for name, x, y in zip(["uT", "du0", "dw1", "db1", "dw2", "db2"], a, b):
    err = float((x - y).abs().max())
    print(f"{name}: max|one-hop|={float(x.abs().max()):.4e} max diff {err:.3e} nan={bool(torch.isnan(y).any())} bitwise_equal={bool(torch.equal(x, y))}")
