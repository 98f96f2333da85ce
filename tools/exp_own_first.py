"""Round 6: the plan's own-first slot tables against the handle's order.  NGPDE_NO_OWN_FIRST is read when a plan is created, so
one process builds both plans on the same handle and times them alternately on the same box: forward and adjoint launch time (events),
u(T) and du0 of one against the other (the orders differ, so the results differ in rounding only)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ngpde_amd as ng
from ngpde_amd import _lib, synth as S
from ngpde_amd.node import _Plan

N, PAIRS, D, STEPS = 16384, 65536, 64, 50
dev = "cuda:0"
_, s, t = S.closest_pairs_graph(N, PAIRS, seed=2)
g = ng.GNNGraph(s, t, num_nodes=N, index_base=0)
lib, p = _lib.load(), _lib.ptr
dv = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32), device=dev)
u0 = dv(S.normal(1000, D * N).reshape(N, D))
w1, w2 = dv(S.glorot_uniform(11, D, D).T), dv(S.glorot_uniform(12, D, D).T)
b1, b2 = dv(0.01 * S.normal(5, D)), dv(0.01 * S.normal(6, D))
seed = torch.ones_like(u0)
stream = torch.cuda.current_stream().cuda_stream
plans = {}
for mode in ("1", "0"):
    os.environ["NGPDE_NO_OWN_FIRST"] = mode
    plans[mode] = _Plan(g.handle((True, None, False)), D, _lib.ACT["relu"], "tsit5", STEPS, 1.0 / 50, True)
os.environ.pop("NGPDE_NO_OWN_FIRST")
plan = None
outs = [torch.empty_like(u0), torch.empty_like(u0), torch.empty_like(w1), torch.empty_like(b1), torch.empty_like(w2), torch.empty_like(b2)]


def fwd():
    _lib.check(lib.ngpde_node_gcn2_forward(plan.ptr, p(u0), p(w1), p(b1), p(w2), p(b2), p(outs[0]), stream))


def bwd():
    _lib.check(lib.ngpde_node_gcn2_backward(plan.ptr, p(seed), p(outs[1]), p(outs[2]), p(outs[3]), p(outs[4]), p(outs[5]), stream))


ref = None
REPS = int(os.environ.get("REPS", "20"))
for rnd in range(int(os.environ.get("ROUNDS", "2"))):
    for mode in ("1", "0"):
        plan = plans[mode]
        for _ in range(3):
            fwd(); bwd()
        torch.cuda.synchronize()
        tf, tb = [], []
        for _ in range(REPS):
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            e[0].record(); fwd(); e[1].record(); bwd(); e[2].record()
            torch.cuda.synchronize()
            tf.append(e[0].elapsed_time(e[1])); tb.append(e[1].elapsed_time(e[2]))
        uT = outs[0].clone()
        du0 = outs[1].clone()
        if ref is None:
            ref = (uT, du0)
        print(f"round {rnd} NGPDE_NO_OWN_FIRST={mode}: forward {np.median(tf):.4f} ms (min {min(tf):.4f}), adjoint {np.median(tb):.4f} ms, "
              f"max|uT - uT0| {float((uT - ref[0]).abs().max()):.3e} (max|uT| {float(ref[0].abs().max()):.3f}), "
              f"max|du0 - du0_0| {float((du0 - ref[1]).abs().max()):.3e}, fault {plan.fault()}", flush=True)
