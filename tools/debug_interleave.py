"""Interleaved (two members per workgroup) vs member-by-member persistent batch solve, output by output (env: N, K, STEPS, TAB)."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ngpde_amd as ng
from ngpde_amd import _lib, synth as S
from ngpde_amd.node import _Plan

N, D, K = int(os.environ.get("N", 16384)), 64, int(os.environ.get("K", 2))
STEPS, TAB = int(os.environ.get("STEPS", 2)), os.environ.get("TAB", "tsit5")
dev = "cuda:0"
_, s, t = S.closest_pairs_graph(N, 4 * N, seed=2)
g = ng.GNNGraph(s, t, num_nodes=N, index_base=0)
lib, p = _lib.load(), _lib.ptr
h = g.handle((True, None, False))
ok, unused = C.c_size_t(), C.c_void_p()
_lib.check(lib.ngpde_graph_array(h.ptr, 0, 13, C.byref(unused), C.byref(ok)))
print("halo_ok(by target) =", ok.value, flush=True)
dv = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32), device=dev)
u0 = dv(S.normal(1000, D * N * K).reshape(N * K, D))
w1, w2 = dv(S.glorot_uniform(11, D, D).T), dv(S.glorot_uniform(12, D, D).T)
b1, b2 = dv(S.normal(5, D) * 0.1), dv(S.normal(6, D) * 0.1)
seed = dv(S.normal(7, D * N * K).reshape(N * K, D))
stream = torch.cuda.current_stream().cuda_stream


def run(interleave, members=K, bwd=True):
    if interleave:
        os.environ.pop("NGPDE_NO_INTERLEAVE", None)
    else:
        os.environ["NGPDE_NO_INTERLEAVE"] = "1"
    plan = _Plan(h, D, _lib.ACT["relu"], TAB, STEPS, 1.0 / 50, True, members=members)
    print("plan", members, sorted(plan.flags()), flush=True)
    n = N * members
    outs = [torch.empty_like(u0[:n]), torch.empty_like(u0[:n]), torch.empty_like(w1), torch.empty_like(b1), torch.empty_like(w2), torch.empty_like(b2)]
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _lib.check(lib.ngpde_node_gcn2_forward(plan.ptr, p(u0), p(w1), p(b1), p(w2), p(b2), p(outs[0]), stream))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        if bwd:
            _lib.check(lib.ngpde_node_gcn2_backward(plan.ptr, p(seed), p(outs[1]), p(outs[2]), p(outs[3]), p(outs[4]), p(outs[5]), stream))
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    af, ab, tot = C.c_int64(), C.c_int64(), C.c_int64()
    _lib.check(lib.ngpde_node_pipeline_stats(plan.ptr, stream, C.byref(af), C.byref(ab), C.byref(tot)))
    print(("interleaved" if interleave else "one by one "), f"fwd {1e3 * (t1 - t0):.3f} ms  bwd {1e3 * (t2 - t1):.3f} ms  fault={plan.fault()}"
          f"  gathered ahead: fwd {af.value}/{tot.value} bwd {ab.value}/{tot.value}", flush=True)
    return [o.clone() for o in outs]


run(False, members=1)
a = run(False)
if os.environ.get("FWD_ONLY"):
    b = run(True, bwd=False)
else:
    b = run(True)
for name, x, y in zip(["uT", "du0", "dw1", "db1", "dw2", "db2"], a, b):
    err = float((x - y).abs().max())
    print(f"{name}: max diff {err:.3e} rel {err / max(float(x.abs().max()), 1e-30):.2e} nan={bool(torch.isnan(y).any())} bitwise_equal={bool(torch.equal(x, y))}")
bb = run(True, bwd=not os.environ.get("FWD_ONLY"))
print("interleaved run-to-run bitwise:", [bool(torch.equal(x, y)) for x, y in zip(b, bb)])
one = run(False, members=1, bwd=False)
for m in range(K):
    sl = slice(m * N, (m + 1) * N)
    da = float((a[0][sl] - b[0][sl]).abs().max())
    print(f"member {m}: |one-by-one - interleaved| = {da:.3e}; rows differing: {int(((a[0][sl] != b[0][sl]).any(dim=1)).sum())} of {N}")
print("member 0 of one-by-one batch == single-member plan:", bool(torch.equal(a[0][:N], one[0])), " interleaved:", bool(torch.equal(b[0][:N], one[0])))
