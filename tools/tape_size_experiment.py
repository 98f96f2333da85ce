"""Diagnostic: per-launch device time of the plan's kernels as a function of the number of ODE steps
(= tape size), to separate TLB / first-touch effects from the kernels' own work."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ngpde_amd as ng
from ngpde_amd import _lib, synth as S
from ngpde_amd.node import _Plan
lib = _lib.load()
N, D = 16384, 64
_, s, t = S.closest_pairs_graph(N, 65536, seed=2)
g = ng.GNNGraph(s, t, num_nodes=N, index_base=0)
h = g.handle((True, None, False))
dev = "cuda"
u0 = torch.randn(N, D, device=dev); w1 = torch.randn(D, D, device=dev) * 0.1; w2 = torch.randn(D, D, device=dev) * 0.1
b1 = torch.zeros(D, device=dev); b2 = torch.zeros(D, device=dev)
uT = torch.empty_like(u0); du0 = torch.empty_like(u0); seed = torch.ones_like(u0)
dw1 = torch.empty_like(w1); dw2 = torch.empty_like(w1); db1 = torch.empty_like(b1); db2 = torch.empty_like(b1)
st = torch.cuda.current_stream().cuda_stream; p = _lib.ptr
for steps in (1, 2, 5, 10, 25, 50):
    plan = _Plan(h, D, 1, "tsit5", steps, 0.02, True)
    for _ in range(3):
        _lib.check(lib.ngpde_node_gcn2_forward(plan.ptr, p(u0), p(w1), p(b1), p(w2), p(b2), p(uT), st))
        _lib.check(lib.ngpde_node_gcn2_backward(plan.ptr, p(seed), p(du0), p(dw1), p(db1), p(dw2), p(db2), st))
    torch.cuda.synchronize()
    us = (C.c_float * 4)(); cnt = (C.c_int32 * 4)()
    _lib.check(lib.ngpde_node_profile(plan.ptr, 1, us, cnt, st))
    print(f"steps={steps:3d} tape={plan.tape_bytes()/1e6:8.1f} MB  fwd1 {us[0]:6.2f}  fwd2 {us[1]:6.2f}  bwd1 {us[2]:6.2f}  bwd2 {us[3]:6.2f} us")
    del plan
