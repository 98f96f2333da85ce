"""Where does a phase of the fused right-hand-side solver go (node_fused_rhs.hip)?  Needs the diagnostic library
(make -C neuralgraphpde.jl_amd/csrc diag).  Shader-clock stamps of thread 0 of every workgroup at 8 points of the first PH phases
of the forward and of the adjoint launch; prints the mean (over workgroups and phases) cycles between consecutive points."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ngpde_amd as ng
from ngpde_amd import _lib, synth as S

os.environ["NGPDE_FUSED_RHS"] = "1"
_lib.LIB_PATH = os.path.join(ROOT, "neuralgraphpde.jl_amd", "libngpde_diag.so")
from ngpde_amd.node import _Plan

N, PAIRS = int(os.environ.get("N", 16384)), int(os.environ.get("PAIRS", 65536))
D, STEPS, PH = 64, 50, 120
dev = "cuda:0"
_, s, t = S.closest_pairs_graph(N, PAIRS, seed=2)
g = ng.GNNGraph(s, t, num_nodes=N, index_base=0)
lib, p = _lib.load(), _lib.ptr
lib.ngpde_debug_set_fused_stamps.argtypes = [C.c_void_p, C.c_int32]
dv = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32), device=dev)
u0 = dv(S.normal(1000, D * N).reshape(N, D))
w1, w2 = dv(S.glorot_uniform(11, D, D).T), dv(S.glorot_uniform(12, D, D).T)
b1, b2 = dv(np.zeros(D)), dv(np.zeros(D))
seed = torch.ones_like(u0)
stream = torch.cuda.current_stream().cuda_stream
plan = _Plan(g.handle((True, None, False)), D, _lib.ACT["relu"], "tsit5", STEPS, 1.0 / 50, True)
assert "fused_rhs" in plan.flags(), plan.flags()
outs = [torch.empty_like(u0), torch.empty_like(u0), torch.empty_like(w1), torch.empty_like(b1), torch.empty_like(w2), torch.empty_like(b2)]
NT = N // 32
names_f = ["wait for the neighbours' flags", "gather 2-hop rows (LDS-DMA) + barrier", "layer-1 aggregation on H1 + barrier", "layer-1 product + epilogue + barrier",
           "layer-2 aggregation + barrier", "layer-2 product + barrier", "epilogue + row stores issued", "drain + barrier + flag (+ masks)"]
names_b = ["prefetch + wait for the flags", "gather 2-hop rows + barrier", "dL/dy1 aggregation on H1, dZ1 + barrier", "G1 product + epilogue + barrier",
           "U-bar aggregation, stage terms, dZ2 + barrier", "G2 product + barrier", "row stores issued", "drain + flag + dW / db products"]


def run(which):
    buf = torch.zeros(NT * PH * 8, dtype=torch.int64, device=dev)
    for rep in range(3):
        lib.ngpde_debug_set_fused_stamps(p(buf) if (rep == 2 and which == "fwd") else None, PH)
        _lib.check(lib.ngpde_node_gcn2_forward(plan.ptr, p(u0), p(w1), p(b1), p(w2), p(b2), p(outs[0]), stream))
        lib.ngpde_debug_set_fused_stamps(p(buf) if (rep == 2 and which == "bwd") else None, PH)
        _lib.check(lib.ngpde_node_gcn2_backward(plan.ptr, p(seed), p(outs[1]), p(outs[2]), p(outs[3]), p(outs[4]), p(outs[5]), stream))
        torch.cuda.synchronize()
    st = buf.cpu().numpy().reshape(NT, PH, 8).astype(np.float64)
    names = names_f if which == "fwd" else names_b
    sel = st[:, 8:PH - 1, :]
    nxt = st[:, 9:PH, 0]
    d = np.diff(np.concatenate([sel, nxt[:, :, None]], axis=2), axis=2)
    whole = nxt - sel[:, :, 0]
    print(f"{which}: " + "; ".join(f"{nm}: {d[:, :, k].mean():.0f}" for k, nm in enumerate(names)) + f" | phase start-to-start: {whole.mean():.0f} cycles")
    for k, nm in enumerate(names):
        print(f"   {nm}: mean {d[:, :, k].mean():.0f}  p10 {np.percentile(d[:, :, k], 10):.0f}  p90 {np.percentile(d[:, :, k], 90):.0f}")
    print("flags", sorted(plan.flags()), "fault", plan.fault())


run("fwd")
run("bwd")
